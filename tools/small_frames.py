import sys, time, os, cProfile, pstats
sys.path.insert(0, "pgsd-sph_amd")
import numpy as np, torch
import pgsd.fl as fl
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = 3000
pos = torch.randn((N, 4), device="cuda"); vel = torch.randn((N, 4), device="cuda")
path = "/dev/shm/small_frames.gsd"
counts = np.array([N])
def run(async_seal):
    f = fl.open(path, "w", application="a", schema="hoomd", schema_version=[1, 4])
    fields = [("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
              ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
              ("particles/typeid", fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True))]
    for i in range(50):
        f.write_chunks(fields, offset=counts); f.end_frame(wait=not async_seal)
    f.frame_sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(F):
        f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
        f.write_chunks(fields, offset=counts)
        f.end_frame(wait=not async_seal)
    f.frame_sync()
    dt = time.perf_counter() - t0
    f.close(); os.unlink(path)
    return dt / F * 1e6
print("N", N, "sync  %.0f us/frame" % run(False))
print("N", N, "async %.0f us/frame" % run(True))
if len(sys.argv) > 2:
    pr = cProfile.Profile(); pr.enable(); run(False); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
