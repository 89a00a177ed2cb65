#!/usr/bin/env python3
"""Long run of `pgsd.hoomd.HOOMDTrajectory.append` with GPU-resident attributes and the default elision (every array
compared with frame 0 / the default value on the GPU in every frame): memory watched along the way.

    python tools/soak_hoomd.py [particles=16384] [frames=20000]

Position and velocity move, type id / mass / density stay, the body holds the default, the image flips between the
default and other values; frames are sealed asynchronously two out of three times."""
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import numpy as np
import torch
import pgsd.fl as fl
import pgsd.hoomd as hoomd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
path = "/dev/shm/pgsd_soak_hoomd_%d.gsd" % os.getpid()
g = torch.Generator(device="cuda").manual_seed(3)
pos4 = torch.rand((N, 4), generator=g, device="cuda")
pos4[:, 3] = torch.randint(0, 4, (N,), generator=g, device="cuda", dtype=torch.int32).view(torch.float32)
vel4 = torch.randn((N, 4), generator=g, device="cuda")
vel4[:, 3] = 1.5
dens = torch.rand((N,), generator=g, device="cuda")
body = torch.full((N,), -1, device="cuda", dtype=torch.int32)
img0 = torch.zeros((N, 4), device="cuda", dtype=torch.int32)
img1 = torch.ones((N, 4), device="cuda", dtype=torch.int32)
D = fl.DeviceField.from_tensor
t = hoomd.open(path, "w")
t0 = time.perf_counter()
marks = []
for i in range(frames):
    pos4[:, :3] += 0.001
    vel4[:, :3] *= 1.0001
    fr = hoomd.Frame()
    fr.configuration.step = i
    fr.particles.N = N
    fr.particles.position = D(pos4, columns=(0, 3))
    fr.particles.typeid = D(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True)
    fr.particles.velocity = D(vel4, columns=(0, 3))
    fr.particles.mass = D(vel4, columns=(3, 4))
    fr.particles.density = dens
    fr.particles.body = body
    fr.particles.image = D(img1 if (i // 7) % 2 else img0, columns=(0, 3))
    t.append(fr, wait=(i % 3 == 0))
    t.file.wait_packed()
    if i % (frames // 10) == 0:
        free = torch.cuda.mem_get_info()[0]
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        marks.append((i, free >> 20, rss >> 10, len(os.listdir("/proc/self/fd"))))
        print("frame %6d  gpu free %d MiB  max rss %d MiB  fds %d  %.1f s" % (marks[-1] + (time.perf_counter() - t0,)), flush=True)
t.file.frame_sync()
dt = time.perf_counter() - t0
t.close()
size = os.path.getsize(path)
with hoomd.open(path, "r") as r:
    last = r[frames - 1]
    ok = (last.particles.position.tobytes() == pos4[:, :3].contiguous().cpu().numpy().tobytes()
          and last.particles.mass.tobytes() == vel4[:, 3].contiguous().cpu().numpy().tobytes()
          and int(last.particles.body[0]) == -1
          and not r.file.chunk_exists(frames - 1, "particles/typeid") and not r.file.chunk_exists(5, "particles/body")
          and r.file.chunk_exists(7, "particles/image") and not r.file.chunk_exists(14, "particles/image"))
os.unlink(path)
print("frames %d  %.1f us per frame  file %.2f GB (%.0f %% of writing everything)  last frame ok %s  gpu-free drift %d MiB  "
      "rss drift %d MiB" % (frames, dt / frames * 1e6, size / 1e9, 100.0 * size / (frames * N * 56.0), ok,
                            marks[1][1] - marks[-1][1], marks[-1][2] - marks[1][2]))
