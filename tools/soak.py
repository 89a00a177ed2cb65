#!/usr/bin/env python3
"""Long run of the device path: many asynchronous frames, memory watched along the way."""
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import numpy as np
import torch
import pgsd.fl as fl

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
path = "/dev/shm/pgsd_soak_%d.gsd" % os.getpid()
pos = torch.randn((N, 4), device="cuda")
vel = torch.randn((N, 4), device="cuda")
f = fl.open(path, "w", application="soak", schema="hoomd", schema_version=[1, 4])
batched = len(sys.argv) > 3 and sys.argv[3] == "batched"
f.frame_exchange = batched                                 # one exchange per frame, chunks staged then placed
t0 = time.perf_counter()
marks = []
for i in range(frames):
    pos.add_(0.001)                                        # the "simulation" between snapshots
    f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
    f.write_chunks([("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                    ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
                    ("particles/typeid", fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True))],
                   offset="auto" if batched else np.array([N]))
    f.end_frame(wait=False)
    f.wait_packed()
    if i % (frames // 10) == 0:
        free = torch.cuda.mem_get_info()[0]
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        nfd = len(os.listdir("/proc/self/fd"))
        marks.append((i, free >> 20, rss >> 10, nfd))
        print("frame %6d  gpu free %d MiB  max rss %d MiB  fds %d  %.1f s" % (i, free >> 20, rss >> 10, nfd, time.perf_counter() - t0), flush=True)
f.frame_sync()
dt = time.perf_counter() - t0
last = f.read_chunk_device(frames - 1, "particles/position")
ok = bool(torch.equal(last, pos[:, :3].contiguous()))
n = f.nframes
f.close()
size = os.path.getsize(path)
os.unlink(path)
print("frames %d  %.2f GB/s  file %.1f GB  last frame ok %s  gpu-free drift %d MiB  rss drift %d MiB" % (
    n, frames * N * 28 / dt / 1e9, size / 1e9, ok, marks[1][1] - marks[-1][1], marks[-1][2] - marks[1][2]))
