#!/usr/bin/env python3
"""Restart-read throughput (BASELINE config 5 shape): write F frames of N particles
(position, velocity, typeid) with the device path, then time reading them back into Scalar4
arrays on the GPU (pread -> pinned slabs -> HBM -> HIP unpack)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))

import numpy as np
import torch

import pgsd.fl as fl

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=10_000_000)
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--part", type=int, nargs=2, default=None, metavar=("ROW0", "ROWS"),
                help="read only this rank-like partition of every chunk (BASELINE config 5: 10 M rows of an 80 M file)")
ap.add_argument("--slab-mib", type=int, default=0)
ap.add_argument("--slabs", type=int, default=0)
a = ap.parse_args()
N = a.N
path = os.path.join(a.dir, "pgsd_read_bench_%d.gsd" % os.getpid())
g = torch.Generator(device="cuda").manual_seed(1)
pos = torch.randn((N, 4), generator=g, device="cuda")
vel = torch.randn((N, 4), generator=g, device="cuda")
f = fl.open(path, 'w', application='read_bench', schema='hoomd', schema_version=[1, 4])
for i in range(a.frames):
    f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                    ('particles/typeid', fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True)),
                    ('particles/velocity', fl.DeviceField.from_tensor(vel, columns=(0, 3)))], offset=np.array([N]))
    f.end_frame()
f.close()
r = fl.open(path, 'r')
if a.slab_mib or a.slabs:
    r.configure_device(slab_bytes=(a.slab_mib or 16) << 20, n_slabs=a.slabs or 16)
row0, n_read = (a.part[0], a.part[1]) if a.part else (0, N)
pos4 = torch.zeros((n_read, 4), dtype=torch.float32, device="cuda")
vel4 = torch.zeros((n_read, 4), dtype=torch.float32, device="cuda")
times = []
for i in range(a.frames):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.read_chunk_device(i, 'particles/position', out=pos4, N=n_read, offset=row0, columns=(0, 3), wait=False)
    r.read_chunk_device(i, 'particles/typeid', out=pos4, N=n_read, offset=row0, columns=(3, 4), bitcast=True, wait=False)
    r.read_chunk_device(i, 'particles/velocity', out=vel4, N=n_read, offset=row0, columns=(0, 3), wait=False)
    r.wait_read()
    times.append(time.perf_counter() - t0)
ok = bool(torch.equal(pos4.view(torch.int32), pos[row0:row0 + n_read].view(torch.int32))
          and torch.equal(vel4[:, :3].view(torch.int32), vel[row0:row0 + n_read, :3].contiguous().view(torch.int32)))
r.close()
os.unlink(path)
best = min(times)
print(json.dumps({"N": N, "part": [row0, n_read], "frames": a.frames, "bytes_per_frame": 28 * n_read, "ms": [round(t * 1e3, 2) for t in times],
                  "best_GBps": round(28 * n_read / best / 1e9, 2), "round_trip_bit_exact": ok}))
