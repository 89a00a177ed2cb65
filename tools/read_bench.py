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
pos4 = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
vel4 = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
times = []
for i in range(a.frames):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.read_chunk_device(i, 'particles/position', out=pos4, columns=(0, 3), wait=False)
    r.read_chunk_device(i, 'particles/typeid', out=pos4, columns=(3, 4), bitcast=True, wait=False)
    r.read_chunk_device(i, 'particles/velocity', out=vel4, columns=(0, 3), wait=False)
    r.wait_read()
    times.append(time.perf_counter() - t0)
ok = bool(torch.equal(pos4.view(torch.int32), pos.view(torch.int32)))
r.close()
os.unlink(path)
best = min(times)
print(json.dumps({"N": N, "frames": a.frames, "bytes_per_frame": 28 * N, "ms": [round(t * 1e3, 2) for t in times],
                  "best_GBps": round(28 * N / best / 1e9, 2), "round_trip_bit_exact": ok}))
