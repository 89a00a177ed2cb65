#!/usr/bin/env python3
"""Kernel-only timing of the unpack (read-path) kernel: dense position / typeid / velocity chunks
in HBM -> Scalar4 arrays. Rotating buffer sets defeat the Infinity Cache."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from pgsd import _lib
import gpu_common as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = 40
with_mass = len(sys.argv) > 2 and sys.argv[2] == "mass"   # velocity.w too: both arrays get whole rows
with_fill = len(sys.argv) > 2 and sys.argv[2] == "fill"   # no mass chunk; velocity.w = 1.0 through fill_rest: whole rows
sets = []
for s in range(3):
    pos = torch.randn((N, 3), device="cuda")
    vel = torch.randn((N, 3), device="cuda")
    tid = torch.randint(0, 5, (N, 1), device="cuda", dtype=torch.int32)
    pos4 = torch.zeros((N, 4), device="cuda")
    vel4 = torch.zeros((N, 4), device="cuda")
    mass = torch.rand((N, 1), device="cuda")
    spec = [(pos, pos4, 3, 0, 0), (tid, pos4, 1, 3, 1), (vel, vel4, 3, 0, 0)] + ([(mass, vel4, 1, 3, 0)] if with_mass else [])
    nj = len(spec)
    jobs = (_lib.UnpackJob * nj)()
    for i, (src, dst, M, c0, bc) in enumerate(spec):
        jobs[i].src = src.data_ptr()
        jobs[i].src_type = G.type_id(str(src.dtype)[6:]) if src.dtype != torch.int32 else 3
        jobs[i].M = M
        jobs[i].dst.dst = dst.data_ptr()
        jobs[i].dst.dst_type = 9
        jobs[i].dst.dst_stride = 4
        jobs[i].dst.dst_col0 = c0
        jobs[i].dst.bitcast = bc
        if with_fill and dst is vel4:
            jobs[i].dst.fill_rest = 1
            jobs[i].dst.fill_bits = 0x3F800000          # 1.0f
    sets.append((jobs, (pos, vel, tid, pos4, vel4, mass)))
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
for i in range(6):
    assert _lib.lib.pgsd_unpack_fields(nj, sets[i % 3][0], N, ctypes.c_void_p(stream)) == 0
for i in range(iters):
    evs[i][0].record()
    _lib.lib.pgsd_unpack_fields(nj, sets[i % 3][0], N, ctypes.c_void_p(stream))
    evs[i][1].record()
torch.cuda.synchronize()
ts = np.array([a.elapsed_time(b) for a, b in evs]) * 1e-3
med = float(np.median(ts))
pos, vel, tid, pos4, vel4, mass = sets[0][1]
per = 64 if with_mass else 56
ok = bool(torch.equal(pos4[:, :3], pos) and torch.equal(vel4[:, :3], vel) and torch.equal(pos4[:, 3].view(torch.int32), tid[:, 0]))
if with_fill:
    ok = ok and bool((vel4[:, 3] == 1.0).all())
print(json.dumps({"N": N, "mass": with_mass, "fill": with_fill, "median_us": round(med * 1e6, 1), "min_us": round(float(ts.min()) * 1e6, 1),
                  "algo_bytes_per_particle": per, "algo_GBps": round(per * N / med / 1e9, 1), "frac": round(per * N / med / 8e12, 3), "correct": ok}))
