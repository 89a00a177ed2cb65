#!/usr/bin/env python3
"""Timing of the stream compaction (pgsd_select_rows: count, scan, scatter) on N flags.  Since round 5 the call returns the
count in HOST memory (the scan kernel stores it into a pinned word; one stream wait inside the call): the figures are the whole call, the three
kernels alone take 20-28 us at 10 M flags (profiles/r01_select_bench.jsonl)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import numpy as np
import torch
from pgsd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
lib = _lib.lib
for p in (0.01, 0.5, 0.99):
    sets = []
    for k in range(4):
        flags = (torch.rand(N, device="cuda") < p).to(torch.uint8)
        index = torch.empty(N, dtype=torch.int32, device="cuda")
        count = ctypes.c_uint64(0)          # host memory: the call returns when the count is known
        sets.append((flags, index, count, None))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    iters = 40
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i in range(iters + 4):
        f, ix, c, w = sets[i % 4]
        if i >= 4:
            evs[i - 4][0].record()
        assert lib.pgsd_select_rows(f.data_ptr(), N, ix.data_ptr(), ctypes.byref(c), stream) == 0
        if i >= 4:
            evs[i - 4][1].record()
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) for a, b in evs]) * 1e-3
    f, ix, c, w = sets[0]
    k = int(c.value)
    ok = bool(torch.equal(ix[:k].to(torch.int64), torch.nonzero(f).flatten()))
    med = float(np.median(ts))
    algo = N + 4 * k          # flag bytes read + index bytes written
    print(json.dumps({"N": N, "keep": p, "selected": k, "median_us": round(med * 1e6, 1), "min_us": round(float(ts.min()) * 1e6, 1),
                      "algo_GBps": round(algo / med / 1e9, 1), "moved_GBps": round((2 * N + 4 * k) / med / 1e9, 1), "correct": ok}))
