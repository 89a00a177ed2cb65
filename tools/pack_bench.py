#!/usr/bin/env python3
"""Kernel-only timing of the pack path (pgsd_pack_fields through the C ABI).

Rotates enough independent buffer sets that no byte is re-read from the 256 MiB Infinity Cache
between launches, times with HIP events on the launch stream, and prints algorithmic and
moved GB/s per workload.  Kernels and launch shapes are selected with the PGSD_PACK_* environment
variables read by the launcher (PGSD_PACK_KERNEL=rows|tiles, PGSD_PACK_ROWS_CFG=256x2, ...).
"""
import argparse
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


def make_jobs(workload, N, gen):
    """-> (jobs for G.hip_pack-like raw call, algorithmic bytes, moved bytes, keepalive)"""
    f32, f64, i32 = torch.float32, torch.float64, torch.int32
    keep = []

    def rnd(shape, dt):
        t = torch.randn(shape, generator=gen, device="cuda", dtype=torch.float32).to(dt)
        keep.append(t)
        return t

    def out(n, m, dt):
        t = torch.empty((n, m), device="cuda", dtype=dt)
        keep.append(t)
        return t

    jobs = []
    if workload == "pos_vel_id":          # float4 pos, float4 vel, u32 id
        pos, vel = rnd((N, 4), f32), rnd((N, 4), f32)
        tid = torch.randint(0, 1 << 30, (N, 1), generator=gen, device="cuda", dtype=i32); keep.append(tid)
        jobs = [(out(N, 3, f32), np.float32, 3, pos, 0, None, False),
                (out(N, 3, f32), np.float32, 3, vel, 0, None, False),
                (out(N, 1, i32), np.uint32, 1, tid, 0, None, False)]
        algo, moved = 56 * N, 64 * N
    elif workload == "hoomd_pvi":         # bench.py's layout: position+velocity+typeid, typeid in pos.w
        pos, vel = rnd((N, 4), f32), rnd((N, 4), f32)
        jobs = [(out(N, 3, f32), np.float32, 3, pos, 0, None, False),
                (out(N, 1, i32), np.uint32, 1, pos, 3, None, True),
                (out(N, 3, f32), np.float32, 3, vel, 0, None, False)]
        algo, moved = 56 * N, 60 * N
    elif workload == "hoomd_w":           # typeid in pos.w, mass in vel.w: every byte is payload
        pos, vel = rnd((N, 4), f32), rnd((N, 4), f32)
        jobs = [(out(N, 3, f32), np.float32, 3, pos, 0, None, False),
                (out(N, 1, i32), np.uint32, 1, pos, 3, None, True),
                (out(N, 3, f32), np.float32, 3, vel, 0, None, False),
                (out(N, 1, f32), np.float32, 1, vel, 3, None, False)]
        algo, moved = 64 * N, 64 * N
    elif workload == "double4":           # Scalar = double builds: f64 -> f32 conversion
        pos, vel = rnd((N, 4), f64), rnd((N, 4), f64)
        tid = torch.randint(0, 1 << 30, (N, 1), generator=gen, device="cuda", dtype=i32); keep.append(tid)
        jobs = [(out(N, 3, f32), np.float32, 3, pos, 0, None, False),
                (out(N, 3, f32), np.float32, 3, vel, 0, None, False),
                (out(N, 1, i32), np.uint32, 1, tid, 0, None, False)]
        algo, moved = (24 + 24 + 4 + 28) * N, (32 + 32 + 4 + 28) * N
    elif workload in ("gather", "gather_hilbert", "gather_identity"):   # tag order through a permutation: uniformly random (adversarial),
        # or lattice-order tags over Hilbert-curve memory order (what HOOMD's SFC sorter leaves; bench_legs.hilbert_order)
        pos, vel = rnd((N, 4), f32), rnd((N, 4), f32)
        if workload == "gather":
            order = torch.randperm(N, generator=gen, device="cuda").to(i32)
        elif workload == "gather_identity":     # right after initialisation: memory order IS tag order
            order = torch.arange(N, device="cuda", dtype=i32)
        else:
            sys.path.insert(0, ROOT)
            import bench_legs
            order = bench_legs.hilbert_order(N, torch)
        keep.append(order)
        jobs = [(out(N, 3, f32), np.float32, 3, pos, 0, order, False),
                (out(N, 3, f32), np.float32, 3, vel, 0, order, False)]
        algo, moved = (24 + 8 + 24) * N, (32 + 8 + 24) * N
    elif workload == "copy4":             # plain float4 copy: the chip's streaming reference point
        pos = rnd((N, 4), f32)
        jobs = [(out(N, 4, f32), np.float32, 4, pos, 0, None, False)]
        algo, moved = 32 * N, 32 * N
    elif workload == "sph_full":          # 112 B/particle PGSD-SPH schema from HOOMD-style arrays
        pos, vel = rnd((N, 4), f32), rnd((N, 4), f32)
        dpe = rnd((N, 4), f32)            # density, pressure, energy, slength packed as a float4
        aux = [rnd((N, 4), f32) for _ in range(4)]
        img = torch.randint(-2, 3, (N, 4), generator=gen, device="cuda", dtype=i32); keep.append(img)
        body = torch.randint(0, 9, (N, 1), generator=gen, device="cuda", dtype=i32); keep.append(body)
        jobs = [(out(N, 3, f32), np.float32, 3, pos, 0, None, False), (out(N, 1, i32), np.uint32, 1, pos, 3, None, True),
                (out(N, 3, f32), np.float32, 3, vel, 0, None, False), (out(N, 1, f32), np.float32, 1, vel, 3, None, False),
                (out(N, 1, f32), np.float32, 1, dpe, 0, None, False), (out(N, 1, f32), np.float32, 1, dpe, 1, None, False),
                (out(N, 1, f32), np.float32, 1, dpe, 2, None, False), (out(N, 1, f32), np.float32, 1, dpe, 3, None, False),
                (out(N, 3, i32), np.int32, 3, img, 0, None, False), (out(N, 1, i32), np.int32, 1, body, 0, None, False)]
        jobs += [(out(N, 3, f32), np.float32, 3, a, 0, None, False) for a in aux]
        algo, moved = 224 * N, (7 * 16 + 16 + 4 + 112) * N
    else:
        raise SystemExit("unknown workload " + workload)
    return jobs, algo, moved, keep


KERNEL_ONLY = False
AB_KEYS = ("PGSD_PACK_KERNEL", "PGSD_PACK_ROWS_CFG", "PGSD_PACK_TILE", "PGSD_PACK_BLOCKS_PER_CU", "PGSD_PACK_PREFETCH")


def to_c(jobs):
    arr = (_lib.PackJob * len(jobs))()
    for i, (dst, out_dt, M, src, col0, order, bitcast) in enumerate(jobs):
        arr[i].dst = dst.data_ptr()
        arr[i].dst_type = G.type_id(out_dt)
        arr[i].M = M
        arr[i].src.src = src.data_ptr()
        arr[i].src.order = order.data_ptr() if order is not None else None
        arr[i].src.src_type = G.type_id(str(src.dtype)[6:])
        arr[i].src.src_stride = src.shape[1]
        arr[i].src.src_col0 = col0
        arr[i].src.bitcast = 1 if bitcast else 0
    return arr


def run(workload, N, iters, warmup, sleep_ms=0.0, variants=None):
    gen = torch.Generator(device="cuda").manual_seed(1234)
    probe, algo, moved, _ = make_jobs(workload, 1024, gen)
    per_set = moved / 1024 * N
    n_sets = max(2, int(np.ceil(600e6 / per_set)) + 1)   # > 256 MiB touched between two uses of a set
    sets = []
    for _ in range(n_sets):
        jobs, algo, moved, keep = make_jobs(workload, N, gen)
        sets.append((to_c(jobs), len(jobs), keep))
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(warmup):
        arr, n, _ = sets[i % n_sets]
        assert _lib.lib.pgsd_pack_fields(n, arr, N, ctypes.c_void_p(stream), None) == 0, _lib.last_error()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    import time
    kernel_ms = []
    for i in range(iters):
        arr, n, _ = sets[i % n_sets]
        if sleep_ms > 0:
            torch.cuda.synchronize()
            time.sleep(sleep_ms * 1e-3)   # let the GPU idle between launches, like a snapshot every few ms
        if variants:
            # interleaved A/B in one process: "K=V+K2=V2" sets those PGSD_PACK_* variables for this launch
            v = variants[i % len(variants)]
            for k in AB_KEYS:
                os.environ.pop(k, None)
            for kv in v.split("+"):
                k, _, val = kv.partition("=")
                os.environ[k] = val
            _lib.lib.pgsd_reload_tuning()     # the library reads its tuning variables once (csrc/pgsd_private.h)
        evs[i][0].record()
        if KERNEL_ONLY:
            # the dispatches' own begin / end stamps (what rocprofv3 reports): no launch latency in the figure
            ms = ctypes.c_float(0)
            assert _lib.lib.pgsd_pack_fields(n, arr, N, ctypes.c_void_p(stream), ctypes.byref(ms)) == 0
            kernel_ms.append(ms.value)
        else:
            _lib.lib.pgsd_pack_fields(n, arr, N, ctypes.c_void_p(stream), None)
        evs[i][1].record()
    torch.cuda.synchronize()
    ts = np.array(kernel_ms if KERNEL_ONLY else [a.elapsed_time(b) for a, b in evs]) * 1e-3
    if variants:
        out = {"workload": workload, "N": N, "interleaved": True}
        for k, v in enumerate(variants):
            tv = ts[k::len(variants)]
            out["var" + v] = {"median_us": round(float(np.median(tv)) * 1e6, 2), "min_us": round(float(tv.min()) * 1e6, 2)}
        return out
    med, mn = float(np.median(ts)), float(ts.min())
    return {"workload": workload, "N": N, "sets": n_sets, "median_us": round(med * 1e6, 2), "min_us": round(mn * 1e6, 2),
            "algo_GBps": round(algo / med / 1e9, 1), "moved_GBps": round(moved / med / 1e9, 1),
            "frac_algo": round(algo / med / 8e12, 4), "timing": "kernel" if KERNEL_ONLY else "stream events", "env": {k: v for k, v in os.environ.items() if k.startswith("PGSD_PACK")}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="pos_vel_id,hoomd_w,double4,copy4")
    ap.add_argument("--N", type=int, default=10_000_000)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sleep-ms", type=float, default=0.0)
    ap.add_argument("--variants", default="", help="comma list of variants to interleave launch by launch: "
                    "KEY=VALUE[+KEY=VALUE] settings of the PGSD_PACK_* variables")
    ap.add_argument("--kernel-only", action="store_true",
                    help="time with the dispatches' own stamps (pgsd_pack_fields' kernel_ms) instead of stream events "
                         "around the call")
    a = ap.parse_args()
    KERNEL_ONLY = a.kernel_only
    for w in a.workloads.split(","):
        r = run(w, a.N, a.iters, a.warmup, a.sleep_ms, [v for v in a.variants.split(',') if v] or None)
        r['sleep_ms'] = a.sleep_ms
        print(json.dumps(r), flush=True)
