#!/usr/bin/env python3
"""What does a draining snapshot cost a kernel that is running?  (north_star: "overlapped with the next chunk's pack";
the reference has no counterpart -- its data is host resident.)

    python tools/overlap_probe.py [--particles 10000000] [--frames 24] [--dir /dev/shm]

A simulation seals a frame asynchronously (`end_frame(wait=False)`), waits for the pack kernels only and goes on
computing while the device->host copies (shader blits, `__amd_rocclr_copyBuffer`, on these boxes) and the `pwrite`s
run behind it.  Here the "simulation" is a queue of identical kernels on a stream of its own, enqueued up front:

  * `stream`: an HBM-bound kernel (copy of a 1 GiB tensor: 2 GiB of traffic per launch, every CU busy), and
  * `fma`:    an fp32 GEMM (8192^3, the matrix cores / vector ALUs busy, little HBM traffic),

each timed over the same number of launches (HIP events around the whole queue) ALONE and WHILE the main thread appends
`--frames` 10 M-particle frames sealed asynchronously and waits for the file (`frame_sync`).  Printed per kernel kind:
ms per launch alone / under the drain, the slowdown in percent, and how much of the queue's run time the drain covered.
One JSON line per kind.  Run it under `rocprofv3 --kernel-trace --stats` for the trace of the same run."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=10_000_000)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--kinds", default="stream,fma")
    ap.add_argument("--label", default="")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    N = args.particles
    g = torch.Generator(device="cuda").manual_seed(7)
    pos = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 100.0
    vel = torch.randn((N, 4), generator=g, device="cuda")
    fields = [("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
              ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
              ("particles/typeid", fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True))]
    path = os.path.join(args.dir, "pgsd_overlap_%d.gsd" % os.getpid())

    src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()        # 1 GiB
    dst = torch.empty_like(src)
    a = torch.randn((8192, 8192), device="cuda")
    b = torch.randn((8192, 8192), device="cuda")
    c = torch.empty_like(a)
    kernels = {"stream": (lambda: dst.copy_(src), "copy of 1 GiB (2 GiB of HBM traffic per launch)"),
               "fma": (lambda: torch.mm(a, b, out=c), "fp32 GEMM 8192^3")}
    side = torch.cuda.Stream()

    def drain(n_frames):
        """n frames, each sealed asynchronously; returns (seconds until the file has them, max wait for a pack)."""
        f = fl.open(path, "w", application="overlap probe", schema="hoomd", schema_version=[1, 4])
        f.frame_exchange = True
        t0 = time.perf_counter()
        worst = 0.0
        for i in range(n_frames):
            f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
            f.write_chunks(fields, offset="auto")
            f.end_frame(wait=False)
            t1 = time.perf_counter()
            f.wait_packed()                                  # what the simulation is blocked for
            worst = max(worst, time.perf_counter() - t1)
        f.frame_sync()
        dt = time.perf_counter() - t0
        stats = f.device_stats()
        f.close()
        os.unlink(path)
        return dt, worst, stats

    drain(3)                                                 # pipelines, pinned slabs, page cache warm
    t_drain, _, _ = drain(args.frames)
    for kind in args.kinds.split(","):
        fn, what = kernels[kind]
        with torch.cuda.stream(side):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
        side.synchronize()
        per = e0.elapsed_time(e1) / 20.0
        launches = max(20, int(t_drain * 1e3 * 0.9 / per))   # a queue about as long as the drain

        def run_queue():
            with torch.cuda.stream(side):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(launches):
                    fn()
                e.record()
            return s, e

        alone = []
        for _ in range(3):
            s, e = run_queue()
            side.synchronize()
            alone.append(s.elapsed_time(e) / launches)
        under, cover, stalls = [], [], []
        for _ in range(3):
            s, e = run_queue()
            dt, worst, stats = drain(args.frames)
            side.synchronize()
            ms = s.elapsed_time(e)
            under.append(ms / launches)
            cover.append(min(1.0, dt * 1e3 / ms))
            stalls.append(worst * 1e3)
        a_ms, u_ms = min(alone), min(under)
        print(json.dumps({"kernel": kind, "what": what, "label": args.label, "launches": launches,
                          "ms_per_launch_alone": round(a_ms, 4), "ms_per_launch_under_drain": round(u_ms, 4),
                          "slowdown_pct": round((u_ms / a_ms - 1.0) * 100.0, 2),
                          "all_alone_ms": [round(x, 4) for x in alone], "all_under_ms": [round(x, 4) for x in under],
                          "drain_covers_fraction_of_queue": round(min(cover), 3),
                          "frames": args.frames, "particles": N, "frame_MB": N * 28 / 1e6,
                          "drain_s_alone": round(t_drain, 3), "max_wait_for_pack_ms": round(max(stalls), 3)}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
