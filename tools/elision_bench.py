#!/usr/bin/env python3
"""GPU-side elision test (pgsd_compare_staged_chunks) at BASELINE size: what the comparison costs and what it saves.

  python tools/elision_bench.py [--N 10000000] [--dir /dev/shm] [--json out.jsonl]

1. kernel: the packed chunks of a 10 M-particle frame (position, velocity, type id, mass, body, smoothing length:
   40 B/particle) against frame 0's rows in HBM -- all equal (the whole of both sides is read: 2 x 400 MB), and with
   position + velocity changed (those two stop at their first stride);
2. through pgsd.hoomd: frames whose position / velocity move while type id, mass, body and smoothing length stay,
   appended with device_elision = True (default: every array compared in every frame), (round 3/4 also: 'once', an array that differed
   is not compared again) and False.  --N 1024 --frames 400: what the modes cost a small frame."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pgsd-sph_amd"))

import numpy as np
import torch

import pgsd.fl as fl
import pgsd.hoomd as hoomd

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=10_000_000)
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--frames", type=int, default=6)
ap.add_argument("--json", default=None)
ap.add_argument("--no-append", action="store_true", help="part 1 only (for profiler runs)")
args = ap.parse_args()
N = args.N
g = torch.Generator(device="cuda").manual_seed(1)
pos4 = torch.rand((N, 4), generator=g, device="cuda")
pos4[:, 3] = torch.randint(0, 4, (N,), generator=g, device="cuda", dtype=torch.int32).view(torch.float32)
vel4 = torch.randn((N, 4), generator=g, device="cuda")
vel4[:, 3] = 1.5
body = torch.full((N,), -1, device="cuda", dtype=torch.int32)
slen = torch.rand((N,), generator=g, device="cuda")


def fields():
    return [("particles/typeid", fl.DeviceField.from_tensor(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True)),
            ("particles/mass", fl.DeviceField.from_tensor(vel4, columns=(3, 4))),
            ("particles/body", fl.DeviceField.from_tensor(body)),
            ("particles/position", fl.DeviceField.from_tensor(pos4, columns=(0, 3))),
            ("particles/velocity", fl.DeviceField.from_tensor(vel4, columns=(0, 3))),
            ("particles/slength", fl.DeviceField.from_tensor(slen))]


out = {"N": N}
path = os.path.join(args.dir, "pgsd_elision_bench_%d.gsd" % os.getpid())
sizes = [4 * N, 4 * N, 4 * N, 12 * N, 12 * N, 4 * N]

# ---- 1. the comparison alone
with fl.open(path, "w", application="x", schema="hoomd", schema_version=[1, 4]) as f:
    f.frame_exchange = True
    t = f.stage_chunks(fields())
    refs = f.copy_staged(t, 0, sizes)
    f.write_staged(t, 0, 6, offset=np.array([N]))
    f.end_frame()
    for label, change in (("all_equal", False), ("position_velocity_changed", True)):
        if change:
            pos4[:, :3] += 1.0
            vel4[:, :3] += 1.0
        best = 1e9
        for rep in range(5):
            t = f.stage_chunks(fields())
            f.wait_packed()
            t0 = time.perf_counter()
            eq = f.compare_staged(t, 0, refs)
            best = min(best, time.perf_counter() - t0)
            f.end_frame()                       # drops the staged chunks: nothing is written
        assert eq == ([True] * 6 if not change else [True, True, True, False, False, True]), eq
        read = 2 * sum(s for s, e in zip(sizes, eq) if e)
        out[label] = {"call_us": round(best * 1e6, 1), "bytes_read_if_equal": read,
                      "GBps_of_call": round(read / best / 1e9, 1)}
        print("compare_staged %-28s %8.1f us per call (launch + wait), %6.1f GB/s of the equal arrays' 2 x bytes"
              % (label, best * 1e6, read / best / 1e9), flush=True)
del refs
os.unlink(path)


# ---- 2. pgsd.hoomd.append, moving position / velocity, static type id / mass / body / smoothing length
def frame(step):
    fr = hoomd.Frame()
    fr.configuration.step = step
    fr.particles.N = N
    for (name, field) in fields():
        setattr(fr.particles, name.split("/")[1], field)
    return fr


for elide in (() if args.no_append else (True, False, True, False)):
    with hoomd.open(path, "w") as t:
        t.device_elision = elide
        t.append(frame(0))
        times = []
        for k in range(1, args.frames):
            pos4[:, :3] += 0.5
            vel4[:, :3] *= 1.01
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t.append(frame(k))
            times.append(time.perf_counter() - t0)
    size = os.path.getsize(path)
    os.unlink(path)
    name = {True: "every frame (default)", False: "off"}[elide]
    key = "append_elision_%s" % name.split()[0]
    mean_us = 1e6 * sum(times) / max(len(times), 1)
    out.setdefault(key, []).append({"ms_per_frame": [round(x * 1e3, 2) for x in times[:12]], "mean_us": round(mean_us, 1),
                                    "file_MB": round(size / 1e6, 1)})
    print("append, comparison %-22s: mean %.1f us per frame over %d frames (first: %s ms), file %.0f MB"
          % (name, mean_us, len(times), " ".join("%.2f" % (x * 1e3) for x in times[:6]), size / 1e6), flush=True)
if args.json:
    with open(args.json, "a") as fh:
        fh.write(json.dumps(out) + "\n")
