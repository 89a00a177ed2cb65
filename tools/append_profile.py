#!/usr/bin/env python3
"""Where the time of `HOOMDTrajectory.append` goes for device-resident frames (N = 1024^2 by default)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import torch
import pgsd.hoomd as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 1024
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
path = "/dev/shm/pgsd_append_profile_%d.gsd" % os.getpid()
pos = torch.rand((N, 3), device="cuda")
ori = torch.rand((N, 4), device="cuda")


def frame(i):
    f = H.Frame()
    f.particles.N = N
    f.configuration.step = i
    f.particles.position = pos
    f.particles.orientation = ori
    return f


with H.open(path, "w") as t:
    t.device_elision = False       # the same tensors every frame: with the comparison on, frames 1.. would be elided
    for i in range(5):
        t.append(frame(i))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        t.append(frame(i))
    dt = time.perf_counter() - t0
    print("append: %.2f ms/frame, %.2f GB/s" % (dt / frames * 1e3, frames * N * 28 / dt / 1e9))
    st = t.file.device_stats()
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
    t0 = time.perf_counter()
    for i in range(frames):
        t.append(frame(i), wait=False)
    t.file.frame_sync()
    dt = time.perf_counter() - t0
    print("append(wait=False): %.2f ms/frame, %.2f GB/s" % (dt / frames * 1e3, frames * N * 28 / dt / 1e9))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(frames):
        t.append(frame(i))
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
os.unlink(path)
