#!/usr/bin/env python3
"""Where the time of a small device read goes: `read_frame_device` on 1024-particle frames, profiled."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import torch
import pgsd.hoomd as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
path = "/dev/shm/pgsd_small_reads_%d.gsd" % os.getpid()
pos = torch.rand((N, 3), device="cuda")
ori = torch.rand((N, 4), device="cuda")
with H.open(path, "w") as t:
    for i in range(300):
        f = H.Frame()
        f.particles.N = N
        f.configuration.step = i
        pos[0, 0] = i
        f.particles.position = pos
        f.particles.orientation = ori
        t.append(f)
with H.open(path, "r") as t:
    for i in range(50):
        t.read_frame_device(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50, 300):
        t.read_frame_device(i)
    torch.cuda.synchronize()
    print("read_frame_device: %.0f us/frame" % ((time.perf_counter() - t0) / 250 * 1e6))
    f = t.file
    t0 = time.perf_counter()
    for i in range(50, 300):
        f.read_chunk_device(i, "particles/position")
    torch.cuda.synchronize()
    print("fl.read_chunk_device (one chunk, wait): %.0f us/call" % ((time.perf_counter() - t0) / 250 * 1e6))
    t0 = time.perf_counter()
    for i in range(50, 300):
        f.read_chunk(i, "particles/position")
    print("fl.read_chunk (host): %.0f us/call" % ((time.perf_counter() - t0) / 250 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(50, 300):
        t.read_frame_device(i)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
os.unlink(path)
