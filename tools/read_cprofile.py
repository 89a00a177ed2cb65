#!/usr/bin/env python3
"""cProfile of `HOOMDTrajectory.read_frame_device` / `trajectory[i]` for small frames: N particles (argv 1),
`host` as argv 2 reads through the host path."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import numpy
import torch
import pgsd.hoomd as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
host = len(sys.argv) > 2 and sys.argv[2] == "host"
path = "/dev/shm/pgsd_read_cprofile_%d.gsd" % os.getpid()
with H.open(path, "w") as t:
    for i in range(200):
        f = H.Frame()
        f.particles.N = N
        f.configuration.step = i
        f.particles.position = numpy.random.random((N, 3)).astype("float32")
        f.particles.orientation = numpy.random.random((N, 4)).astype("float32")
        t.append(f)
with H.open(path, "r") as t:
    read = (lambda i: t[i]) if host else (lambda i: t.read_frame_device(i))
    for i in range(50):
        read(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(5):
        for i in range(200):
            read(i)
    torch.cuda.synchronize()
    print("== %s read, N=%d: %.1f us/frame" % ("host" if host else "device", N, (time.perf_counter() - t0) / 1000 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for rep in range(5):
        for i in range(200):
            read(i)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
os.unlink(path)
