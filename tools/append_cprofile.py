#!/usr/bin/env python3
"""cProfile of `HOOMDTrajectory.append`: N particles (argv 1), arrays in HBM or host memory (argv 2 = host)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import numpy
import torch
import pgsd.hoomd as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
host = len(sys.argv) > 2 and sys.argv[2] == "host"
path = "/dev/shm/pgsd_append_cprofile_%d.gsd" % os.getpid()
pos = numpy.random.random((N, 3)).astype("float32")
ori = numpy.random.random((N, 4)).astype("float32")
if not host:
    pos, ori = torch.from_numpy(pos).cuda(), torch.from_numpy(ori).cuda()


def frame(i):
    f = H.Frame()
    f.particles.N = N
    f.configuration.step = i
    if host:
        pos[0, 0] = i       # both arrays differ from frame 0 every frame: nothing is elided (for device arrays the comparison is switched off below)
        ori[0, 0] = i
    f.particles.position = pos
    f.particles.orientation = ori
    return f


with H.open(path, "w") as t:
    t.device_elision = False       # the same tensors every frame: with the comparison on, frames 1.. would be elided
    for i in range(50):
        t.append(frame(i))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(2000):
        t.append(frame(50 + i))
    pr.disable()
    print("== %s arrays, N=%d, 2000 frames" % ("host" if host else "HBM", N))
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
os.unlink(path)
