#!/usr/bin/env python3
"""A short run of `HOOMDTrajectory.append` for the profiler: N particles (argv 1), K frames (argv 2), arrays in
HBM (default) or host memory (argv 3 = host).  Run under
    PGSD_TRACE=1 rocprofv3 --hip-trace --kernel-trace --marker-trace --memory-copy-trace --stats -d <dir> -- python3 tools/append_trace.py 1024 200
the library's roctx ranges (pgsd:*) then sit next to the HIP calls, kernels and copies they bracket."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import numpy
import torch
import pgsd.hoomd as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
host = len(sys.argv) > 3 and sys.argv[3] == "host"
path = "/dev/shm/pgsd_append_trace_%d.gsd" % os.getpid()
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
    for i in range(20):
        t.append(frame(i))
    t0 = time.perf_counter()
    for i in range(frames):
        t.append(frame(20 + i))
    dt = time.perf_counter() - t0
    print("append (%s arrays, N=%d): %.1f us/frame" % ("host" if host else "HBM", N, dt / frames * 1e6))
os.unlink(path)
