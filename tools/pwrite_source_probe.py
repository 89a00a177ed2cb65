#!/usr/bin/env python3
"""Why does the host path's pwrite into a tmpfs file run at 11-15 GB/s when the pipeline's writer thread gets 6-7?
One thread, one growing file, 8 MiB pwrite pieces, different SOURCE buffers:
  numpy-small   29 MB numpy array, rewritten every round (what benchmark_hoomd's host rows do)
  numpy-big     560 MB numpy array (larger than the L3: cold lines)
  pinned-cpu    29 MB pinned (hipHostMalloc) buffer, filled by the CPU before every round
  pinned-d2h    29 MB pinned buffer, filled by a device->host copy before every round (what the writer thread sees)
  pinned-d2h-touched   the same, every cache line read once by this thread before the pwrite
"""
import os
import sys
import time

import numpy as np
import torch

PIECE = 8 << 20
path = "/dev/shm/pgsd_pwrite_probe_%d" % os.getpid()


def run(label, make_src, nbytes, rounds):
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    off = 0
    t_write = 0.0
    for r in range(rounds):
        mv = make_src(r)
        t0 = time.perf_counter()
        for o in range(0, nbytes, PIECE):
            os.pwrite(fd, mv[o:o + PIECE], off + o)
        t_write += time.perf_counter() - t0
        off += nbytes
    os.close(fd)
    os.unlink(path)
    print("%-22s %6.2f GB/s  (%d x %.0f MB)" % (label, rounds * nbytes / t_write / 1e9, rounds, nbytes / 1e6), flush=True)


n_small = 29_360_128
small = np.random.randint(0, 255, n_small, dtype=np.uint8)
big = np.random.randint(0, 255, 560_000_000, dtype=np.uint8)
dev = torch.randint(0, 255, (n_small,), dtype=torch.uint8, device="cuda")
pinned = torch.empty((n_small,), dtype=torch.uint8, pin_memory=True)
pin_np = pinned.numpy()


def src_small(r):
    small[0] = r
    return memoryview(small)


def src_big(r):
    big[0] = r
    return memoryview(big)


def src_pinned_cpu(r):
    pin_np[:] = small
    return memoryview(pin_np)


def src_pinned_d2h(r):
    pinned.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
    return memoryview(pin_np)


def src_pinned_d2h_touched(r):
    pinned.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
    pin_np[::64].sum()
    return memoryview(pin_np)


for rep in range(2):
    run("numpy-small", src_small, n_small, 100)
    run("numpy-big", src_big, big.size, 6)
    run("pinned-cpu", src_pinned_cpu, n_small, 100)
    run("pinned-d2h", src_pinned_d2h, n_small, 100)
    run("pinned-d2h-touched", src_pinned_d2h_touched, n_small, 100)
