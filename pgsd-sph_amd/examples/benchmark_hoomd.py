#!/usr/bin/env python3
"""Schema-level read/write benchmark of `pgsd.hoomd` -- the counterpart of the reference's
pgsd/scripts/benchmark-hoomd.py (which cannot run against the reference: its `HOOMDTrajectory.append` is a
sketch, hoomd.py:568).

Same protocol: for N = 32^2, 128^2 and 1024^2 particles write frames of position (N x 3 f32) + orientation
(N x 4 f32) until the file holds `--size` MiB, then time opening the file, reading frames in order and reading
them in random order (at most 256 MiB each).  Differences: no `sudo sysctl vm.drop_caches` (the job has no
root; reads come from the page cache) and `--device` keeps the arrays in HBM: frames are then written by the
fused pack kernel and read back with `read_frame_device`.

    python pgsd-sph_amd/examples/benchmark_hoomd.py [--size MiB] [--device] [--dir /dev/shm]
"""
import argparse
import math
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy

import pgsd.hoomd

BYTES_PER_PARTICLE = (3 + 4) * 4


def make_frame(step, position, orientation):
    frame = pgsd.hoomd.Frame()
    frame.particles.N = position.shape[0]
    frame.configuration.step = step * 10
    frame.particles.position = position
    frame.particles.orientation = orientation
    return frame


def run(N, size, path, device):
    read_budget = 256 * 1024 ** 2
    nframes = max(2, int(math.ceil(size / (BYTES_PER_PARTICLE * N))))
    nframes_read = max(1, min(nframes, int(read_budget / (BYTES_PER_PARTICLE * N))))
    position = numpy.random.random((N, 3)).astype('float32')
    orientation = numpy.random.random((N, 4)).astype('float32')
    if device:
        import torch
        position = torch.from_numpy(position).cuda()
        orientation = torch.from_numpy(orientation).cuda()

    def write_all():
        with pgsd.hoomd.open(name=path, mode='w') as hf:
            for i in range(nframes):
                if not device:
                    position[0, 0] = i          # every frame differs from frame 0 (benchmark-hoomd.py:27-28):
                    orientation[0, 0] = i       # nothing is elided.  Arrays in HBM are never compared, hence never
                                                # elided, and a simulation changes them with kernels of its own, not
                                                # with a per-frame torch scalar store from the host (~20 us each)
                hf.append(make_frame(i, position, orientation))

    write_all()                                 # warm the target
    t0 = time.perf_counter()
    write_all()
    t_write = time.perf_counter() - t0

    out = {}
    out['write'] = nframes * BYTES_PER_PARTICLE * N / 1024 ** 2 / t_write
    out['write_us_per_frame'] = t_write / nframes * 1e6
    t0 = time.perf_counter()
    with pgsd.hoomd.open(name=path, mode='r') as hf:
        out['open_ms'] = (time.perf_counter() - t0) * 1e3
        assert len(hf) == nframes

        def read(idx):
            if device:
                fr = hf.read_frame_device(idx)
                return fr
            return hf[idx]

        t0 = time.perf_counter()
        for i in range(nframes_read):
            read(i)
        if device:
            torch.cuda.synchronize()
        out['seq_read'] = nframes_read * BYTES_PER_PARTICLE * N / 1024 ** 2 / (time.perf_counter() - t0)
        frames = list(range(nframes))
        random.Random(7).shuffle(frames)
        t0 = time.perf_counter()
        for f in frames[:nframes_read]:
            read(f)
        if device:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out['random_read'] = nframes_read * BYTES_PER_PARTICLE * N / 1024 ** 2 / dt
        out['random_read_ms'] = dt / nframes_read * 1e3
        last = hf[nframes - 1]
        want = position.cpu().numpy() if device else position
        assert last.configuration.step == (nframes - 1) * 10
        assert numpy.array_equal(last.particles.position, want)
    os.unlink(path)
    out['nframes'] = nframes
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=float, default=128, help="file size in MiB")
    ap.add_argument("--device", action="store_true", help="arrays in HBM (needs an MI355X)")
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--sizes", default="32,128,1024", help="sqrt(N) values")
    args = ap.parse_args()
    path = os.path.join(args.dir, "pgsd_benchmark_hoomd_%d.gsd" % os.getpid())
    print("arrays in %s, %g MiB files in %s" % ("HBM" if args.device else "host memory", args.size, args.dir))
    print("{:<8} {:<8} {:<10} {:<12} {:<14} {:<12} {:<14} {:<12}".format(
        "N", "frames", "open (ms)", "write (MB/s)", "write (us/frm)", "seq read", "random read", "random (ms)"))
    for root in [int(v) for v in args.sizes.split(",")]:
        r = run(root * root, args.size * 1024 ** 2, path, args.device)
        print("{:<8} {:<8} {:<10.3g} {:<12.4g} {:<14.4g} {:<12.4g} {:<14.4g} {:<12.3g}".format(
            "%d^2" % root, r['nframes'], r['open_ms'], r['write'], r['write_us_per_frame'], r['seq_read'],
            r['random_read'], r['random_read_ms']), flush=True)


if __name__ == "__main__":
    main()
