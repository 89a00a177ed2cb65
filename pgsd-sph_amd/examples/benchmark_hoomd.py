#!/usr/bin/env python3
"""Schema-level read/write benchmark of `pgsd.hoomd` -- the counterpart of the reference's
pgsd/scripts/benchmark-hoomd.py (which cannot run against the reference: its `HOOMDTrajectory.append` is a
sketch, hoomd.py:568).

Same protocol: for N = 32^2, 128^2 and 1024^2 particles write frames of position (N x 3 f32) + orientation
(N x 4 f32) until the file holds `--size` MiB, then time opening the file, reading frames in order and reading
them in random order (at most 256 MiB each).  Differences: no `sudo sysctl vm.drop_caches` (the job has no
root; reads come from the page cache) and `--mode`:

    host          arrays in host memory (the reference's own protocol)
    hbm           arrays in HBM, written by the fused pack kernel, read back with `read_frame_device` (= --device)
    hbm-via-host  arrays in HBM, copied to host arrays by the caller every frame (`tensor.cpu()`) and written through
                  the host path: what a GPU-resident simulation pays WITHOUT the device path
    hbm-async     arrays in HBM, frames sealed with `append(frame, wait=False)`: the simulation waits for the pack
                  kernels only; the file is complete at the final `frame_sync()` (inside the timed region)

    python pgsd-sph_amd/examples/benchmark_hoomd.py [--size MiB] [--mode host|hbm|hbm-via-host|hbm-async] [--dir /dev/shm]
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


def run(N, size, path, mode):
    device = mode != "host"
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
            # the SAME two tensors are appended every frame: with the GPU-side elision test on (the default) frames
            # 1.. would equal frame 0 and hold no particle data.  A simulation's arrays move; this measures the write
            hf.device_elision = False
            for i in range(nframes):
                if mode == "host":
                    position[0, 0] = i          # every frame differs from frame 0 (benchmark-hoomd.py:27-28):
                    orientation[0, 0] = i       # nothing is elided.  For the arrays in HBM the comparison is switched
                                                # off above instead: a simulation changes them with kernels of its own,
                                                # not with a per-frame torch scalar store from the host (~20 us each)
                    hf.append(make_frame(i, position, orientation))
                elif mode == "hbm-via-host":
                    pos_h, ori_h = position.cpu().numpy(), orientation.cpu().numpy()
                    pos_h[0, 0] = i
                    ori_h[0, 0] = i
                    hf.append(make_frame(i, pos_h, ori_h))
                else:
                    hf.append(make_frame(i, position, orientation), wait=(mode != "hbm-async"))
            if mode == "hbm-async":
                hf.file.frame_sync()

    write_all()                                 # warm the target
    os.unlink(path)                             # ... and keep its truncation out of the timed pass: giving a 1 GiB tmpfs
                                                # file's pages back takes 60-100 ms, a third of the 1024^2 pass
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
        if mode == "hbm-via-host":
            want[0, 0] = nframes - 1
        assert last.configuration.step == (nframes - 1) * 10
        assert numpy.array_equal(last.particles.position, want)
    os.unlink(path)
    out['nframes'] = nframes
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=float, default=128, help="file size in MiB")
    ap.add_argument("--device", action="store_true", help="= --mode hbm")
    ap.add_argument("--mode", choices=["host", "hbm", "hbm-via-host", "hbm-async"], default=None)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--sizes", default="32,128,1024", help="sqrt(N) values")
    args = ap.parse_args()
    mode = args.mode or ("hbm" if args.device else "host")
    path = os.path.join(args.dir, "pgsd_benchmark_hoomd_%d.gsd" % os.getpid())
    print("mode %s: arrays in %s, %g MiB files in %s" % (mode, "host memory" if mode == "host" else "HBM", args.size, args.dir))
    print("{:<8} {:<8} {:<10} {:<12} {:<14} {:<12} {:<14} {:<12}".format(
        "N", "frames", "open (ms)", "write (MB/s)", "write (us/frm)", "seq read", "random read", "random (ms)"))
    for root in [int(v) for v in args.sizes.split(",")]:
        r = run(root * root, args.size * 1024 ** 2, path, mode)
        print("{:<8} {:<8} {:<10.3g} {:<12.4g} {:<14.4g} {:<12.4g} {:<14.4g} {:<12.3g}".format(
            "%d^2" % root, r['nframes'], r['open_ms'], r['write'], r['write_us_per_frame'], r['seq_read'],
            r['random_read'], r['random_read_ms']), flush=True)


if __name__ == "__main__":
    main()
