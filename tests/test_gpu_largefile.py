"""Chunks and files beyond 2 GiB (the reference's test/test_largefile.py:13-42: uint32 ramps of 2^27, 2^28 and
2^29+1 elements, written as one chunk and read back).  A single pwrite/pread moves at most 0x7ffff000 bytes, a
32-bit byte count wraps at 4 GiB, MPI-IO counts are ints: this is where such limits show.  Run on the GPU box
(fast /dev/shm, plenty of memory); the host path and the device path write the same file."""
import gc
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import pgsd.fl as fl


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 24), b""):
            h.update(block)
    return h.hexdigest()


@pytest.mark.parametrize("N", [2 ** 27, 2 ** 29 + 1, 2 ** 30 + 3])     # 512 MiB, 2 GiB + 4 B, 4 GiB + 12 B
def test_large_n(N):
    import torch
    gc.collect()
    host_path = "/dev/shm/pgsd_large_host_%d.gsd" % os.getpid()
    dev_path = "/dev/shm/pgsd_large_dev_%d.gsd" % os.getpid()
    try:
        data = np.arange(N, dtype=np.uint32)
        with fl.open(host_path, 'x', application='test_large_N', schema='none', schema_version=[1, 2]) as f:
            f.write_chunk(name='data', data=data)
            f.end_frame()
        assert os.path.getsize(host_path) == 5376 + 4 * N      # header + index + names (5376), then the chunk
        with fl.open(host_path, 'r') as f:
            got = f.read_chunk(frame=0, name='data')
            assert got.dtype == np.uint32 and got.shape == (N,)
            assert np.array_equal(got, data)
            del got
            tail = f.read_chunk(0, 'data', N=3, M=1, offset=N - 3, r_all=True)[:3]   # a row slab past 2 GiB
            np.testing.assert_array_equal(tail, [N - 3, N - 2, N - 1])
        del data
        gc.collect()

        # the same chunk out of HBM, and back into it
        dev = torch.arange(N, dtype=torch.int32, device="cuda")
        with fl.open(dev_path, 'x', application='test_large_N', schema='none', schema_version=[1, 2]) as f:
            f.write_chunk('data', fl.DeviceField.from_tensor(dev, out_dtype=np.uint32, bitcast=True),
                          offset=np.array([N]))
            f.end_frame()
            back = f.read_chunk_device(0, 'data')
            torch.cuda.synchronize()
            assert back.shape[0] == N
            assert bool(torch.equal(back.view(torch.int32).reshape(-1), dev))
            del back
        del dev
        torch.cuda.empty_cache()
        assert _sha(dev_path) == _sha(host_path)
    finally:
        for p in (host_path, dev_path):
            if os.path.exists(p):
                os.unlink(p)
