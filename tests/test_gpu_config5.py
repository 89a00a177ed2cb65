"""BASELINE config 5 at its own size: an 80 M-particle position + velocity + typeid file (2.24 GB per frame).

Written by the device path from float4 arrays in HBM, byte-identical to the file the CPU oracle writes from
host copies of the same values; then one rank's share -- rows [35 M, 45 M), the 10 M particles a GPU of an
8-GPU run would own -- is read back through the device read path (pread -> pinned ring -> HBM -> HIP unpack)
into HOOMD-style Scalar4 arrays and compared with the slices the pure-Python reader (`pgsd.pypgsd`, the
reference's reader restated) returns for the same file.  Bit-exact throughout."""
import ctypes
import os

import numpy as np
import pytest

import scenario as S

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

N = 80_000_000
ROW0, ROWS = 35_000_000, 10_000_000


def _same_bytes(a, b):
    if os.path.getsize(a) != os.path.getsize(b):
        return False
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            x, y = fa.read(1 << 26), fb.read(1 << 26)
            if x != y:
                return False
            if not x:
                return True


def test_80M_particle_file_written_on_device_equals_oracle_and_reads_back_by_partition():
    import pgsd.fl as fl
    import pgsd.pypgsd as pypgsd
    d = "/dev/shm"
    mine = os.path.join(d, "pgsd_c5_mine_%d.gsd" % os.getpid())
    ref = os.path.join(d, "pgsd_c5_ref_%d.gsd" % os.getpid())
    try:
        # every 32-bit word of the sources distinct (a misplaced row, column or byte cannot go unnoticed), cheap
        # to make at this size: pos words = 4i .. 4i+3 as bit patterns, vel words offset by 2^30 and scrambled
        pos = torch.empty((N, 4), dtype=torch.float32, device="cuda")
        pos.view(torch.int32).copy_(torch.arange(4 * N, dtype=torch.int32, device="cuda").view(N, 4))
        vel = torch.empty((N, 4), dtype=torch.float32, device="cuda")
        vel.view(torch.int32).copy_((torch.arange(4 * N, dtype=torch.int32, device="cuda") * 3 + (1 << 30)).view(N, 4))
        # keep the bit patterns away from NaN payload canonicalisation worries: they are only ever moved as bits
        with fl.open(mine, "w", application="app", schema="hoomd", schema_version=[1, 4]) as f:
            f.write_chunk("configuration/step", np.array([5], dtype=np.uint64), write_all=False)
            f.write_chunk("particles/N", np.array([N], dtype=np.uint32), write_all=False)
            f.write_chunks([("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                            ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
                            ("particles/typeid", fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32,
                                                                            bitcast=True))], offset=np.array([N]))
            f.end_frame()
        assert os.path.getsize(mine) > 28 * N

        hpos, hvel = pos.cpu().numpy(), vel.cpu().numpy()
        lib = S.oracle_lib()
        rc = ctypes.c_int(0)
        h = lib.oracle_create_and_open(ref.encode(), 1, b"app", b"hoomd", lib.oracle_make_version(1, 4), 1, 0,
                                       ctypes.byref(rc))
        assert rc.value == 0
        S.oracle_write_chunk(lib, h, "configuration/step", 4, [np.array([[5]], dtype=np.uint64)], 1, 1, 1, [0], [1], False)
        S.oracle_write_chunk(lib, h, "particles/N", 3, [np.array([[N]], dtype=np.uint32)], 1, 1, 1, [0], [1], False)
        p3 = np.ascontiguousarray(hpos[:, :3])
        S.oracle_write_chunk(lib, h, "particles/position", 9, [p3], 3, N, 3, [0], [3 * N], True)
        del p3
        v3 = np.ascontiguousarray(hvel[:, :3])
        S.oracle_write_chunk(lib, h, "particles/velocity", 9, [v3], 3, N, 3, [0], [3 * N], True)
        del v3
        tid = np.ascontiguousarray(hpos[:, 3]).view(np.uint32).reshape(-1, 1)
        S.oracle_write_chunk(lib, h, "particles/typeid", 3, [tid], 1, N, 1, [0], [N], True)
        assert lib.oracle_end_frame(h) == 0 and lib.oracle_close(h) == 0
        assert _same_bytes(mine, ref), "device-written 80 M-particle file differs from the oracle's"
        os.unlink(ref)

        # one rank's partition into Scalar4 arrays on the device
        pos4 = torch.zeros((ROWS, 4), dtype=torch.float32, device="cuda")
        vel4 = torch.full((ROWS, 4), 1.0, dtype=torch.float32, device="cuda")
        with fl.open(mine, "r") as f:
            f.read_chunk_device(0, "particles/position", out=pos4, N=ROWS, offset=ROW0, columns=(0, 3), wait=False)
            f.read_chunk_device(0, "particles/typeid", out=pos4, N=ROWS, offset=ROW0, columns=(3, 4), bitcast=True,
                                wait=False)
            f.read_chunk_device(0, "particles/velocity", out=vel4, N=ROWS, offset=ROW0, columns=(0, 3), wait=False)
            f.wait_read()
        py = pypgsd.PGSDFile(open(mine, "rb"))
        try:
            assert py.read_chunk(0, "particles/N")[0] == N
            want_pos = py.read_chunk(0, "particles/position")[ROW0:ROW0 + ROWS]
            want_tid = py.read_chunk(0, "particles/typeid")[ROW0:ROW0 + ROWS]
            want_vel = py.read_chunk(0, "particles/velocity")[ROW0:ROW0 + ROWS]
        finally:
            py.close()
        got_pos, got_vel = pos4.cpu().numpy(), vel4.cpu().numpy()
        assert np.ascontiguousarray(got_pos[:, :3]).tobytes() == np.ascontiguousarray(want_pos).tobytes()
        assert (got_pos[:, 3].view(np.uint32) == want_tid).all()
        assert np.ascontiguousarray(got_vel[:, :3]).tobytes() == np.ascontiguousarray(want_vel).tobytes()
        assert (got_vel[:, 3] == 1.0).all()                               # the column no chunk feeds is left alone
        # and they are the rows the partition owns in the source arrays
        assert torch.equal(pos4.view(torch.int32), pos[ROW0:ROW0 + ROWS].view(torch.int32))
    finally:
        for p in (mine, ref):
            if os.path.exists(p):
                os.unlink(p)
