"""The pack oracle (oracle_pack_rows) against plain numpy semantics, and the integer identity the
HIP kernel's row/column split relies on."""
import numpy as np
import pytest

import gpu_common as G


@pytest.mark.parametrize("sdt,ddt", [("float32", "float32"), ("float64", "float32"), ("float32", "float64"),
                                     ("int8", "int32"), ("uint8", "uint32"), ("int32", "int8"),
                                     ("int64", "int32"), ("uint32", "float32"), ("int16", "float64"),
                                     ("int32", "float32")])
def test_oracle_pack_equals_numpy_cast(sdt, ddt):
    rng = np.random.default_rng(1)
    src = G.rand_array(rng, (501, 5), sdt)
    got = G.oracle_pack(src, 3, col0=1, out_dtype=ddt)
    with np.errstate(over="ignore", invalid="ignore"):
        expect = src[:, 1:4].astype(ddt)
    assert got.tobytes() == np.ascontiguousarray(expect).tobytes()


def test_oracle_pack_gather_and_bitcast():
    rng = np.random.default_rng(2)
    src = G.rand_array(rng, (300, 4), np.float32)
    order = rng.permutation(300)[:120].astype(np.uint32)
    got = G.oracle_pack(src, 3, order=order)
    assert got.tobytes() == np.ascontiguousarray(src[order, :3]).tobytes()
    ids = rng.integers(0, 1 << 31, size=300, dtype=np.uint32)
    src[:, 3] = ids.view(np.float32)
    got = G.oracle_pack(src, 1, col0=3, out_dtype=np.uint32, bitcast=True)
    assert (got[:, 0] == ids).all()
    wide = G.rand_array(rng, (10, 2), np.float64)
    low = G.oracle_pack(wide, 2, out_dtype=np.uint32, bitcast=True)
    assert (low == (wide.view(np.uint64) & 0xFFFFFFFF).astype(np.uint32)).all()


def test_rejected_conversions():
    lib = G.S.oracle_lib()
    a = np.zeros((4, 1), dtype=np.float32)
    out = np.zeros((4, 1), dtype=np.int32)
    assert lib.oracle_pack_rows(out.ctypes.data, 7, a.ctypes.data, 9, 4, 1, 1, 0, None, 0) == -2   # float -> int
    b = np.zeros((4, 1), dtype=np.int64)
    assert lib.oracle_pack_rows(a.ctypes.data, 9, b.ctypes.data, 8, 4, 1, 1, 0, None, 0) == -2     # i64 -> float
    assert lib.oracle_pack_rows(a.ctypes.data, 9, a.ctypes.data, 9, 4, 2, 1, 0, None, 0) == -2     # M > stride


@pytest.mark.parametrize("M", [2, 3, 5, 6, 7, 12, 96, 1000, 1024])
def test_multiply_high_division_is_exact(M):
    """pack kernels compute row = e / M as umulhi(e, ceil(2^32 / M)) for e < tile_rows * M."""
    magic = ((1 << 32) + M - 1) // M
    e = np.arange(0, 1024 * M, dtype=np.uint64)
    assert ((e * np.uint64(magic)) >> np.uint64(32) == e // np.uint64(M)).all()
