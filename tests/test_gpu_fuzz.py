"""Randomised field sets through the fused pack and unpack launches against the oracle's element
rules (oracle_pack_rows). Seeds are fixed: every run checks the same cases."""
import numpy as np
import pytest

import gpu_common as G

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["rows", "tiles"], autouse=True)
def unpack_kernel_family(request, monkeypatch):
    """The read-path tests run through both unpack kernels: row-per-lane (Scalar4 destinations, dense arrays)
    and the LDS-tiled one, which takes everything when PGSD_UNPACK_KERNEL=tiles."""
    monkeypatch.setenv("PGSD_UNPACK_KERNEL", request.param)
    return request.param


@pytest.fixture(params=["rows", "tiles"], autouse=True)
def pack_kernel_family(request, monkeypatch):
    """Every parity test runs through both kernel families: the row-per-lane kernel (the default for 4- and
    8-byte elements) and the LDS-tiled kernel, which takes everything when PGSD_PACK_KERNEL=tiles."""
    monkeypatch.setenv("PGSD_PACK_KERNEL", request.param)
    return request.param

torch = pytest.importorskip("torch")

INTS = ["uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "int64"]
FLOATS = ["float32", "float64"]
import os
N_SEEDS = int(os.environ.get("PGSD_FUZZ_SEEDS", "32"))     # a one-off wider campaign: PGSD_FUZZ_SEEDS=1000
SIZES = [1, 2, 15, 16, 17, 63, 64, 65, 255, 1000, 1023, 1024, 1025, 2049, 4097, 10_007]
if os.environ.get("PGSD_FUZZ_BIG"):       # one-off campaign around the launch-geometry switches (2^21 rows) and odd tails
    SIZES = [262_143, 524_289, 1_000_003, 2_097_151, 2_097_152, 2_097_153, 2_500_001]


def dev_bytes(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], -1)).cuda()


def out_types(sdt, rng):
    """destination types the path accepts for a source type (launch_pack / launch_unpack rules)"""
    s = np.dtype(sdt)
    if s.kind == 'f':
        return FLOATS
    ok = list(INTS)
    if s.itemsize <= 4:
        ok += FLOATS
    return ok


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_fused_pack(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice(SIZES))
    sources = []
    for _ in range(int(rng.integers(1, 5))):
        dt = str(rng.choice(INTS + FLOATS + ["float32"] * 4))
        stride = int(rng.integers(1, 10))
        n_src = N + int(rng.integers(0, 50)) if rng.random() < 0.4 else N
        a = G.rand_array(rng, (n_src, stride), dt)
        order = None
        if n_src != N or rng.random() < 0.3:
            order = rng.choice(n_src, size=N, replace=False).astype(np.uint32)
        sources.append((a, dev_bytes(a), order, torch.from_numpy(order.view(np.int32)).cuda() if order is not None else None))
    from pgsd import _lib
    nf = int(rng.integers(1, 11))
    jobs = (_lib.PackJob * nf)()
    outs, expect = [], []
    for i in range(nf):
        a, d, order, d_order = sources[int(rng.integers(0, len(sources)))]
        stride = a.shape[1]
        M = int(rng.integers(1, stride + 1))
        col0 = int(rng.integers(0, stride - M + 1))
        ddt = str(rng.choice(out_types(a.dtype, rng)))
        bitcast = bool(np.dtype(ddt).itemsize == a.dtype.itemsize and rng.random() < 0.5)
        if bitcast and rng.random() < 0.5:
            ddt = str(rng.choice([t for t in INTS + FLOATS if np.dtype(t).itemsize == a.dtype.itemsize]))
        elif np.dtype(ddt).kind == 'f' and a.dtype.kind != 'f' and bitcast:
            bitcast = False
        out = torch.full((N, M * np.dtype(ddt).itemsize), 0xA5, dtype=torch.uint8, device="cuda")
        jobs[i].dst = out.data_ptr()
        jobs[i].dst_type = G.type_id(ddt)
        jobs[i].M = M
        jobs[i].src.src = d.data_ptr()
        jobs[i].src.order = d_order.data_ptr() if d_order is not None else None
        jobs[i].src.src_type = G.type_id(a.dtype)
        jobs[i].src.src_stride = stride
        jobs[i].src.src_col0 = col0
        jobs[i].src.bitcast = 1 if bitcast else 0
        outs.append(out)
        expect.append(G.oracle_pack(a, M, col0=col0, out_dtype=ddt, order=order, bitcast=bitcast))
    torch.cuda.synchronize()
    rc = _lib.lib.pgsd_pack_fields(nf, jobs, N, None, None)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    for i, (out, exp) in enumerate(zip(outs, expect)):
        assert out.cpu().numpy().tobytes() == exp.tobytes(), (seed, i, N)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_fused_unpack(seed):
    """chunks -> destination arrays; several chunks may share an array (disjoint columns), some
    arrays end up completely restored (row assembly), others keep untouched columns."""
    rng = np.random.default_rng(2000 + seed)
    N = int(rng.choice(SIZES))
    from pgsd import _lib
    specs = []       # (chunk np, dst index, M, col0, bitcast)
    dsts = []        # (tensor bytes view, dtype, stride, order np or None, device order)
    fills = {}       # dst index -> fill element (pgsd_field_dst.fill_rest), for about a third of the arrays
    for _ in range(int(rng.integers(1, 5))):
        ddt = str(rng.choice(["float32", "float32", "float64", "int32", "uint32", "int64", "uint16", "uint8"]))
        stride = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8, 16]))
        order = rng.permutation(N).astype(np.uint32) if rng.random() < 0.3 else None
        t = torch.full((N, stride * np.dtype(ddt).itemsize), 0x5A, dtype=torch.uint8, device="cuda")
        dsts.append((t, ddt, stride, order, torch.from_numpy(order.view(np.int32)).cuda() if order is not None else None))
        if rng.random() < 0.35:
            fills[len(dsts) - 1] = G.rand_array(rng, (1,), ddt)[0]
        # split the columns into chunks; sometimes leave a hole
        col = 0
        while col < stride:
            M = int(rng.integers(1, stride - col + 1))
            if rng.random() < 0.8:
                if np.dtype(ddt).kind == 'f':
                    cands = FLOATS + ["int32", "uint16", "int8", "uint32"]
                else:
                    cands = INTS
                sdt = str(rng.choice(cands))
                bitcast = False
                if rng.random() < 0.3:
                    same = [x for x in INTS + FLOATS if np.dtype(x).itemsize == np.dtype(ddt).itemsize]
                    sdt, bitcast = str(rng.choice(same)), True
                specs.append((G.rand_array(rng, (N, M), sdt), len(dsts) - 1, M, col, bitcast))
            col += M
    if not specs:
        return
    order_of_jobs = rng.permutation(len(specs))
    jobs = (_lib.UnpackJob * len(specs))()
    keep = []
    for k, si in enumerate(order_of_jobs):
        chunk, di, M, col0, bitcast = specs[si]
        t, ddt, stride, order, d_order = dsts[di]
        src = torch.from_numpy(chunk.view(np.uint8).reshape(-1)).cuda()
        keep.append(src)
        jobs[k].src = src.data_ptr()
        jobs[k].src_type = G.type_id(chunk.dtype)
        jobs[k].M = M
        jobs[k].dst.dst = t.data_ptr()
        jobs[k].dst.order = d_order.data_ptr() if d_order is not None else None
        jobs[k].dst.dst_type = G.type_id(ddt)
        jobs[k].dst.dst_stride = stride
        jobs[k].dst.dst_col0 = col0
        jobs[k].dst.bitcast = 1 if bitcast else 0
        if di in fills:
            # columns of this array's rows that NO chunk of the launch writes take the fill element
            jobs[k].dst.fill_rest = 1
            jobs[k].dst.fill_bits = int(np.array([fills[di]], dtype=ddt).view("u%d" % np.dtype(ddt).itemsize)[0])
    torch.cuda.synchronize()
    rc = _lib.lib.pgsd_unpack_fields(len(specs), jobs, N, None)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    for di, (t, ddt, stride, order, _) in enumerate(dsts):
        got = t.cpu().numpy().view(ddt).reshape(N, stride)
        exp = np.frombuffer(bytes([0x5A]) * (N * stride * np.dtype(ddt).itemsize), dtype=ddt).reshape(N, stride).copy()
        if di in fills and any(dj == di for _, dj, _, _, _ in specs):
            exp[:, :] = fills[di]       # every row is touched (the scatter index is a permutation): holes = the fill
        for chunk, dj, M, col0, bitcast in specs:
            if dj != di:
                continue
            vals = G.oracle_pack(chunk, M, out_dtype=ddt, bitcast=bitcast)
            if order is not None:
                exp[order, col0:col0 + M] = vals
            else:
                exp[:, col0:col0 + M] = vals
        assert got.tobytes() == exp.tobytes(), (seed, di, ddt, stride, N)
