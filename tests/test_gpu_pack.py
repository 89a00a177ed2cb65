"""Parity of the HIP pack / select kernels (called through the C ABI) against the CPU oracle.

Bit-exact: the path is byte movement plus IEEE conversions with a single defined rounding
(f64 -> f32 round-to-nearest-even), so no tolerance is needed anywhere.
"""
import ctypes

import numpy as np
import pytest

import gpu_common as G

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["rows", "tiles"], autouse=True)
def pack_kernel_family(request, monkeypatch):
    """Every parity test runs through both kernel families: the row-per-lane kernel (the default for 4- and
    8-byte elements) and the LDS-tiled kernel, which takes everything when PGSD_PACK_KERNEL=tiles."""
    monkeypatch.setenv("PGSD_PACK_KERNEL", request.param)
    return request.param

torch = pytest.importorskip("torch")

SIZES = [1, 3, 63, 64, 65, 1000, 1023, 1024, 1025, 4099, 100003]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def empty_out(N, M, dt):
    # poison so that unwritten bytes are caught
    t = torch.empty((N * M * np.dtype(dt).itemsize + 64,), dtype=torch.uint8, device="cuda")
    t.fill_(0xAB)
    return t


def check(out_t, expect):
    got = out_t.cpu().numpy()
    nbytes = expect.nbytes
    assert got[:nbytes].tobytes() == expect.tobytes()
    assert (got[nbytes:] == 0xAB).all(), "kernel wrote past the end of the chunk"


@pytest.mark.parametrize("N", SIZES)
def test_float4_to_position_velocity_typeid(N):
    """HOOMD layout: pos = (x, y, z, type-as-int-bits), vel = (vx, vy, vz, mass)."""
    rng = np.random.default_rng(N)
    pos = G.rand_array(rng, (N, 4), np.float32)
    vel = G.rand_array(rng, (N, 4), np.float32)
    typeid = rng.integers(0, 5, size=N, dtype=np.uint32)
    pos[:, 3] = typeid.view(np.float32)
    dpos, dvel = dev(pos), dev(vel)
    o_pos, o_vel = empty_out(N, 3, np.float32), empty_out(N, 3, np.float32)
    o_tid, o_mass = empty_out(N, 1, np.uint32), empty_out(N, 1, np.float32)
    G.hip_pack([(o_pos, np.float32, 3, dpos, 0, None, False),
                (o_tid, np.uint32, 1, dpos, 3, None, True),
                (o_vel, np.float32, 3, dvel, 0, None, False),
                (o_mass, np.float32, 1, dvel, 3, None, False)], N)
    check(o_pos, G.oracle_pack(pos, 3))
    check(o_vel, G.oracle_pack(vel, 3))
    check(o_tid, G.oracle_pack(pos, 1, col0=3, out_dtype=np.uint32, bitcast=True))
    check(o_mass, G.oracle_pack(vel, 1, col0=3))
    assert (o_tid.cpu().numpy()[:4 * N].view(np.uint32) == typeid).all()


@pytest.mark.parametrize("N", [1, 64, 1000, 1025, 50001])
def test_double4_to_float3(N):
    """Scalar4 = double4 builds of HOOMD: f64 -> f32 with round-to-nearest-even,
    including values that round to inf, denormals, and signed zeros."""
    rng = np.random.default_rng(7 * N)
    pos = G.rand_array(rng, (N, 4), np.float64)
    special = np.array([0.0, -0.0, 1e-40, -1e-45, 3.5e38, -3.5e38, 1e39, np.inf, -np.inf,
                        1.0 + 2.0 ** -24, 1.0 + 2.0 ** -24 + 2.0 ** -50, 16777217.0, 1e-320])
    pos.reshape(-1)[:min(special.size, pos.size)] = special[:min(special.size, pos.size)]
    d = dev(pos)
    out = empty_out(N, 3, np.float32)
    out64 = empty_out(N, 4, np.float64)
    G.hip_pack([(out, np.float32, 3, d, 0, None, False), (out64, np.float64, 4, d, 0, None, False)], N)
    check(out, G.oracle_pack(pos, 3, out_dtype=np.float32))
    check(out64, G.oracle_pack(pos, 4))


def test_nan_payload_preserved_in_same_type_copy():
    N = 257
    a = np.full((N, 4), np.nan, dtype=np.float32)
    a.view(np.uint32)[:, 1] = 0x7FC12345
    a.view(np.uint32)[:, 2] = 0xFF800001
    d = dev(a)
    out = empty_out(N, 3, np.float32)
    G.hip_pack([(out, np.float32, 3, d, 0, None, False)], N)
    check(out, G.oracle_pack(a, 3))


@pytest.mark.parametrize("dt", ["uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "int64",
                                "float32", "float64"])
@pytest.mark.parametrize("stride,M,col0", [(1, 1, 0), (3, 3, 0), (4, 3, 1), (7, 2, 5)])
def test_all_types_same_type(dt, stride, M, col0):
    N = 2051
    rng = np.random.default_rng(abs(hash((dt, stride, M))) % 2 ** 32)
    src = G.rand_array(rng, (N, stride), dt)
    # torch lacks device copies for some unsigned types: move raw bytes
    d = dev(src.view(np.uint8).reshape(N, -1))
    out = empty_out(N, M, dt)
    from pgsd import _lib
    job = (_lib.PackJob * 1)()
    job[0].dst = out.data_ptr()
    job[0].dst_type = G.type_id(dt)
    job[0].M = M
    job[0].src.src = d.data_ptr()
    job[0].src.src_type = G.type_id(dt)
    job[0].src.src_stride = stride
    job[0].src.src_col0 = col0
    torch.cuda.synchronize()
    assert _lib.lib.pgsd_pack_fields(1, job, N, None, None) == 0
    torch.cuda.synchronize()
    check(out, G.oracle_pack(src, M, col0=col0))


CONVERSIONS = [("int8", "int32"), ("int16", "int64"), ("uint8", "uint32"), ("uint16", "uint64"),
               ("int32", "int8"), ("uint64", "uint16"), ("int32", "uint32"), ("int64", "int32"),
               ("int32", "float32"), ("uint32", "float32"), ("int16", "float64"), ("uint8", "float32"),
               ("float32", "float64"), ("float64", "float32"), ("int32", "float64"), ("uint32", "float64")]


@pytest.mark.parametrize("sdt,ddt", CONVERSIONS)
def test_conversions(sdt, ddt):
    N, stride, M, col0 = 3001, 4, 3, 1
    rng = np.random.default_rng(abs(hash((sdt, ddt))) % 2 ** 32)
    src = G.rand_array(rng, (N, stride), sdt)
    d = dev(src.view(np.uint8).reshape(N, -1))
    out = empty_out(N, M, ddt)
    from pgsd import _lib
    job = (_lib.PackJob * 1)()
    job[0].dst = out.data_ptr()
    job[0].dst_type = G.type_id(ddt)
    job[0].M = M
    job[0].src.src = d.data_ptr()
    job[0].src.src_type = G.type_id(sdt)
    job[0].src.src_stride = stride
    job[0].src.src_col0 = col0
    torch.cuda.synchronize()
    assert _lib.lib.pgsd_pack_fields(1, job, N, None, None) == 0
    torch.cuda.synchronize()
    check(out, G.oracle_pack(src, M, col0=col0, out_dtype=ddt))


def test_unsupported_conversion_is_rejected():
    from pgsd import _lib
    d = torch.zeros(16, dtype=torch.float32, device="cuda")
    out = torch.zeros(16, dtype=torch.int32, device="cuda")
    job = (_lib.PackJob * 1)()
    job[0].dst = out.data_ptr()
    job[0].dst_type = G.type_id("int32")
    job[0].M = 1
    job[0].src.src = d.data_ptr()
    job[0].src.src_type = G.type_id("float32")
    job[0].src.src_stride = 1
    assert _lib.lib.pgsd_pack_fields(1, job, 16, None, None) == _lib.ERROR_INVALID_ARGUMENT


@pytest.mark.parametrize("N", [1, 1000, 1025, 70001])
@pytest.mark.parametrize("sdt,stride,M,ddt", [("float32", 4, 3, "float32"), ("float64", 4, 3, "float32"),
                                              ("float32", 3, 3, "float32"), ("int32", 1, 1, "int32"),
                                              ("uint8", 5, 2, "uint8")])
def test_gather_in_tag_order(N, sdt, stride, M, ddt):
    """order = HOOMD's reverse tag array: row i of the chunk is source row order[i]."""
    rng = np.random.default_rng(N + stride)
    Nsrc = N + 17
    src = G.rand_array(rng, (Nsrc, stride), sdt)
    order = rng.permutation(Nsrc)[:N].astype(np.uint32)
    d = dev(src.view(np.uint8).reshape(Nsrc, -1))
    dorder = dev(order.view(np.int32))
    out = empty_out(N, M, ddt)
    from pgsd import _lib
    job = (_lib.PackJob * 1)()
    job[0].dst = out.data_ptr()
    job[0].dst_type = G.type_id(ddt)
    job[0].M = M
    job[0].src.src = d.data_ptr()
    job[0].src.order = dorder.data_ptr()
    job[0].src.src_type = G.type_id(sdt)
    job[0].src.src_stride = stride
    torch.cuda.synchronize()
    assert _lib.lib.pgsd_pack_fields(1, job, N, None, None) == 0
    torch.cuda.synchronize()
    check(out, G.oracle_pack(src, M, out_dtype=ddt, order=order))


@pytest.mark.parametrize("N", [1, 2047, 2048, 2049, 300001])
def test_gather_of_a_whole_frame_mixed_with_streamed_fields(N):
    """A tag-ordered frame in one call: position.xyz + the type id in position.w (one source array, one index list),
    velocity.xyz + mass out of a double4 array (f64 -> f32 on the way), a dense int32 array whose ROWS are gathered, a
    second index list -- and two fields that are not gathered at all (another launch of the same call).  Every chunk
    == oracle_pack_rows.  (Written for round 5's row-per-lane gather kernel, which lost its A/B against the LDS-tiled
    gather -- profiles/r05_gather_cfg_ab.jsonl -- and was not kept; the case it pins is kernel-independent.)"""
    rng = np.random.default_rng(N)
    Nsrc = N + 5
    pos = G.rand_array(rng, (Nsrc, 4), np.float32)
    vel = G.rand_array(rng, (Nsrc, 4), np.float64)
    body = G.rand_array(rng, (Nsrc, 1), np.int32)
    img = G.rand_array(rng, (Nsrc, 3), np.int32)
    order = rng.permutation(Nsrc)[:N].astype(np.uint32)
    order2 = rng.permutation(Nsrc)[:N].astype(np.uint32)
    dpos, dvel, dbody, dimg = dev(pos), dev(vel), dev(body), dev(img)
    do, do2 = dev(order.view(np.int32)), dev(order2.view(np.int32))
    spec = [(np.float32, 3, dpos, pos, 0, do, order, False), (np.uint32, 1, dpos, pos, 3, do, order, True),
            (np.float32, 3, dvel, vel, 0, do, order, False), (np.float32, 1, dvel, vel, 3, do, order, False),
            (np.int32, 1, dbody, body, 0, do, order, False), (np.int32, 3, dimg, img, 0, do2, order2, False),
            (np.float32, 3, dpos, pos, 0, None, None, False), (np.int32, 1, dbody, body, 0, None, None, False)]
    outs = [empty_out(N, M, dt) for dt, M, *_ in spec]
    G.hip_pack([(o, dt, M, d, c0, dord, bc) for o, (dt, M, d, _, c0, dord, _, bc) in zip(outs, spec)], N)
    for o, (dt, M, _, host, c0, _, ordr, bc) in zip(outs, spec):
        src = host if ordr is not None else host[:N]
        check(o, G.oracle_pack(src, M, col0=c0, out_dtype=dt, order=ordr, bitcast=bc))


def test_unaligned_pointers_take_the_generic_kernel():
    N = 5003
    rng = np.random.default_rng(5)
    src = G.rand_array(rng, (N, 4), np.float32)
    raw = torch.empty(src.nbytes + 16, dtype=torch.uint8, device="cuda")
    raw[4:4 + src.nbytes] = dev(src.view(np.uint8).reshape(-1))
    out = empty_out(N + 1, 3, np.float32)
    from pgsd import _lib
    job = (_lib.PackJob * 1)()
    job[0].dst = out.data_ptr() + 4
    job[0].dst_type = G.type_id("float32")
    job[0].M = 3
    job[0].src.src = raw.data_ptr() + 4
    job[0].src.src_type = G.type_id("float32")
    job[0].src.src_stride = 4
    torch.cuda.synchronize()
    assert _lib.lib.pgsd_pack_fields(1, job, N, None, None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got[4:4 + N * 12].tobytes() == G.oracle_pack(src, 3).tobytes()
    assert (got[:4] == 0xAB).all() and (got[4 + N * 12:] == 0xAB).all()


def test_wide_rows():
    N = 300
    rng = np.random.default_rng(11)
    src = G.rand_array(rng, (N, 96), np.float64)  # 768-byte rows
    d = dev(src)
    out = empty_out(N, 40, np.float32)
    G.hip_pack([(out, np.float32, 40, d, 13, None, False)], N)
    check(out, G.oracle_pack(src, 40, col0=13, out_dtype=np.float32))


def test_more_fields_than_one_launch_holds():
    N = 1500
    rng = np.random.default_rng(3)
    srcs = [G.rand_array(rng, (N, 4), np.float32) for _ in range(11)]
    ds = [dev(s) for s in srcs]
    outs = [empty_out(N, 3, np.float32) for _ in range(11)]
    outs2 = [empty_out(N, 1, np.float32) for _ in range(11)]
    jobs = []
    for d, o, o2 in zip(ds, outs, outs2):
        jobs.append((o, np.float32, 3, d, 0, None, False))
        jobs.append((o2, np.float32, 1, d, 3, None, False))
    G.hip_pack(jobs, N)
    for s, o, o2 in zip(srcs, outs, outs2):
        check(o, G.oracle_pack(s, 3))
        check(o2, G.oracle_pack(s, 1, col0=3))


@pytest.mark.parametrize("N,p", [(1, 1.0), (15, 0.5), (16, 0.5), (4096, 0.3), (4097, 0.9), (100003, 0.01),
                                 (1 << 20, 0.5), (5000, 0.0), (5000, 1.0)])
def test_select_rows(N, p):
    from pgsd import _lib
    rng = np.random.default_rng(N)
    flags = (rng.random(N) < p).astype(np.uint8) * rng.integers(1, 255, size=N, dtype=np.uint8)
    dflags = dev(flags)
    out = torch.full((N + 8,), -1, dtype=torch.int32, device="cuda")
    import ctypes
    count = ctypes.c_uint64(1 << 40)
    torch.cuda.synchronize()
    rc = _lib.lib.pgsd_select_rows(dflags.data_ptr(), N, out.data_ptr(), ctypes.byref(count), None)   # count: host memory
    assert rc == 0, _lib.last_error()
    expect = np.nonzero(flags)[0].astype(np.int32)
    assert int(count.value) == expect.size
    got = out.cpu().numpy()
    assert (got[:expect.size] == expect).all()
    assert (got[expect.size:] == -1).all()


@pytest.mark.parametrize("N,shift", [(10_000_001, 0), (300_007, 3), (4099, 1), (17, 5)])
def test_select_rows_with_unaligned_flags_on_a_callers_stream(N, shift):
    """Flags that do not start on a 16-byte boundary (the vector loads fall back to bytes), on a stream of the caller's
    behind a kernel that produces them; several calls in a row (the scratch space and the pinned count are reused)."""
    import ctypes
    from pgsd import _lib
    rng = np.random.default_rng(N + shift)
    stream = torch.cuda.Stream()
    buf = torch.zeros((N + 16,), dtype=torch.uint8, device="cuda")
    out = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    for keep in (0.5, 0.02, 0.97):
        flags = (rng.random(N) < keep).astype(np.uint8)
        host = torch.from_numpy(flags).pin_memory()
        with torch.cuda.stream(stream):
            buf[shift:shift + N].copy_(host, non_blocking=True)          # produced on the caller's stream
            count = ctypes.c_uint64(0)
            rc = _lib.lib.pgsd_select_rows(buf.data_ptr() + shift, N, out.data_ptr(), ctypes.byref(count), stream.cuda_stream)
        assert rc == 0, _lib.last_error()
        expect = np.nonzero(flags)[0].astype(np.int32)
        assert int(count.value) == expect.size
        stream.synchronize()
        assert (out[:expect.size].cpu().numpy() == expect).all()


@pytest.mark.parametrize("prefetch", ["default", "1", "0"])
def test_full_size_pack_matches_independent_gpu_slicing(prefetch, monkeypatch):
    """BASELINE size (10 M particles): compare against torch's own strided copy on the GPU
    (an independent implementation), bit for bit, plus a checksum of checksums. Runs through the
    plain tiled kernel and through the software-pipelined one (many tiles per workgroup)."""
    if prefetch != "default":
        monkeypatch.setenv("PGSD_PACK_PREFETCH", prefetch)
    N = 10_000_000 if prefetch != "1" else 10_000_000 - 777     # ragged last tile for the pipelined loop
    g = torch.Generator(device="cuda").manual_seed(1234)
    pos = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 100.0
    vel = torch.randn((N, 4), generator=g, device="cuda")
    tid = torch.randperm(N, generator=g, device="cuda").to(torch.int32)
    o_pos = torch.empty((N, 3), dtype=torch.float32, device="cuda")
    o_vel = torch.empty((N, 3), dtype=torch.float32, device="cuda")
    o_tid = torch.empty((N,), dtype=torch.int32, device="cuda")
    G.hip_pack([(o_pos, np.float32, 3, pos, 0, None, False), (o_vel, np.float32, 3, vel, 0, None, False),
                (o_tid, np.uint32, 1, tid.view(N, 1), 0, None, False)], N)
    assert torch.equal(o_pos.view(torch.int32), pos[:, :3].contiguous().view(torch.int32))
    assert torch.equal(o_vel.view(torch.int32), vel[:, :3].contiguous().view(torch.int32))
    assert torch.equal(o_tid, tid)
    # idempotence: packing the packed chunk with stride == M is the identity
    again = torch.empty_like(o_pos)
    G.hip_pack([(again, np.float32, 3, o_pos, 0, None, False)], N)
    assert torch.equal(again.view(torch.int32), o_pos.view(torch.int32))
    assert int(o_tid.to(torch.int64).sum().item()) == N * (N - 1) // 2


def test_arrays_larger_than_4_GiB():
    """300 M float4 rows (4.8 GB source, 3.6 GB chunk): byte offsets past 2^32 in the pack and in
    the unpack; checked against torch's own slicing on the GPU."""
    N = 300_000_000
    src = torch.empty((N, 4), dtype=torch.float32, device="cuda")
    src.view(torch.int32).copy_(torch.arange(4 * N, dtype=torch.int32, device="cuda").view(N, 4))  # every word distinct
    out = torch.empty((N, 3), dtype=torch.float32, device="cuda")
    w = torch.empty((N,), dtype=torch.int32, device="cuda")
    G.hip_pack([(out, np.float32, 3, src, 0, None, False), (w, np.uint32, 1, src, 3, None, True)], N)
    for lo in (0, 89_478_480, 178_956_960, 268_435_440, N - 5000):           # around the 2^30-row / 2^32-byte lines
        hi = lo + 5000
        assert torch.equal(out[lo:hi].view(torch.int32), src[lo:hi, :3].contiguous().view(torch.int32)), lo
        assert torch.equal(w[lo:hi], src[lo:hi, 3].view(torch.int32)), lo
    assert torch.equal(out.view(torch.int32)[:, 1], src.view(torch.int32)[:, 1])             # one whole column
    # and back: chunk -> a fresh Scalar4 array in one launch (whole rows)
    back = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    from pgsd import _lib
    jobs = (_lib.UnpackJob * 2)()
    for i, (chunk, M, c0, bc) in enumerate(((out, 3, 0, 0), (w, 1, 3, 1))):
        jobs[i].src = chunk.data_ptr()
        jobs[i].src_type = 9 if i == 0 else 3
        jobs[i].M = M
        jobs[i].dst.dst = back.data_ptr()
        jobs[i].dst.dst_type = 9
        jobs[i].dst.dst_stride = 4
        jobs[i].dst.dst_col0 = c0
        jobs[i].dst.bitcast = bc
    torch.cuda.synchronize()
    assert _lib.lib.pgsd_unpack_fields(2, jobs, N, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(back.view(torch.int32), src.view(torch.int32))


@pytest.mark.parametrize("layout", ["float4_idw", "float4_separate_id", "double4_separate_id"])
def test_config2_one_mebi_rows_position_velocity_id(layout):
    """BASELINE config 2 at its own size: 2^20 particles, position + velocity + id packed by one launch,
    bit-exact against `oracle_pack_rows` -- HOOMD float4 arrays with the id in position.w, float4 arrays with
    the id in its own uint32 array (SURVEY 8(d)'s sketch), and the double4 -> float32 conversion variant.
    Values as SURVEY 8(d) prescribes: pos = U(-L/2, L/2), L = 100; vel = N(0, 1); id = a permutation; seed 1234.
    (Runs through both kernel families: the autouse fixture of this file.)"""
    N = 1 << 20
    rng = np.random.default_rng(1234)
    ft = np.float64 if layout.startswith("double4") else np.float32
    pos = ((rng.random((N, 4)) - 0.5) * 100.0).astype(ft)
    vel = rng.standard_normal((N, 4)).astype(ft)
    ids = rng.permutation(N).astype(np.uint32)
    if layout == "float4_idw":
        pos[:, 3] = ids.view(np.float32)
    dpos, dvel = dev(pos), dev(vel)
    dids = dev(ids.view(np.int32).reshape(N, 1))
    o_pos, o_vel, o_id = empty_out(N, 3, np.float32), empty_out(N, 3, np.float32), empty_out(N, 1, np.uint32)
    id_job = (o_id, np.uint32, 1, dpos, 3, None, True) if layout == "float4_idw" else \
        (o_id, np.uint32, 1, dids.view(torch.int32), 0, None, False)
    G.hip_pack([(o_pos, np.float32, 3, dpos, 0, None, False), (o_vel, np.float32, 3, dvel, 0, None, False), id_job], N)
    check(o_pos, G.oracle_pack(pos, 3, out_dtype=np.float32))
    check(o_vel, G.oracle_pack(vel, 3, out_dtype=np.float32))
    if layout == "float4_idw":
        check(o_id, G.oracle_pack(pos, 1, col0=3, out_dtype=np.uint32, bitcast=True))
    else:
        check(o_id, G.oracle_pack(ids.view(np.int32).reshape(N, 1), 1, out_dtype=np.uint32))
    assert (o_id.cpu().numpy()[:4 * N].view(np.uint32) == ids).all()
    assert int(o_id.cpu().numpy()[:4 * N].view(np.uint32).astype(np.int64).sum()) == N * (N - 1) // 2


@pytest.mark.parametrize("cfg", ["64x4", "1024x1", "0x0", "junk", "256x8"])
def test_unknown_rows_tuning_pair_is_ignored_not_obeyed(cfg, monkeypatch):
    """PGSD_PACK_ROWS_CFG / PGSD_UNPACK_ROWS_CFG name launch shapes for tuning sweeps.  A pair that is not
    instantiated used to size the grid from the requested T x U while the kernel ran the 128 x 1 fallback: part of
    every chunk stayed unwritten ('0x0' divided by zero).  Unknown pairs are now ignored."""
    monkeypatch.setenv("PGSD_PACK_ROWS_CFG", cfg)
    monkeypatch.setenv("PGSD_UNPACK_ROWS_CFG", cfg)
    N = 70_001
    rng = np.random.default_rng(3)
    pos = G.rand_array(rng, (N, 4), np.float32)
    d = dev(pos)
    out, w = empty_out(N, 3, np.float32), empty_out(N, 1, np.float32)
    G.hip_pack([(out, np.float32, 3, d, 0, None, False), (w, np.float32, 1, d, 3, None, False)], N)
    check(out, G.oracle_pack(pos, 3))
    check(w, G.oracle_pack(pos, 1, col0=3))
    from pgsd import _lib
    back = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    jobs = (_lib.UnpackJob * 2)()
    for i, (chunk, M, c0) in enumerate(((out, 3, 0), (w, 1, 3))):
        jobs[i].src, jobs[i].src_type, jobs[i].M = chunk.data_ptr(), 9, M
        jobs[i].dst.dst, jobs[i].dst.dst_type, jobs[i].dst.dst_stride, jobs[i].dst.dst_col0 = back.data_ptr(), 9, 4, c0
    torch.cuda.synchronize()
    assert _lib.lib.pgsd_unpack_fields(2, jobs, N, None) == 0
    torch.cuda.synchronize()
    assert back.cpu().numpy().tobytes() == pos.tobytes()
