"""Read path on the GPU (BASELINE config 5): file -> pinned slabs -> HBM -> HIP unpack, against
the pure-Python reader (pgsd.pypgsd) on the same file. Bit-exact."""
import ctypes
import os

import numpy as np
import pytest

import gpu_common as G
import scenario as S

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["rows", "tiles"], autouse=True)
def unpack_kernel_family(request, monkeypatch):
    """The read-path tests run through both unpack kernels: row-per-lane (Scalar4 destinations, dense arrays)
    and the LDS-tiled one, which takes everything when PGSD_UNPACK_KERNEL=tiles."""
    monkeypatch.setenv("PGSD_UNPACK_KERNEL", request.param)
    return request.param

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def unpack(src_np, dst_t, M, col0=0, order=None, bitcast=False, N=None):
    from pgsd import _lib
    src = dev(src_np.view(np.uint8).reshape(-1))
    job = (_lib.UnpackJob * 1)()
    job[0].src = src.data_ptr()
    job[0].src_type = G.type_id(src_np.dtype)
    job[0].M = M
    job[0].dst.dst = dst_t.data_ptr()
    job[0].dst.order = order.data_ptr() if order is not None else None
    job[0].dst.dst_type = G.type_id(str(dst_t.dtype)[6:])
    job[0].dst.dst_stride = dst_t.shape[1] if dst_t.dim() == 2 else 1
    job[0].dst.dst_col0 = col0
    job[0].dst.bitcast = 1 if bitcast else 0
    torch.cuda.synchronize()
    rc = _lib.lib.pgsd_unpack_fields(1, job, N if N is not None else src_np.shape[0], None)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("N", [1, 63, 1000, 1025, 70001])
def test_unpack_float3_and_typeid_into_scalar4(N):
    rng = np.random.default_rng(N)
    pos = G.rand_array(rng, (N, 3), np.float32)
    tid = rng.integers(0, 9, size=(N, 1)).astype(np.uint32)
    pos4 = torch.full((N, 4), 7.5, dtype=torch.float32, device="cuda")
    unpack(pos, pos4, 3)
    got = pos4.cpu().numpy()
    assert got[:, :3].tobytes() == pos.tobytes() and (got[:, 3] == 7.5).all()   # w untouched
    unpack(tid, pos4, 1, col0=3, bitcast=True)
    got = pos4.cpu().numpy()
    assert got[:, :3].tobytes() == pos.tobytes()
    assert (got[:, 3].view(np.uint32) == tid[:, 0]).all()


@pytest.mark.parametrize("sdt,ddt", [("float32", "float64"), ("float64", "float32"), ("int32", "int64"),
                                     ("uint8", "uint32"), ("int16", "float32"), ("uint32", "float64"),
                                     ("int64", "int32"), ("float32", "float32"), ("uint16", "uint16")])
def test_unpack_conversions(sdt, ddt):
    N, M, S, c0 = 4099, 3, 5, 1
    rng = np.random.default_rng(abs(hash((sdt, ddt))) % 2 ** 32)
    src = G.rand_array(rng, (N, M), sdt)
    dst = torch.zeros((N, S * np.dtype(ddt).itemsize), dtype=torch.uint8, device="cuda")
    from pgsd import _lib
    dsrc = dev(src.view(np.uint8).reshape(-1))
    job = (_lib.UnpackJob * 1)()
    job[0].src = dsrc.data_ptr()
    job[0].src_type = G.type_id(sdt)
    job[0].M = M
    job[0].dst.dst = dst.data_ptr()
    job[0].dst.dst_type = G.type_id(ddt)
    job[0].dst.dst_stride = S
    job[0].dst.dst_col0 = c0
    torch.cuda.synchronize()
    assert _lib.lib.pgsd_unpack_fields(1, job, N, None) == 0
    torch.cuda.synchronize()
    got = dst.cpu().numpy().view(ddt).reshape(N, S)
    expect = G.oracle_pack(src, M, out_dtype=ddt)          # same element conversion rules
    assert got[:, c0:c0 + M].tobytes() == expect.tobytes()
    assert (got[:, :c0] == 0).all() and (got[:, c0 + M:] == 0).all()


def test_unpack_scatter_in_tag_order():
    N = 5000
    rng = np.random.default_rng(3)
    src = G.rand_array(rng, (N, 3), np.float32)
    order = rng.permutation(N).astype(np.int32)
    dst = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    unpack(src, dst, 3, order=dev(order))
    got = dst.cpu().numpy()
    assert got[order, :3].tobytes() == src.tobytes()


def unpack_many(specs, N, keep):
    """specs: (src_np, dst_tensor, M, col0, order, bitcast); ONE pgsd_unpack_fields call."""
    from pgsd import _lib
    jobs = (_lib.UnpackJob * len(specs))()
    for i, (src_np, dst_t, M, col0, order, bitcast) in enumerate(specs):
        src = dev(src_np.view(np.uint8).reshape(-1))
        keep.append(src)
        jobs[i].src = src.data_ptr()
        jobs[i].src_type = G.type_id(src_np.dtype)
        jobs[i].M = M
        jobs[i].dst.dst = dst_t.data_ptr()
        jobs[i].dst.order = order.data_ptr() if order is not None else None
        jobs[i].dst.dst_type = G.type_id(str(dst_t.dtype)[6:])
        jobs[i].dst.dst_stride = dst_t.shape[1] if dst_t.dim() == 2 else 1
        jobs[i].dst.dst_col0 = col0
        jobs[i].dst.bitcast = 1 if bitcast else 0
    torch.cuda.synchronize()
    rc = _lib.lib.pgsd_unpack_fields(len(specs), jobs, N, None)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("N", [1, 15, 63, 1000, 1025, 70001])
@pytest.mark.parametrize("use_order", [False, True])
def test_unpack_one_launch_assembles_whole_rows(N, use_order):
    """position+typeid -> pos4 and velocity+mass -> vel4 (16-byte rows assembled in registers),
    image -> dense int32 rows (element path), all in one launch."""
    rng = np.random.default_rng(N + 5)
    pos = G.rand_array(rng, (N, 3), np.float32)
    vel = G.rand_array(rng, (N, 3), np.float32)
    mass = G.rand_array(rng, (N, 1), np.float32)
    tid = rng.integers(0, 2 ** 32, size=(N, 1), dtype=np.uint64).astype(np.uint32)
    img = rng.integers(-5, 6, size=(N, 3)).astype(np.int32)
    order_np = rng.permutation(N).astype(np.int32) if use_order else np.arange(N, dtype=np.int32)
    order = dev(order_np) if use_order else None
    pos4 = torch.full((N, 4), 7.5, dtype=torch.float32, device="cuda")
    vel4 = torch.full((N, 4), 7.5, dtype=torch.float32, device="cuda")
    img3 = torch.zeros((N, 3), dtype=torch.int32, device="cuda")
    keep = []
    unpack_many([(pos, pos4, 3, 0, order, False), (vel, vel4, 3, 0, order, False), (img, img3, 3, 0, None, False),
                 (tid, pos4, 1, 3, order, True), (mass, vel4, 1, 3, order, False)], N, keep)
    p, v = pos4.cpu().numpy(), vel4.cpu().numpy()
    assert p[order_np, :3].tobytes() == pos.tobytes()
    assert p[order_np, 3].view(np.uint32).tobytes() == tid.tobytes()
    assert v[order_np, :3].tobytes() == vel.tobytes() and v[order_np, 3].tobytes() == mass.tobytes()
    assert img3.cpu().numpy().tobytes() == img.tobytes()


def test_unpack_wide_rows_with_conversion():
    """f32 chunks into a double4 array (32-byte rows), f64 + i32->f32 into float4, ints into 64-byte rows."""
    N = 33_333
    rng = np.random.default_rng(8)
    pos = G.rand_array(rng, (N, 3), np.float32)
    w = G.rand_array(rng, (N, 1), np.float32)
    d4 = torch.zeros((N, 4), dtype=torch.float64, device="cuda")
    vd = G.rand_array(rng, (N, 2), np.float64)
    vi = rng.integers(-1000, 1000, size=(N, 2)).astype(np.int32)
    f4 = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    a = rng.integers(-2 ** 31, 2 ** 31, size=(N, 5)).astype(np.int32)
    b = rng.integers(-2 ** 31, 2 ** 31, size=(N, 3)).astype(np.int32)
    i8 = torch.zeros((N, 8), dtype=torch.int64, device="cuda")
    u16 = torch.zeros((N, 16), dtype=torch.int32, device="cuda")
    c = rng.integers(0, 2 ** 16, size=(N, 16)).astype(np.uint16)
    keep = []
    unpack_many([(pos, d4, 3, 0, None, False), (w, d4, 1, 3, None, False),
                 (vd, f4, 2, 0, None, False), (vi, f4, 2, 2, None, False),
                 (b, i8, 3, 5, None, False), (a, i8, 5, 0, None, False),
                 (c, u16, 16, 0, None, False)], N, keep)
    got = d4.cpu().numpy()
    assert (got[:, :3] == pos.astype(np.float64)).all() and (got[:, 3] == w[:, 0].astype(np.float64)).all()
    got = f4.cpu().numpy()
    assert got[:, :2].tobytes() == vd.astype(np.float32).tobytes() and (got[:, 2:] == vi.astype(np.float32)).all()
    got = i8.cpu().numpy()
    assert (got[:, :5] == a.astype(np.int64)).all() and (got[:, 5:] == b.astype(np.int64)).all()
    assert (u16.cpu().numpy() == c.astype(np.int32)).all()


def test_unpack_later_chunk_wins_and_many_jobs():
    N = 5003
    rng = np.random.default_rng(9)
    first = G.rand_array(rng, (N, 4), np.float32)
    second = G.rand_array(rng, (N, 2), np.float32)
    dst = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    keep = []
    unpack_many([(first, dst, 4, 0, None, False), (second, dst, 2, 1, None, False)], N, keep)
    got = dst.cpu().numpy()
    assert got[:, 1:3].tobytes() == second.tobytes() and got[:, 0].tobytes() == first[:, 0].tobytes()
    assert got[:, 3].tobytes() == first[:, 3].tobytes()
    # more chunks than one launch holds, rows too wide to stage together
    srcs = [G.rand_array(rng, (N, 100 + i), np.float32) for i in range(15)]
    dsts = [torch.zeros((N, 101 + i), dtype=torch.float32, device="cuda") for i in range(15)]
    unpack_many([(s_, d_, s_.shape[1], 1, None, False) for s_, d_ in zip(srcs, dsts)], N, keep)
    for s_, d_ in zip(srcs, dsts):
        g_ = d_.cpu().numpy()
        assert g_[:, 1:].tobytes() == s_.tobytes() and (g_[:, 0] == 0).all()


def _write_file(path, N, frames=2):
    import pgsd.fl as fl
    rng = np.random.default_rng(17)
    data = []
    with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        for i in range(frames):
            d = {'particles/position': G.rand_array(rng, (N, 3), np.float32),
                 'particles/velocity': G.rand_array(rng, (N, 3), np.float32),
                 'particles/typeid': rng.integers(0, 4, size=N).astype(np.uint32),
                 'particles/mass': (rng.random(N) + 0.5).astype(np.float32),
                 'particles/image': rng.integers(-2, 3, size=(N, 3)).astype(np.int32)}
            f.write_chunk('configuration/step', np.array([10 * i], dtype=np.uint64))
            f.write_chunk('particles/N', np.array([N], dtype=np.uint32))
            for k, v in d.items():
                f.write_chunk(k, v)
            f.end_frame()
            data.append(d)
    return data


@pytest.mark.parametrize("N", [1, 1000, 300_007])
def test_read_chunk_device_matches_python_reader(N, tmp_gsd):
    import pgsd.fl as fl
    import pgsd.pypgsd as pypgsd
    _write_file(tmp_gsd, N)
    ref = pypgsd.PGSDFile(open(tmp_gsd, 'rb'))
    with fl.open(tmp_gsd, 'r') as f:
        f.configure_device(slab_bytes=64 * 1024, n_slabs=4)      # many pieces per chunk
        for frame in (0, 1):
            for name in ('particles/position', 'particles/typeid', 'particles/image', 'particles/mass'):
                got = f.read_chunk_device(frame, name)
                exp = ref.read_chunk(frame, name)
                assert tuple(got.shape) == exp.shape
                assert got.cpu().numpy().tobytes() == exp.tobytes(), (frame, name)
        # a rank's partition, into a Scalar4 array, typeid into w
        row0, n = N // 3, N - N // 3
        pos4 = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        f.read_chunk_device(1, 'particles/position', out=pos4, N=n, offset=row0, columns=(0, 3), wait=False)
        f.read_chunk_device(1, 'particles/typeid', out=pos4, N=n, offset=row0, columns=(3, 4), bitcast=True, wait=False)
        f.wait_read()
        got = pos4.cpu().numpy()
        assert got[:, :3].tobytes() == ref.read_chunk(1, 'particles/position').reshape(N, 3)[row0:].tobytes()
        assert (got[:, 3].view(np.uint32) == ref.read_chunk(1, 'particles/typeid')[row0:]).all()
        # f32 chunk into a double-precision array (Scalar = double builds)
        pos_d = torch.zeros((N, 4), dtype=torch.float64, device="cuda")
        f.read_chunk_device(0, 'particles/position', out=pos_d, columns=(0, 3))
        assert (pos_d.cpu().numpy()[:, :3] == ref.read_chunk(0, 'particles/position').reshape(N, 3).astype(np.float64)).all()
        with pytest.raises(KeyError):
            f.read_chunk_device(0, 'particles/nope')
        with pytest.raises(ValueError):
            f.read_chunk_device(0, 'particles/position', N=N + 1)
    ref.close()


def test_hoomd_read_frame_device_round_trip(tmp_gsd):
    """write from the GPU, read back to the GPU (partition + Scalar4 assembly)."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    N = 50_001
    g = torch.Generator(device="cuda").manual_seed(5)
    pos4 = torch.randn((N, 4), generator=g, device="cuda")
    vel4 = torch.randn((N, 4), generator=g, device="cuda")
    tid = torch.randint(0, 5, (N,), generator=g, device="cuda", dtype=torch.int32)
    pos4[:, 3] = tid.view(torch.float32)
    with hoomd.open(tmp_gsd, 'w') as t:
        fr = hoomd.Frame()
        fr.configuration.step = 123
        fr.configuration.box = [9, 9, 9, 0, 0, 0]
        fr.particles.N = N
        fr.particles.types = ['a', 'b', 'c', 'd', 'e']
        fr.particles.position = fl.DeviceField.from_tensor(pos4, columns=(0, 3))
        fr.particles.typeid = fl.DeviceField.from_tensor(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True)
        fr.particles.velocity = fl.DeviceField.from_tensor(vel4, columns=(0, 3))
        fr.particles.mass = fl.DeviceField.from_tensor(vel4, columns=(3, 4))
        t.append(fr)
    with hoomd.open(tmp_gsd, 'r') as t:
        s = t.read_frame_device(0, scalar4=True)
        assert s.configuration.step == 123 and s.particles.N == N and s.particles.N_global == N
        assert s.particles.types == ['a', 'b', 'c', 'd', 'e']
        assert torch.equal(s.particles.pos4.view(torch.int32), pos4.view(torch.int32))
        assert torch.equal(s.particles.vel4.view(torch.int32), vel4.view(torch.int32))
        assert torch.equal(s.particles.position, pos4[:, :3].contiguous())
        assert torch.equal(s.particles.typeid.view(torch.int32), tid)
        assert float(s.particles.density.abs().sum()) == 0.0          # default
        # fields no frame holds are ONE default row broadcast over the particles (the host reader's read-only
        # broadcast arrays, hoomd.py:872-881): right shape and dtype, no storage of their own until asked for
        body, image = s.particles.body, s.particles.image
        assert body.shape == (N,) and body.dtype == torch.int32 and body.stride(0) == 0
        assert image.shape == (N, 3) and image.dtype == torch.int32 and image.stride(0) == 0
        assert bool((body == -1).all()) and bool((image == 0).all()) and bool((s.particles.slength == 1).all())
        own = body.contiguous()
        own[0] = 7
        assert int(own[0]) == 7 and int(body[0]) == -1
        part = t.read_frame_device(0, part=(100, 777))
        assert part.particles.N == 777
        assert torch.equal(part.particles.velocity, vel4[100:877, :3].contiguous())


def test_many_reading_handles_share_one_reader_engine(tmp_path):
    """Six trajectories open for reading at once: their pieces go through ONE set of reader threads
    and one pinned ring; handles are closed while others still have reads in flight."""
    import pgsd.fl as fl
    import pgsd.pypgsd as pypgsd
    N = 150_001
    paths, handles = [], []
    for k in range(6):
        p = str(tmp_path / ("t%d.gsd" % k))
        _write_file(p, N + k, frames=1)
        paths.append(p)
    import os
    before = len(os.listdir('/proc/self/task'))
    handles = [fl.open(p, 'r') for p in paths]
    outs = []
    for k, f in enumerate(handles):
        outs.append((f.read_chunk_device(0, 'particles/position', wait=False),
                     f.read_chunk_device(0, 'particles/image', wait=False)))
    n_threads = len(os.listdir('/proc/self/task')) - before
    for k in (5, 0, 3):
        handles[k].wait_read()
        handles[k].close()
    for k in (1, 2, 4):
        handles[k].wait_read()
    for k in range(6):
        ref = pypgsd.PGSDFile(open(paths[k], 'rb'))
        assert outs[k][0].cpu().numpy().tobytes() == ref.read_chunk(0, 'particles/position').tobytes(), k
        assert outs[k][1].cpu().numpy().tobytes() == ref.read_chunk(0, 'particles/image').tobytes(), k
        ref.close()
    for k in (1, 2, 4):
        handles[k].close()
    # 6 handles x (dispatcher + writer) + 16 shared readers, not 6 x 16 readers
    assert n_threads < 60, n_threads
    # a fresh handle after every reader was released builds a new engine
    with fl.open(paths[0], 'r') as f:
        got = f.read_chunk_device(0, 'particles/typeid')
        assert got.shape[0] == N


def _fnv1a(data):
    h = 0xcbf29ce484222325
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("P", [1, 2, 4])
def test_device_reads_reproduce_the_references_own_read_hashes(P):
    """tests/golden/files/readback.p<P>.{gsd,log} were written AND read by the compiled reference
    (pgsd_find_chunk + pgsd_read_chunk under mpiexec; the log holds an FNV-1a hash of the bytes each read
    returned).  The device read path (pread -> pinned slabs -> HBM -> unpack kernel) must return the same
    bytes for the same frame / chunk / row slab: whole chunks, slabs, buffered small chunks."""
    import re
    import pgsd.fl as fl
    gsd = os.path.join(S.GOLDEN, "readback.p%d.gsd" % P)
    lines = [ln for ln in S.read_log(gsd[:-4] + ".log") if ln.startswith("read ")]
    f = fl.open(gsd, "r")
    checked = 0
    for ln in lines:
        m = re.match(r"read line=\d+ frame=(\d+) name=(\S+) N=(\d+) M=(\d+) offset=(\d+) all=(\d) rc=(-?\d+) bytes=(\d+) "
                     r"fnv=([0-9a-f]{16})", ln)
        frame, name, N, M, off, all_, rc, nbytes, fnv = m.groups()
        if int(rc) != 0 or int(frame) >= f.nframes:
            continue
        if int(all_):
            t = f.read_chunk_device(int(frame), name, N=int(N), offset=int(off))
        else:
            t = f.read_chunk_device(int(frame), name)
        got = t.cpu().numpy().tobytes()
        assert len(got) == int(nbytes), ln
        assert "%016x" % _fnv1a(got) == fnv, ln
        checked += 1
    f.close()
    assert checked >= 10


def unpack_filled(specs, N, keep):
    """like unpack_many, specs carry a fill value: (src_np, dst_tensor, M, col0, order, bitcast, fill or None)"""
    from pgsd import _lib
    jobs = (_lib.UnpackJob * len(specs))()
    for i, (src_np, dst_t, M, col0, order, bitcast, fill) in enumerate(specs):
        src = dev(src_np.view(np.uint8).reshape(-1))
        keep.append(src)
        jobs[i].src = src.data_ptr()
        jobs[i].src_type = G.type_id(src_np.dtype)
        jobs[i].M = M
        jobs[i].dst.dst = dst_t.data_ptr()
        jobs[i].dst.order = order.data_ptr() if order is not None else None
        jobs[i].dst.dst_type = G.type_id(str(dst_t.dtype)[6:])
        jobs[i].dst.dst_stride = dst_t.shape[1] if dst_t.dim() == 2 else 1
        jobs[i].dst.dst_col0 = col0
        jobs[i].dst.bitcast = 1 if bitcast else 0
        if fill is not None:
            np_dt = np.dtype(str(dst_t.dtype)[6:])
            jobs[i].dst.fill_rest = 1
            jobs[i].dst.fill_bits = int(np.array([fill], dtype=np_dt).view("u%d" % np_dt.itemsize)[0])
    torch.cuda.synchronize()
    rc = _lib.lib.pgsd_unpack_fields(len(specs), jobs, N, None)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("N", [1, 63, 1000, 1025, 70001])
def test_unpack_fill_makes_whole_rows_of_partial_chunks(N):
    """pgsd_field_dst.fill_rest: velocity without a mass chunk -> (vx, vy, vz, 1.0) rows stored whole (float4 and
    double4 destinations), xy + w chunks with z filled, a lone w chunk with xyz filled; poisoned destinations show
    that every column is written."""
    rng = np.random.default_rng(N + 11)
    vel = G.rand_array(rng, (N, 3), np.float32)
    xy = G.rand_array(rng, (N, 2), np.float32)
    w = G.rand_array(rng, (N, 1), np.float32)
    keep = []
    v4 = torch.full((N, 4), 7.5, dtype=torch.float32, device="cuda")
    d4 = torch.full((N, 4), 7.5, dtype=torch.float64, device="cuda")
    q4 = torch.full((N, 4), 7.5, dtype=torch.float32, device="cuda")
    w4 = torch.full((N, 4), 7.5, dtype=torch.float32, device="cuda")
    unpack_filled([(vel, v4, 3, 0, None, False, 1.0), (vel, d4, 3, 0, None, False, -2.5),
                   (xy, q4, 2, 0, None, False, None), (w, q4, 1, 3, None, False, 0.25),
                   (w, w4, 1, 3, None, False, 0.0)], N, keep)
    got = v4.cpu().numpy()
    assert got[:, :3].tobytes() == vel.tobytes() and (got[:, 3] == 1.0).all()
    got = d4.cpu().numpy()
    assert (got[:, :3] == vel.astype(np.float64)).all() and (got[:, 3] == -2.5).all()
    got = q4.cpu().numpy()
    assert got[:, :2].tobytes() == xy.tobytes() and got[:, 3].tobytes() == w.tobytes() and (got[:, 2] == 0.25).all()
    got = w4.cpu().numpy()
    assert (got[:, :3] == 0.0).all() and got[:, 3].tobytes() == w.tobytes()


def test_unpack_fill_on_the_generic_paths():
    """fill_rest where rows are not assembled by the row kernel: a scatter index, 2-byte elements, five columns"""
    N = 4099
    rng = np.random.default_rng(12)
    vel = G.rand_array(rng, (N, 3), np.float32)
    order_np = rng.permutation(N).astype(np.int32)
    order = dev(order_np)
    s16 = rng.integers(-30000, 30000, size=(N, 2)).astype(np.int16)
    keep = []
    v4 = torch.full((N, 4), 7.5, dtype=torch.float32, device="cuda")
    h5 = torch.full((N, 5), 77, dtype=torch.int16, device="cuda")
    unpack_filled([(vel, v4, 3, 0, order, False, 1.0), (s16, h5, 2, 1, None, False, -3)], N, keep)
    got = v4.cpu().numpy()
    assert got[order_np, :3].tobytes() == vel.tobytes() and (got[:, 3] == 1.0).all()
    got = h5.cpu().numpy()
    assert (got[:, 1:3] == s16).all() and (got[:, 0] == -3).all() and (got[:, 3:] == -3).all()


def test_read_frame_device_scalar4_without_mass_and_typeid_chunks(tmp_gsd):
    """A file that holds position and velocity only: `read_frame_device(scalar4=True)` restores
    pos4 = (x, y, z, type id 0) and vel4 = (vx, vy, vz, mass 1.0) through the chunks' fill, arrays allocated
    uninitialised."""
    import pgsd.hoomd as hoomd
    N = 10_007
    rng = np.random.default_rng(5)
    pos, vel = G.rand_array(rng, (N, 3), np.float32), G.rand_array(rng, (N, 3), np.float32)
    with hoomd.open(tmp_gsd, 'w') as t:
        fr = hoomd.Frame()
        fr.particles.N = N
        fr.particles.position, fr.particles.velocity = pos, vel
        t.append(fr)
    with hoomd.open(tmp_gsd, 'r') as t:
        assert not t.file.chunk_exists(0, 'particles/mass') and not t.file.chunk_exists(0, 'particles/typeid')
        s = t.read_frame_device(0, scalar4=True)
        p4, v4 = s.particles.pos4.cpu().numpy(), s.particles.vel4.cpu().numpy()
        assert p4[:, :3].tobytes() == pos.tobytes() and (p4[:, 3].view(np.uint32) == 0).all()
        assert v4[:, :3].tobytes() == vel.tobytes() and (v4[:, 3] == 1.0).all()
        # a partition of it
        s = t.read_frame_device(0, part=(1000, 2345), scalar4=True)
        v4 = s.particles.vel4.cpu().numpy()
        assert v4.shape == (2345, 4) and v4[:, :3].tobytes() == vel[1000:3345].tobytes() and (v4[:, 3] == 1.0).all()
