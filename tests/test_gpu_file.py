"""End-to-end parity of the device write path: HBM arrays -> HIP pack -> pinned slabs ->
pwrite, through pgsd.fl (C ABI underneath), against (a) the golden file the compiled
reference wrote for the same closed-form data and (b) the CPU oracle on random data."""
import multiprocessing as mp
import os
import uuid

import numpy as np
import pytest

import gpu_common as G
import scenario as S

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def embed4(rows3, w):
    """(N,3) values + (N,) 32-bit payload -> (N,4) HOOMD-style array of the same dtype."""
    out = np.zeros((rows3.shape[0], 4), dtype=rows3.dtype)
    out[:, :3] = rows3
    out[:, 3] = w
    return out


def test_device_path_reproduces_reference_golden_posvelid(tmp_gsd):
    """Same chunk sequence and values as tests/golden/scenarios/posvelid.scn (P=1), but the
    per-particle arrays live on the GPU as float4 (w = type id bits for position)."""
    import pgsd.fl as fl
    N = 1000
    f = fl.open(tmp_gsd, 'w', application='pgsd_amd_test', schema='hoomd', schema_version=[1, 4])
    part = np.array([N])
    for frame, seed in enumerate((1234, 1235, 1236)):
        pos = S.gen_data(9, seed, 0, N, 3)
        vel = S.gen_data(9, seed, 0, N, 3)
        tid = S.gen_data(3, seed, 0, N, 1)
        f.write_chunk('configuration/step', S.gen_data(4, seed, 0, 1, 1), write_all=False)
        if frame == 0:
            f.write_chunk('particles/N', S.gen_data(3, seed, 0, 1, 1), write_all=False)
        dpos = dev(embed4(pos, tid[:, 0].view(np.float32)))
        dvel = dev(embed4(vel, np.float32(1.0)))
        f.write_chunk('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3)), offset=part)
        f.write_chunk('particles/velocity', fl.DeviceField.from_tensor(dvel, columns=(0, 3)), offset=part)
        if frame < 2:
            f.write_chunk('particles/typeid',
                          fl.DeviceField.from_tensor(dpos, columns=(3, 4), out_dtype=np.uint32, bitcast=True),
                          offset=part)
        f.end_frame()
    f.close()
    with open(tmp_gsd, 'rb') as a, open(os.path.join(S.GOLDEN, 'posvelid.p1.gsd'), 'rb') as b:
        assert a.read() == b.read()


def _oracle_frames(path, P, frames):
    """frames: list of lists of (name, type_id, M, all, [per-rank arrays])."""
    import ctypes
    lib = S.oracle_lib()
    rc = ctypes.c_int(0)
    h = lib.oracle_create_and_open(path.encode(), P, b'app', b'hoomd', lib.oracle_make_version(1, 4), 1, 0,
                                   ctypes.byref(rc))
    assert rc.value == 0
    for chunks in frames:
        for name, t, M, all_, arrays in chunks:
            counts = [a.shape[0] for a in arrays]
            if all_:
                row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
                Ng = int(sum(counts))
                r = S.oracle_write_chunk(lib, h, name, t, arrays, M, Ng, M, [int(x) * M for x in row0],
                                         [Ng * M] * P, True)
            else:
                r = S.oracle_write_chunk(lib, h, name, t, arrays, M, counts[0], M, [0] * P,
                                         [c * M for c in counts], False)
            assert r == 0
        assert lib.oracle_end_frame(h) == 0
    assert lib.oracle_close(h) == 0


@pytest.mark.parametrize("N", [1, 777, 200_003])
def test_fused_device_write_matches_oracle(N, tmp_path):
    """write_chunks (one fused launch) with a small staging ring so that chunks span many
    slabs; double4 sources converted to float32 chunks; scalars through the device path."""
    import pgsd.fl as fl
    rng = np.random.default_rng(N)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    f = fl.open(mine, 'w', application='app', schema='hoomd', schema_version=[1, 4])
    f.configure_device(slab_bytes=64 * 1024, n_slabs=3, n_writers=2, profile=True)
    frames = []
    for frame in range(3):
        pos = G.rand_array(rng, (N, 4), np.float64)
        vel = G.rand_array(rng, (N, 4), np.float32)
        img = rng.integers(-3, 3, size=(N, 3), dtype=np.int32)
        dens = G.rand_array(rng, (N,), np.float32)
        step = np.array([frame * 100], dtype=np.uint64)
        dpos, dvel, dimg, ddens = dev(pos), dev(vel), dev(img), dev(dens)
        f.write_chunk('configuration/step', step, write_all=False)
        f.write_chunks([
            ('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3), out_dtype=np.float32)),
            ('particles/velocity', fl.DeviceField.from_tensor(dvel, columns=(0, 3))),
            ('particles/mass', fl.DeviceField.from_tensor(dvel, columns=(3, 4))),
            ('particles/image', dimg),
            ('particles/density', ddens),
        ], offset=np.array([N]))
        f.end_frame()
        frames.append([
            ('configuration/step', 4, 1, False, [step.reshape(1, 1)]),
            ('particles/position', 9, 3, True, [G.oracle_pack(pos, 3, out_dtype=np.float32)]),
            ('particles/velocity', 9, 3, True, [G.oracle_pack(vel, 3)]),
            ('particles/mass', 9, 1, True, [G.oracle_pack(vel, 1, col0=3)]),
            ('particles/image', 7, 3, True, [img]),
            ('particles/density', 9, 1, True, [dens.reshape(-1, 1)]),
        ])
    st = f.device_stats()
    assert st['pack_launches'] == 3 and st['written_bytes'] == 3 * N * (12 + 12 + 4 + 12 + 4)
    f.close()
    _oracle_frames(ref, 1, frames)
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()
    # and the file reads back through the same API
    g = fl.open(mine, 'r')
    assert g.nframes == 3
    np.testing.assert_array_equal(g.read_chunk(2, 'particles/image'), frames[2][4][4][0])
    g.close()


def _rank_main(rank, P, shm, path, counts, seed, q, async_seal=False, batched=False):
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
        sys.path.insert(0, os.path.join(root, "tests"))
        import torch
        import pgsd.fl as fl
        from pgsd import _lib
        __import__("pgsd.dist").dist.init_shm(shm, rank, P)
        torch.cuda.set_device(0)
        n = counts[rank]
        row0 = int(sum(counts[:rank]))
        f = fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4])
        declared = batched == "declared"
        f.frame_exchange = bool(batched) and not declared      # batched: ONE exchange per frame, the partition derived from it
        if declared:                    # pgsd_set_partition: the ranks' row counts are declared, NO exchange per frame
            f.set_partition(counts)
        c_open = f.collective_count
        for frame in range(2):
            pos = S.gen_data(9, seed + frame, row0, n, 4)
            tid = S.gen_data(3, seed + frame, row0, n, 1)
            dpos = torch.from_numpy(pos).cuda()
            dtid = torch.from_numpy(tid.view(np.int32)).cuda()
            f.write_chunk('configuration/step', np.array([frame], dtype=np.uint64), write_all=False)
            f.write_chunks([('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3))),
                            ('particles/typeid', fl.DeviceField.from_tensor(dtid, out_dtype=np.uint32))],
                           offset="auto" if batched else np.array(counts), rank=rank)
            f.end_frame(wait=not async_seal)
        if declared:
            assert f.collective_count == c_open        # two frames, not one collective
        elif batched:
            assert f.collective_count <= 2 + 2 + 1     # create/open, one per frame, the barrier of close (counted after)
        f.close()
        _lib.lib.pgsd_comm_finalize()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("batched", [False, True, "declared"])
@pytest.mark.parametrize("async_seal", [False, True])
@pytest.mark.parametrize("counts", [[600, 401], [0, 333, 1]])
def test_multi_rank_device_write_matches_oracle(counts, async_seal, batched, tmp_path):
    """P processes share cuda:0 (<= 3 ranks), talk through the shm communicator, each packs
    and writes its own partition; the file equals the oracle's P-rank file."""
    P = len(counts)
    seed = 99
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = "pgsdgpu_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_rank_main, args=(r, P, shm, mine, counts, seed, q, async_seal, batched)) for r in range(P)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg in results), results
    frames = []
    row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
    for frame in range(2):
        pos = [S.gen_data(9, seed + frame, int(row0[r]), counts[r], 4)[:, :3].copy() for r in range(P)]
        tid = [S.gen_data(3, seed + frame, int(row0[r]), counts[r], 1) for r in range(P)]
        step = [np.array([[frame]], dtype=np.uint64)] * P
        frames.append([('configuration/step', 4, 1, False, step),
                       ('particles/position', 9, 3, True, pos),
                       ('particles/typeid', 3, 1, True, tid)])
    _oracle_frames(ref, P, frames)
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()


@pytest.mark.parametrize("batched", [False, True, "declared"])
def test_five_ranks_share_the_gpu(batched, tmp_path):
    """as many ranks as the box lets use the card next to the test process itself (6 processes in all): uneven
    partition with an empty and a one-row rank"""
    test_multi_rank_device_write_matches_oracle([5000, 0, 1, 77, 4096], False, batched, tmp_path)


def test_hoomd_append_with_device_fields(tmp_path):
    """pgsd.hoomd.HOOMDTrajectory.append with GPU-resident attributes (one fused launch) equals
    the same trajectory written from host copies."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    N = 12345
    rng = np.random.default_rng(4)
    gpu_path, cpu_path = str(tmp_path / "gpu.gsd"), str(tmp_path / "cpu.gsd")
    tg, tc = hoomd.open(gpu_path, 'w'), hoomd.open(cpu_path, 'w')
    for i in range(3):
        pos4 = G.rand_array(rng, (N, 4), np.float64)
        vel4 = G.rand_array(rng, (N, 4), np.float32)
        tid = rng.integers(0, 3, size=N, dtype=np.uint32)
        vel4[:, 3] = rng.random(N, dtype=np.float32) + 1.0          # mass in w
        dens = G.rand_array(rng, (N,), np.float32)
        dpos, dvel, dtid, ddens = dev(pos4), dev(vel4), dev(tid.view(np.int32)), dev(dens)
        for t, on_gpu in ((tg, True), (tc, False)):
            f = hoomd.Frame()
            f.configuration.step = 7 * i + 1
            f.configuration.box = [20, 20, 20, 0, 0, 0]
            f.particles.N = N
            f.particles.types = ['a', 'b', 'c']
            if on_gpu:
                f.particles.position = fl.DeviceField.from_tensor(dpos, columns=(0, 3), out_dtype=np.float32)
                f.particles.velocity = fl.DeviceField.from_tensor(dvel, columns=(0, 3))
                f.particles.mass = fl.DeviceField.from_tensor(dvel, columns=(3, 4))
                f.particles.typeid = fl.DeviceField.from_tensor(dtid, out_dtype=np.uint32)
                f.particles.density = ddens
            else:
                f.particles.position = pos4[:, :3].astype(np.float32)
                f.particles.velocity = vel4[:, :3]
                f.particles.mass = vel4[:, 3]
                f.particles.typeid = tid
                f.particles.density = dens
            t.append(f)
    tg.close()
    tc.close()
    # what is elided may differ between the two (byte equality on the GPU, array_equal on the host): compare contents
    with hoomd.open(gpu_path, 'r') as a, hoomd.open(cpu_path, 'r') as b:
        assert len(a) == len(b) == 3
        for i in range(3):
            sa, sb = a[i], b[i]
            assert sa.configuration.step == sb.configuration.step and sa.particles.types == sb.particles.types
            for name in ('position', 'velocity', 'mass', 'typeid', 'density'):
                ga, gb = getattr(sa.particles, name), getattr(sb.particles, name)
                assert ga.dtype == gb.dtype and ga.tobytes() == gb.tobytes(), name


def test_rccl_communicator_single_rank():
    """The native RCCL back end (dlopen, ncclCommInitRank, ncclAllGather on the private stream);
    more than one rank needs more than one GPU, so the multi-rank run is bench.py --gpus N."""
    import ctypes
    from pgsd import _lib
    import pgsd.dist as pdist
    uid = (ctypes.c_uint8 * 128)()
    assert _lib.lib.pgsd_comm_rccl_unique_id(uid) == 0, _lib.last_error()
    assert pdist.init_rccl(bytes(uid), 0, 1, 0) == 0, _lib.last_error()
    try:
        counts, row0, n_global = pdist.partition_rows(4242)
        assert list(counts) == [4242] and row0 == 0 and n_global == 4242
        send = (ctypes.c_uint8 * 300)(*range(44, 344 - 44) if False else [i % 251 for i in range(300)])
        recv = (ctypes.c_uint8 * 300)()
        assert _lib.lib.pgsd_comm_allgather(send, recv, 300) == 0
        assert bytes(recv) == bytes(send)
        assert _lib.lib.pgsd_comm_barrier() == 0
    finally:
        pdist.finalize()


def test_full_size_frame_matches_oracle_file(tmp_path):
    """BASELINE config 3 at its real size: 10 M particles, position+velocity+typeid from float4 /
    int32 arrays in HBM, two frames (560 MB of chunks) -- the file must be byte-identical to the
    one the CPU oracle writes from host copies of the same values."""
    import hashlib
    import pgsd.fl as fl
    N = 10_000_000
    d = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    mine = os.path.join(d, "pgsd_full_mine_%d.gsd" % os.getpid())
    ref = os.path.join(d, "pgsd_full_ref_%d.gsd" % os.getpid())
    try:
        g = torch.Generator(device="cuda").manual_seed(1234)
        f = fl.open(mine, 'w', application='app', schema='hoomd', schema_version=[1, 4])
        frames = []
        for frame in range(2):
            pos = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 100.0
            vel = torch.randn((N, 4), generator=g, device="cuda")
            tid = torch.randperm(N, generator=g, device="cuda").to(torch.int32)
            f.write_chunk('configuration/step', np.array([frame], dtype=np.uint64), write_all=False)
            f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                            ('particles/velocity', fl.DeviceField.from_tensor(vel, columns=(0, 3))),
                            ('particles/typeid', fl.DeviceField.from_tensor(tid, out_dtype=np.uint32))],
                           offset=np.array([N]))
            f.end_frame()
            frames.append([('configuration/step', 4, 1, False, [np.array([[frame]], dtype=np.uint64)]),
                           ('particles/position', 9, 3, True, [np.ascontiguousarray(pos.cpu().numpy()[:, :3])]),
                           ('particles/velocity', 9, 3, True, [np.ascontiguousarray(vel.cpu().numpy()[:, :3])]),
                           ('particles/typeid', 3, 1, True, [tid.cpu().numpy().view(np.uint32).reshape(-1, 1)])])
        f.close()
        _oracle_frames(ref, 1, frames)
        assert os.path.getsize(mine) == os.path.getsize(ref)

        def digest(p):
            h = hashlib.sha256()
            with open(p, 'rb') as fh:
                for block in iter(lambda: fh.read(1 << 24), b''):
                    h.update(block)
            return h.hexdigest()
        assert digest(mine) == digest(ref)
    finally:
        for p in (mine, ref):
            if os.path.exists(p):
                os.unlink(p)


def test_two_trajectories_written_concurrently_from_two_threads(tmp_path):
    """A simulation keeps a trajectory and a restart file open at once; here two threads drive
    two handles (two device pipelines) at the same time, asynchronous sealing included. Each file
    must equal the oracle's file for its own frames."""
    import threading
    import pgsd.fl as fl
    results = {}

    def writer(tag, N, n_frames, async_seal):
        try:
            rng = np.random.default_rng(len(tag) + N)
            mine = str(tmp_path / (tag + ".gsd"))
            f = fl.open(mine, 'w', application='app', schema='hoomd', schema_version=[1, 4])
            f.configure_device(slab_bytes=256 * 1024, n_slabs=4)
            frames = []
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for i in range(n_frames):
                    pos = G.rand_array(rng, (N, 4), np.float32)
                    vel = G.rand_array(rng, (N, 4), np.float32)
                    dpos, dvel = dev(pos), dev(vel)
                    step = np.array([i], dtype=np.uint64)
                    f.write_chunk('configuration/step', step, write_all=False)
                    f.write_chunks([('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3))),
                                    ('particles/typeid', fl.DeviceField.from_tensor(dpos, columns=(3, 4), out_dtype=np.uint32,
                                                                                   bitcast=True)),
                                    ('particles/velocity', fl.DeviceField.from_tensor(dvel, columns=(0, 3)))],
                                   offset=np.array([N]))
                    f.end_frame(wait=not async_seal)
                    frames.append([('configuration/step', 4, 1, False, [step.reshape(1, 1)]),
                                   ('particles/position', 9, 3, True, [G.oracle_pack(pos, 3)]),
                                   ('particles/typeid', 3, 1, True, [G.oracle_pack(pos, 1, col0=3, out_dtype=np.uint32, bitcast=True)]),
                                   ('particles/velocity', 9, 3, True, [G.oracle_pack(vel, 3)])])
                f.close()
            results[tag] = (mine, frames)
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            results[tag] = e

    threads = [threading.Thread(target=writer, args=("trajectory", 40_003, 6, True)),
               threading.Thread(target=writer, args=("restart", 150_001, 3, False))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    for tag in ("trajectory", "restart"):
        assert not isinstance(results[tag], Exception), results[tag]
        mine, frames = results[tag]
        ref = str(tmp_path / (tag + "_oracle.gsd"))
        _oracle_frames(ref, 1, frames)
        with open(mine, 'rb') as a, open(ref, 'rb') as b:
            assert a.read() == b.read(), tag


RCCL_FROM_TORCH = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29%%03d" %% (os.getpid() %% 1000))
torch.cuda.set_device(0)
dist.init_process_group(backend=sys.argv[1], rank=0, world_size=1,
                        **({"device_id": torch.device("cuda", 0)} if sys.argv[1] == "nccl" else {}))
import pgsd.dist as pdist
name = pdist.init_from_torch(device=0, _single_rank_too=True, prefer_rccl=sys.argv[2] == "1")
counts, row0, n = pdist.partition_rows(12345)
print(name, [int(c) for c in counts], row0, n)
pdist.finalize()
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("backend,prefer,expect", [("nccl", "1", "rccl"), ("nccl", "0", "torch-nccl"), ("gloo", "1", "torch-gloo")])
def test_communicator_from_a_torch_process_group(backend, prefer, expect):
    """pgsd.dist.init_from_torch end to end in a one-rank group: the unique id travels through a
    torch broadcast, the library builds its own RCCL communicator next to PyTorch's, the self-check
    exchange and the cross-rank agreement run, and the row-count allgather works through it (the
    multi-rank case needs more GPUs than the test box has). Second case: the fallback bench.py takes when
    that communicator cannot be built -- host callbacks into torch.distributed on the same nccl group."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", RCCL_FROM_TORCH % os.path.join(root, "pgsd-sph_amd"), backend, prefer],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.strip().splitlines()[-1] == "%s [12345] 0 12345" % expect, p.stdout
