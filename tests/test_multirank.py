"""N > 1 ranks on CPU: world_size-2 (and 3) `gloo` process groups drive the product through
pgsd.fl / pgsd.hoomd with the torch.distributed communicator back end; the file must equal the
CPU oracle's P-rank file (duplicated small chunks, per-rank offsets and all)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

import scenario as S

torch = pytest.importorskip("torch")
import torch.multiprocessing as tmp_mp  # noqa: E402


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, path, counts, mode):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pgsd.dist as pdist
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    import scenario as S
    assert pdist.init_from_torch() == "torch-gloo"
    n = counts[rank]
    got_counts, row0, n_global = pdist.partition_rows(n)
    assert list(got_counts) == counts and row0 == sum(counts[:rank]) and n_global == sum(counts)
    if mode == "fl":
        f = fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4])
        for frame in range(3):
            seed = 40 + frame
            f.write_chunk('configuration/step', np.array([frame], dtype=np.uint64), write_all=False)
            f.write_chunk('particles/N', np.array([n_global], dtype=np.uint32), write_all=False)
            f.write_chunk('particles/position', S.gen_data(9, seed, row0, n, 3), offset=got_counts, rank=rank)
            f.write_chunk('particles/typeid', S.gen_data(3, seed, row0, n, 1), offset=got_counts, rank=rank)
            f.end_frame()
            # replicated metadata: every rank can look chunks up (the reference cannot)
            assert f.chunk_exists(frame, 'particles/position')
            assert f.read_chunk(frame, 'particles/N')[0] == n_global
        f.close()
    elif mode == "fl_defaults":
        # write_chunk(name, data) exactly as the reference's binding is called by default
        # (fl.pyx:526 write_all=True, offset=None, rank=0): replicated rows on the direct path
        f = fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4])
        for frame in range(2):
            f.write_chunk('log/energy', S.gen_data(9, 70 + frame, 0, 10, 3))
            f.write_chunk('log/count', S.gen_data(3, 80 + frame, 0, 7, 1)[:, 0])
            f.write_chunk('particles/position', S.gen_data(9, 90 + frame, row0, n, 3), offset=got_counts, rank=rank)
            f.end_frame()
        f.close()
    elif mode == "hoomd_default_rank":
        # one rank's host data equals the DEFAULT value (zero velocity, typeid 0), the other rank's does not,
        # over more frames than it takes for a batched seal to be pending: the elision
        # test's frame-0 lookup must not be a collective that only some ranks enter
        t = hoomd.open(path, 'w')
        for frame in range(5):
            seed = 160 + frame
            fr = hoomd.Frame()
            fr.configuration.step = frame
            fr.particles.N = n
            fr.particles.position = S.gen_data(9, seed, row0, n, 3)
            if rank == 0:
                fr.particles.velocity = np.zeros((n, 3), dtype=np.float32)
                fr.particles.typeid = np.zeros(n, dtype=np.uint32)
            else:
                fr.particles.velocity = S.gen_data(9, seed + 100, row0, n, 3)
                fr.particles.typeid = S.gen_data(3, seed, row0, n, 1)[:, 0] % 3 + 1
            t.append(fr)
        t.close()
    elif mode == "fl_declared_mismatch":
        # a declared partition takes the sizes of chunks that are not partitioned on trust; here the ranks bring
        # different lengths: the next synchronisation point tells every rank
        f = fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4])
        f.set_partition(counts)
        f.write_chunk('particles/position', S.gen_data(9, 1, row0, n, 3), offset="auto")
        f.write_chunk('log/values', np.zeros(4 + rank))
        f.end_frame()
        try:
            f.flush()
            outcome = "flushed"
        except RuntimeError as e:
            outcome = "RuntimeError" if "disagree about the file (file size" in str(e) else "RuntimeError? " + str(e)
        with open(path + ".rank%d" % rank, "w") as out:
            out.write(outcome)
        try:
            f.close()
        except RuntimeError:
            pass
    elif mode == "hoomd_partition_change":
        # frame 1 has another partition (known only from the frame's allgather); rank 0's own count is
        # unchanged, it elides position and velocity against ITS rows of frame 0 -- the other ranks, whose counts
        # changed, cannot compare (and leave the velocity unset).  Whether those chunks are written must be decided
        # alike on every rank
        t = hoomd.open(path, 'w')
        c1 = [counts[0]] + [c + d for c, d in zip(counts[1:], [1, -1] + [0] * (world - 3))]
        for frame, cs in enumerate((counts, c1, c1)):
            n1, r1 = cs[rank], sum(cs[:rank])
            fr = hoomd.Frame()
            fr.configuration.step = frame
            fr.particles.N = n1
            fr.particles.position = S.gen_data(9, 400, r1, n1, 3)
            if frame == 0 or rank == 0:
                fr.particles.velocity = S.gen_data(9, 401, r1, n1, 3)
            t.append(fr)
        t.close()
    elif mode == "hoomd_log_shapes_differ":
        t = hoomd.open(path, 'w')
        fr = hoomd.Frame()
        fr.configuration.step = 0
        fr.particles.N = n
        fr.particles.position = S.gen_data(9, 500, row0, n, 3)
        fr.log['energy'] = np.zeros(3 + (rank == world - 1))            # one rank brings a longer array
        try:
            t.append(fr)
            outcome = "appended"
        except ValueError as e:
            outcome = "ValueError" if "the ranks differ" in str(e) else "ValueError?"
        with open(path + ".rank%d" % rank, "w") as out:
            out.write(outcome)
        t.file.end_frame()          # every rank left the frame at the same point: the file goes on
        t.close()
    elif mode == "hoomd_lone_compare":
        # frame 1 writes nothing but the buffered step chunk (every array elided), so pgsd_end_frame does
        # not flush and metadata stays pending; in frame 2 rank 0 ALONE compares an array it has not compared before
        # (velocity: None on the other ranks) -- the lookup and read of frame 0's rows must stay local
        t = hoomd.open(path, 'w')
        pos = S.gen_data(9, 300, row0, n, 3)
        vel = S.gen_data(9, 301, row0, n, 3)
        for frame in range(4):
            fr = hoomd.Frame()
            fr.configuration.step = frame
            fr.particles.N = n
            fr.part_dist = np.array(counts)
            fr.particles.position = pos
            if frame == 0 or (frame >= 2 and rank == 0):
                fr.particles.velocity = vel
            t.append(fr)
        t.close()
    else:
        t = hoomd.open(path, 'w')
        for frame in range(2):
            seed = 60 + frame
            fr = hoomd.Frame()
            fr.configuration.step = 5 + frame
            fr.configuration.box = [3, 3, 3, 0, 0, 0]
            fr.particles.N = n
            fr.particles.types = ['A', 'B']
            fr.particles.position = S.gen_data(9, seed, row0, n, 3)
            fr.particles.velocity = S.gen_data(9, seed + 100, row0, n, 3)
            fr.particles.typeid = S.gen_data(3, seed, row0, n, 1)[:, 0] % 2
            if rank == 0:
                fr.particles.density = S.gen_data(9, seed + 200, row0, n, 1)[:, 0]   # only rank 0 sets it
            t.append(fr)
        t.close()
    pdist.finalize()
    dist.destroy_process_group()


def _oracle_file(path, P, frames):
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
                r = S.oracle_write_chunk(lib, h, name, t, arrays, M, Ng, M, [int(x) * M for x in row0], [Ng * M] * P,
                                         True)
            else:
                r = S.oracle_write_chunk(lib, h, name, t, arrays, M, counts[0], M, [0] * P, [c * M for c in counts],
                                         False)
            assert r == 0
        assert lib.oracle_end_frame(h) == 0
    assert lib.oracle_close(h) == 0


@pytest.mark.parametrize("counts", [[5, 9], [4, 0, 7]])
def test_fl_two_and_three_ranks_match_oracle(counts, tmp_path):
    P = len(counts)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    tmp_mp.spawn(_worker, args=(P, free_port(), mine, counts, "fl"), nprocs=P, join=True)
    row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
    Ng = sum(counts)
    frames = []
    for frame in range(3):
        seed = 40 + frame
        frames.append([
            ('configuration/step', 4, 1, False, [np.array([[frame]], dtype=np.uint64)] * P),
            ('particles/N', 3, 1, False, [np.array([[Ng]], dtype=np.uint32)] * P),
            ('particles/position', 9, 3, True, [S.gen_data(9, seed, int(row0[r]), counts[r], 3) for r in range(P)]),
            ('particles/typeid', 3, 1, True, [S.gen_data(3, seed, int(row0[r]), counts[r], 1) for r in range(P)]),
        ])
    _oracle_file(ref, P, frames)
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()


@pytest.mark.parametrize("counts", [[5, 9], [4, 0, 7]])
def test_fl_write_chunk_default_arguments_match_oracle(counts, tmp_path):
    """`write_chunk(name, data)` with the binding's default arguments on every rank: all=1, offset 0,
    N_global = N, global_size = N*M (fl.pyx:592-598, 640-652).  The reference advances the file by the
    SUM of the ranks' sizes (pgsd.c:2240-2246), P copies' worth for replicated data; round 1's
    product trusted global_size and diverged at P > 1.  The oracle's handling
    of this call shape is pinned to reference-written files by tests/golden/scenarios/defaultargs.scn."""
    P = len(counts)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    tmp_mp.spawn(_worker, args=(P, free_port(), mine, counts, "fl_defaults"), nprocs=P, join=True)
    row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
    lib = S.oracle_lib()
    rc = ctypes.c_int(0)
    h = lib.oracle_create_and_open(ref.encode(), P, b'app', b'hoomd', lib.oracle_make_version(1, 4), 1, 0,
                                   ctypes.byref(rc))
    assert rc.value == 0
    sizes = []
    for frame in range(2):
        e = S.gen_data(9, 70 + frame, 0, 10, 3)
        c = S.gen_data(3, 80 + frame, 0, 7, 1)
        assert S.oracle_write_chunk(lib, h, 'log/energy', 9, [e] * P, 3, 10, 3, [0] * P, [30] * P, True) == 0
        assert S.oracle_write_chunk(lib, h, 'log/count', 3, [c] * P, 1, 7, 1, [0] * P, [7] * P, True) == 0
        pos = [S.gen_data(9, 90 + frame, int(row0[r]), counts[r], 3) for r in range(P)]
        Ng = sum(counts)
        assert S.oracle_write_chunk(lib, h, 'particles/position', 9, pos, 3, Ng, 3, [int(x) * 3 for x in row0],
                                    [Ng * 3] * P, True) == 0
        assert lib.oracle_end_frame(h) == 0
        sizes.append(lib.oracle_get_file_size(h))
    assert lib.oracle_close(h) == 0
    # P copies' worth of the replicated chunks per frame: (120 + 28) * P + 12 * Ng
    assert sizes[0] == 5376 + (120 + 28) * P + 12 * sum(counts)
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()


def test_hoomd_append_two_ranks(tmp_path):
    """HOOMDTrajectory.append across two ranks: partitioned arrays land in global order, ranks
    agree on which chunks exist (rank 1 contributes defaults for a field only rank 0 set)."""
    counts = [6, 3]
    mine = str(tmp_path / "traj.gsd")
    tmp_mp.spawn(_worker, args=(2, free_port(), mine, counts, "hoomd"), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgsd-sph_amd"))
    import pgsd.hoomd as hoomd
    with hoomd.open(mine, 'r') as t:
        assert len(t) == 2
        for frame in range(2):
            seed = 60 + frame
            s = t[frame]
            assert s.particles.N == 9 and s.configuration.step == 5 + frame
            np.testing.assert_array_equal(s.particles.position, S.gen_data(9, seed, 0, 9, 3))
            np.testing.assert_array_equal(s.particles.velocity, S.gen_data(9, seed + 100, 0, 9, 3))
            np.testing.assert_array_equal(s.particles.typeid, S.gen_data(3, seed, 0, 9, 1)[:, 0] % 2)
            dens = S.gen_data(9, seed + 200, 0, 9, 1)[:, 0]
            np.testing.assert_array_equal(s.particles.density[:6], dens[:6])
            np.testing.assert_array_equal(s.particles.density[6:], np.zeros(3, dtype=np.float32))


@pytest.mark.parametrize("counts", [[6, 3], [4, 0, 5]])
def test_hoomd_append_rank_with_default_valued_fields(counts, tmp_path):
    """A rank whose slice is all-default (zero velocity, typeid 0, or no particles at all) next to a rank
    whose slice is not, over five frames: from the third append on round 2's `_should_write` issued a
    collective (the flush inside `chunk_exists`) on the all-default rank only -- PGSD_ERROR_COMM over shm and
    gloo, a hang over RCCL / MPI."""
    P = len(counts)
    mine = str(tmp_path / "traj.gsd")
    tmp_mp.spawn(_worker, args=(P, free_port(), mine, counts, "hoomd_default_rank"), nprocs=P, join=True)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgsd-sph_amd"))
    import pgsd.hoomd as hoomd
    Ng, n0 = sum(counts), counts[0]
    with hoomd.open(mine, 'r') as t:
        assert len(t) == 5
        for frame in range(5):
            seed = 160 + frame
            s = t[frame]
            assert s.particles.N == Ng and s.configuration.step == frame
            np.testing.assert_array_equal(s.particles.position, S.gen_data(9, seed, 0, Ng, 3))
            vel = S.gen_data(9, seed + 100, 0, Ng, 3)
            vel[:n0] = 0
            np.testing.assert_array_equal(s.particles.velocity, vel)
            tid = S.gen_data(3, seed, 0, Ng, 1)[:, 0] % 3 + 1
            tid[:n0] = 0
            np.testing.assert_array_equal(s.particles.typeid, tid)


@pytest.mark.parametrize("counts", [[6, 3], [4, 2, 5]])
def test_hoomd_append_one_rank_alone_compares_an_array_while_metadata_is_pending(counts, tmp_path):
    """`_read_frame0_rows` is a local read, but the lookup in front of it ran the collective
    flush whenever metadata was pending -- which is what a frame with every array elided leaves behind.  Rank 0 alone
    compares its velocities in frames 2 and 3: the run must finish (it hung on RCCL / MPI, PGSD_ERROR_COMM on shm and
    gloo) and the file hold frame 0's arrays once."""
    P = len(counts)
    mine = str(tmp_path / "traj.gsd")
    tmp_mp.spawn(_worker, args=(P, free_port(), mine, counts, "hoomd_lone_compare"), nprocs=P, join=True)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgsd-sph_amd"))
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    Ng = sum(counts)
    with hoomd.open(mine, 'r') as t:
        assert len(t) == 4
        for frame in range(4):
            s = t[frame]
            assert s.particles.N == Ng and s.configuration.step == frame
            np.testing.assert_array_equal(s.particles.position, S.gen_data(9, 300, 0, Ng, 3))
            np.testing.assert_array_equal(s.particles.velocity, S.gen_data(9, 301, 0, Ng, 3))
    f = fl.open(mine, 'r')
    for frame in (1, 2, 3):                     # everything but the step was elided
        assert not f.chunk_exists(frame, 'particles/position') and not f.chunk_exists(frame, 'particles/velocity')
    f.close()


def test_hoomd_append_partition_change_is_decided_alike_on_every_rank(tmp_path):
    """after the partition changed, `_elision_outcome` rewrote plan entries to "write" on the ranks
    that had compared (a rank-local condition) -- one rank placed a chunk the others did not.  Now the ranks carry
    "elided against my rows of frame 0" in the vote and decide from the gathered bits: the chunks are written by all,
    the ranks without a velocity contribute default rows."""
    counts = [4, 2, 5]
    P = len(counts)
    mine = str(tmp_path / "traj.gsd")
    tmp_mp.spawn(_worker, args=(P, free_port(), mine, counts, "hoomd_partition_change"), nprocs=P, join=True)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgsd-sph_amd"))
    import pgsd.hoomd as hoomd
    Ng = sum(counts)
    with hoomd.open(mine, 'r') as t:
        assert len(t) == 3
        vel = S.gen_data(9, 401, 0, Ng, 3)
        for frame in range(3):
            s = t[frame]
            assert s.particles.N == Ng
            np.testing.assert_array_equal(s.particles.position, S.gen_data(9, 400, 0, Ng, 3))
            want = vel.copy()
            if frame > 0:
                want[counts[0]:] = 0                    # only rank 0 set it: the others' rows are the default
                assert t.file.chunk_exists(frame, 'particles/velocity') and t.file.chunk_exists(frame, 'particles/position')
            np.testing.assert_array_equal(s.particles.velocity, want)


def test_hoomd_append_replicated_shapes_must_agree(tmp_path):
    """with a declared partition a log / state / replicated chunk is placed from this rank's own
    size; a rank that brings another shape would silently shift its replicated offsets.  The shapes ride in the vote
    (a digest): every rank raises the same ValueError, nobody has written anything of the frame."""
    counts = [3, 4]
    mine = str(tmp_path / "traj.gsd")
    tmp_mp.spawn(_worker, args=(2, free_port(), mine, counts, "hoomd_log_shapes_differ"), nprocs=2, join=True)
    assert [open(mine + ".rank%d" % r).read() for r in range(2)] == ["ValueError", "ValueError"]


def test_declared_partition_ranks_that_disagree_about_the_file_are_told(tmp_path):
    """ADVICE r3 (low), C-ABI side: with pgsd_set_partition a chunk that is not partitioned is placed from this rank's
    own size; ranks that bring different sizes end up with different layouts of one file.  The flush's status exchange
    now carries every rank's view of the file (size, frame counter, names, index entries) and all ranks fail alike."""
    counts = [3, 4]
    mine = str(tmp_path / "traj.gsd")
    tmp_mp.spawn(_worker, args=(2, free_port(), mine, counts, "fl_declared_mismatch"), nprocs=2, join=True)
    assert [open(mine + ".rank%d" % r).read() for r in range(2)] == ["RuntimeError", "RuntimeError"]
