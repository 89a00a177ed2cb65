"""The batched frame exchange (pgsd_set_frame_exchange): how many collectives a frame costs.

A fake two-rank communicator lives in this process: rank 0 is real, the allgather callback answers for a
peer that sends exactly what rank 0 sends (same calls, same sizes) and counts the calls.  The layout side
of the batching -- files byte-identical to the reference's -- is covered by the golden and fuzz suites
(tests/test_product_golden.py, tests/test_fuzz_parity.py: every scenario replayed with `batch 1`)."""
import ctypes

import numpy as np
import pytest

from pgsd import _lib
import pgsd.fl as fl


FRAME_MESSAGE = 512     # bytes per rank of a frame exchange (64 words: status, count, 62 sizes), pgsd_placement.cpp


class MirrorComm:
    """rank 0 of 2; the peer mirrors rank 0's contribution"""

    def __init__(self, peer_empty=False):
        self.calls = []

        def allgather(ctx, send, recv, nbytes):
            self.calls.append(nbytes)
            ctypes.memmove(recv, send, nbytes)
            ctypes.memmove(recv + nbytes, send, nbytes)
            if peer_empty and nbytes >= 512:
                # a frame exchange [status, count, sizes...] (512 bytes; the flush's status exchange, 40 bytes, carries
                # the ranks' views of the file, which are the same): the peer made the same calls with no rows
                ctypes.memset(recv + nbytes + 16, 0, nbytes - 16)
            return 0

        self._ag = _lib.ALLGATHER_FN(allgather)
        comm = _lib.Comm()
        comm.ctx = None
        comm.rank, comm.size = 0, 2
        comm.allgather = self._ag
        assert _lib.lib.pgsd_comm_set_default(ctypes.byref(comm)) == 0
        self._comm = comm

    def close(self):
        _lib.lib.pgsd_comm_finalize()


@pytest.fixture
def mirror():
    m = MirrorComm()
    yield m
    m.close()


def small_frame(f, i):
    f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
    f.write_chunk("configuration/box", np.arange(6, dtype=np.float32), write_all=False)
    f.write_chunk("log/e", np.array([0.5 * i]), write_all=False)


@pytest.mark.parametrize("batched", [False, True])
def test_collectives_per_frame_host_chunks(batched, mirror, tmp_gsd):
    f = fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4])
    f.frame_exchange = batched
    base = f.collective_count
    per_frame = []
    for i in range(4):
        small_frame(f, i)
        f.write_chunk("particles/position", np.full((5, 3), i, np.float32), offset=np.array([5, 5]), rank=0)
        f.end_frame()
        per_frame.append(f.collective_count - base - sum(per_frame))
    # unbatched: one exchange per chunk + the status exchange of the flush; batched: the per-particle host
    # chunk needs its offset at once and carries the three queued chunks with it -- one collective per frame
    assert per_frame == ([1, 1, 1, 1] if batched else [5, 5, 5, 5])
    assert f.collective_count == len(mirror.calls)       # the two exchanges of create/open included
    f.close()           # the barrier the batched frames skipped is made up here (plus the final verdict)
    assert f is not None


def test_batched_frames_of_small_chunks_exchange_once_and_match_the_unbatched_file(mirror, tmp_path):
    paths = []
    for batched in (False, True):
        p = str(tmp_path / ("b%d.gsd" % batched))
        f = fl.open(p, "w", application="app", schema="hoomd", schema_version=[1, 4])
        f.frame_exchange = batched
        n0 = len(mirror.calls)
        for i in range(5):
            small_frame(f, i)
            f.end_frame()
        if batched:
            assert len(mirror.calls) - n0 == 5          # one exchange per frame (3 sizes + status + count)
            assert mirror.calls[-1] == FRAME_MESSAGE    # ... of the fixed length, whatever is queued
        else:
            assert len(mirror.calls) - n0 == 15         # one per chunk; small chunks alone never flush
        f.close()
        paths.append(p)
    with open(paths[0], "rb") as a, open(paths[1], "rb") as b:
        assert a.read() == b.read()


def test_frame_exchange_messages_have_one_length_whatever_is_queued(mirror, tmp_path):
    """ncclAllGather (and MPI_Allgather) need equal send counts on all ranks: the frame exchange never sends a
    message whose length depends on what THIS rank queued.  0, 1, 5, 62 chunks: one
    512-byte message; 63, 130 chunks: further rounds of the same length; file == the unbatched one."""
    for n_chunks in (1, 5, 62, 63, 130):
        paths = []
        for batched in (False, True):
            p = str(tmp_path / ("m%d_%d.gsd" % (n_chunks, batched)))
            f = fl.open(p, "w", application="app", schema="hoomd", schema_version=[1, 4])
            f.frame_exchange = batched
            n0 = len(mirror.calls)
            for fr in range(2):
                for c in range(n_chunks):
                    f.write_chunk("log/q%03d" % c, np.arange(c % 7 + 1, dtype=np.float32) + fr, write_all=False)
                f.end_frame()
            if batched:
                rounds = 1 if n_chunks <= 62 else 1 + -(-(n_chunks - 62) // 64)
                assert mirror.calls[n0:] == [FRAME_MESSAGE] * (2 * rounds), (n_chunks, mirror.calls[n0:])
            f.close()
            paths.append(p)
        with open(paths[0], "rb") as a, open(paths[1], "rb") as b:
            assert a.read() == b.read(), n_chunks


def test_ranks_that_queued_different_numbers_of_chunks_are_told_apart_by_content(tmp_path):
    """A peer that queued one chunk fewer still sends a well-formed message of the common length; the count
    word gives it away and the call fails with PGSD_ERROR_COMM instead of an allgather of unequal sizes."""
    lengths = []

    def allgather(ctx, send, recv, nbytes):
        lengths.append(nbytes)
        ctypes.memmove(recv, send, nbytes)
        ctypes.memmove(recv + nbytes, send, nbytes)
        if nbytes == FRAME_MESSAGE:
            words = (ctypes.c_uint64 * 64).from_address(recv + nbytes)
            words[1] -= 1                      # the peer's count
        return 0

    ag = _lib.ALLGATHER_FN(allgather)
    comm = _lib.Comm()
    comm.ctx, comm.rank, comm.size, comm.allgather = None, 0, 2, ag
    assert _lib.lib.pgsd_comm_set_default(ctypes.byref(comm)) == 0
    try:
        f = fl.open(str(tmp_path / "k.gsd"), "w", application="app", schema="hoomd", schema_version=[1, 4])
        f.frame_exchange = True
        small_frame(f, 0)
        with pytest.raises(RuntimeError, match="different numbers of chunks"):
            f.end_frame()
        assert lengths[-1] == FRAME_MESSAGE
        stats = f.exchange_stats()
        assert stats["count"] == f.collective_count and stats["max_us"] >= 0 and stats["total_us"] >= stats["max_us"]
    finally:
        _lib.lib.pgsd_comm_finalize()


def test_auto_partition_equals_the_callers_allgather(mirror, tmp_path):
    """offset='auto' (PGSD_PARTITION_AUTO): global row count and first row out of the chunk's own exchange."""
    data = np.arange(21, dtype=np.float32).reshape(7, 3)
    files = []
    for auto in (False, True):
        p = str(tmp_path / ("a%d.gsd" % auto))
        f = fl.open(p, "w", application="app", schema="hoomd", schema_version=[1, 4])
        f.write_chunk("particles/position", data, offset="auto" if auto else np.array([7, 7]), rank=0)
        f.end_frame()
        assert f.read_chunk(0, "particles/position").shape == (14, 3)
        f.close()
        files.append(p)
    with open(files[0], "rb") as a, open(files[1], "rb") as b:
        assert a.read() == b.read()


def test_mirror_lags_until_the_exchange(tmp_gsd):
    """Batched: file_size of the handle mirror moves when the queue is resolved, not per call."""
    f = fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4])
    f.frame_exchange = True
    small_frame(f, 0)
    assert f.nframes == 0
    f.exchange_now()
    f.end_frame()
    assert f.nframes == 1 and f.chunk_exists(0, "log/e")
    f.frame_exchange = False
    assert not f.frame_exchange
    f.close()


@pytest.fixture
def empty_peer():
    m = MirrorComm(peer_empty=True)
    yield m
    m.close()


@pytest.mark.gpu
def test_one_collective_per_device_frame(empty_peer, tmp_gsd):
    """The bench / HOOMD-SPH frame: a replicated step chunk, three per-particle chunks packed by one fused
    launch with the partition taken from the exchange, end_frame -- ONE allgather (it goes out after the
    pack launch, so on the RCCL back end the ncclAllGather overlaps the kernel)."""
    torch = pytest.importorskip("torch")
    mirror = empty_peer
    N = 5000
    pos = torch.randn((N, 4), device="cuda")
    vel = torch.randn((N, 4), device="cuda")
    f = fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4])
    f.frame_exchange = True
    fields = [("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
              ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
              ("particles/typeid", fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True))]
    for i in range(3):
        n0 = len(mirror.calls)
        f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
        f.write_chunks(fields, offset="auto")
        f.end_frame()
        assert len(mirror.calls) - n0 == 1
        assert mirror.calls[-1] == FRAME_MESSAGE        # status, count, four chunk sizes, padding
    f.close()
    g = fl.open(tmp_gsd, "r")
    got = g.read_chunk(2, "particles/position")
    assert got.shape == (N, 3)                          # the peer made the same calls with no rows of its own
    np.testing.assert_array_equal(got, pos[:, :3].cpu().numpy())
    assert g.read_chunk(1, "configuration/step")[0] == 1
    g.close()


def test_a_handle_keeps_its_communicator_alive_across_finalize(mirror, tmp_gsd):
    """pgsd_comm_finalize / re-initialisation while a file is open: the handle goes on using (and keeps
    alive) the communicator it was opened with; the context is destroyed with the last handle, not under it."""
    f = fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4])
    small_frame(f, 0)
    f.end_frame()
    assert _lib.lib.pgsd_comm_finalize() == 0 and _lib.lib.pgsd_comm_size() == 1
    n0 = len(mirror.calls)
    small_frame(f, 1)
    f.write_chunk("particles/position", np.zeros((4, 3), np.float32), offset=np.array([4, 4]), rank=0)
    f.end_frame()
    f.close()
    assert len(mirror.calls) > n0                      # still the two-rank communicator of the open
    g = fl.open(tmp_gsd, "r")                          # a new handle sees the new (single-rank) default
    assert g.nframes == 2 and g.read_chunk(1, "particles/position").shape == (8, 3)
    g.close()


@pytest.mark.gpu
def test_async_frames_do_not_pin_their_source_tensors(tmp_gsd):
    """A long run of end_frame(wait=False) without frame_sync: only the newest frames' device fields are kept
    alive (their pack kernels may still read them); older ones are released, synchronous calls release all."""
    torch = pytest.importorskip("torch")
    N = 2000
    f = fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4])
    import weakref
    refs = []
    for i in range(12):
        t = torch.randn((N, 4), device="cuda")
        refs.append(weakref.ref(t))
        f.write_chunks([("particles/position", fl.DeviceField.from_tensor(t, columns=(0, 3)))], offset=np.array([N]))
        f.end_frame(wait=False)
        del t
    import gc
    gc.collect()
    assert f._async_frames_kept() <= 2
    assert sum(r() is not None for r in refs) <= 2
    f.flush()
    gc.collect()
    assert f._async_frames_kept() == 0 and all(r() is None for r in refs)
    f.close()
    g = fl.open(tmp_gsd, "r")
    assert g.nframes == 12
    g.close()


def _bad_args_rank(rank, shm, path, batched, q):
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
        from pgsd import _lib as L
        import pgsd.fl as F
        __import__("pgsd.dist").dist.init_shm(shm, rank, 2)
        f = F.open(path, "w", application="app", schema="hoomd", schema_version=[1, 4])
        f.frame_exchange = batched
        h = f._h()
        data = np.arange(15, dtype=np.float32)
        out = []
        # rank 1 passes M = 0 (pgsd.c:2096-2099: invalid argument) for a per-particle chunk, rank 0 is fine
        M = 0 if rank == 1 else 3
        out.append(L.lib.pgsd_write_chunk(h, b"particles/position", 9, 5, M, 10, M, 5 * rank * 3, 30, True, 0,
                                          data.ctypes.data))
        # ... and a NULL data pointer for a replicated chunk
        ptr = None if rank == 1 else data.ctypes.data
        out.append(L.lib.pgsd_write_chunk(h, b"log/x", 9, 3, 1, 3, 1, 0, 3, False, 0, ptr))
        out.append(L.lib.pgsd_end_frame(h))
        # the next frame is healthy on both ranks
        f.write_chunk("particles/position", data.reshape(5, 3), offset="auto")
        f.end_frame()
        ok = f.chunk_exists(1, "particles/position") and not f.chunk_exists(0, "particles/position")
        f.close()
        L.lib.pgsd_comm_finalize()
        q.put((rank, out, bool(ok)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), False))
        raise


@pytest.mark.parametrize("batched", [False, True])
def test_bad_arguments_on_one_rank_fail_the_call_everywhere_without_a_hang(batched, tmp_gsd):
    """The reference returns from its argument checks before the first collective (pgsd.c:2090-2105): the
    rank with the bad argument leaves, the others wait in MPI_Barrier for ever.  Here the verdict travels with
    the size exchange: the per-particle chunk fails on BOTH ranks, in either exchange mode, nothing hangs and
    the file goes on."""
    import multiprocessing as mp
    import uuid
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = "pgsdbad_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_bad_args_rank, args=(r, shm, tmp_gsd, batched, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    for rank, out, ok in results:
        assert isinstance(out, list), out
        assert out[0] == -2, (rank, out)                      # PGSD_ERROR_INVALID_ARGUMENT on both ranks
        assert ok, (rank, out)
    # the replicated chunk: the rank with the NULL pointer knows at once; everyone by the end of the frame
    r0, r1 = results[0][1], results[1][1]
    assert r1[1] == -2 and (r0[1] == -2 or r0[2] == -2)


def test_hoomd_append_costs_one_collective_per_frame(mirror, tmp_path):
    """`HOOMDTrajectory.append` at P > 1: ONE collective per frame -- its own allgather of row counts and write/skip
    votes -- however many host arrays, state and log chunks the frame holds: with the counts in hand the partition is
    declared to the library (`pgsd_set_partition`), which places every chunk without an exchange of its own.
    (Round 2: two, plus one per host per-particle chunk and per log chunk.)"""
    import pgsd.hoomd as hoomd
    t = hoomd.open(str(tmp_path / "h.gsd"), "w")
    rng = np.random.default_rng(1)
    per_frame = []
    for i in range(4):
        fr = hoomd.Frame()
        fr.configuration.step = i
        fr.particles.N = 50
        fr.particles.position = rng.random((50, 3), dtype=np.float32)
        fr.particles.velocity = rng.random((50, 3), dtype=np.float32)
        fr.particles.typeid = rng.integers(0, 3, size=50).astype(np.uint32)
        fr.particles.density = rng.random(50, dtype=np.float32)
        fr.log["energy"] = np.array([1.5 * i])
        fr.log["virial"] = np.arange(6, dtype=np.float64) + i
        fr.state["hpmc/d"] = np.array([0.1 * i])
        n0 = len(mirror.calls)
        t.append(fr)
        per_frame.append(len(mirror.calls) - n0)
    # (the second append reads frame 0 back as its elision reference: a read is a synchronisation point and
    # makes up the barrier the frames before it skipped -- once per trajectory)
    assert per_frame == [1, 2, 1, 1], per_frame
    t.close()
    with hoomd.open(str(tmp_path / "h.gsd"), "r") as r:
        assert len(r) == 4 and r[3].particles.N == 100                 # the mirrored peer wrote the same rows
        assert r[2].log["energy"][0] == 3.0


def test_declared_partition_frames_cost_no_collective_and_match_the_exchanged_file(mirror, tmp_path):
    """`pgsd_set_partition`: the caller declares every rank's row count once; frames of replicated chunks,
    per-particle host arrays (offset='auto') and default-argument chunks are then placed with NO exchange, and the file
    is the one the per-chunk exchanges produce."""
    rng = np.random.default_rng(2)
    data = [(rng.random((7, 3), dtype=np.float32), rng.integers(0, 5, size=7).astype(np.uint32), rng.random(4)) for _ in range(3)]
    paths = []
    for declared in (False, True):
        p = str(tmp_path / ("d%d.gsd" % declared))
        f = fl.open(p, "w", application="app", schema="hoomd", schema_version=[1, 4])
        if declared:
            f.set_partition([7, 7])                     # the mirrored peer brings the same rows
        per_frame = []
        for i, (pos, tid, log) in enumerate(data):
            n0 = len(mirror.calls)
            small_frame(f, i)
            f.write_chunk("particles/position", pos, offset="auto" if declared else np.array([7, 7]), rank=0)
            f.write_chunk("particles/typeid", tid, offset="auto" if declared else np.array([7, 7]), rank=0)
            f.write_chunk("log/virial", log)            # default arguments: every rank writes the same rows
            f.end_frame()
            per_frame.append(len(mirror.calls) - n0)
        assert per_frame == ([0, 0, 0] if declared else [7, 7, 7]), per_frame
        if declared:
            f.set_partition(None)
        f.close()
        paths.append(p)
    with open(paths[0], "rb") as a, open(paths[1], "rb") as b:
        assert a.read() == b.read()
    g = fl.open(paths[0], "r")
    assert g.nframes == 3 and g.read_chunk(2, "particles/position").shape == (14, 3)
    g.close()


def test_declared_partition_refuses_another_share_at_once_and_reports_it_at_the_next_sync(mirror, tmp_path):
    """A chunk that does not bring this rank's declared share is refused by the call itself; the other ranks (who
    cannot be told without an exchange) learn of it at the next synchronisation point, where every rank returns it --
    the chunk keeps its place in the layout meanwhile, as on the peers."""
    f = fl.open(str(tmp_path / "r.gsd"), "w", application="app", schema="hoomd", schema_version=[1, 4])
    f.set_partition([7, 7])
    small_frame(f, 0)
    with pytest.raises(RuntimeError, match="Invalid pgsd argument"):
        f.write_chunk("particles/position", np.zeros((5, 3), np.float32), offset="auto")
    size_after = f.file_size
    with pytest.raises(RuntimeError, match="Invalid pgsd argument"):
        f.end_frame()                                   # this rank is reminded when it seals the frame ...
    with pytest.raises(RuntimeError, match="Invalid pgsd argument"):
        f.flush()                                       # ... and the synchronisation point tells everybody (once)
    f.set_partition(None)
    f.close()
    g = fl.open(str(tmp_path / "r.gsd"), "r")
    assert g.nframes == 1 and g.read_chunk(0, "particles/position").shape == (14, 3)      # the place is there
    assert size_after >= 5376 + 14 * 12
    g.close()


def _declared_bad_rank(rank, shm, path, q):
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
        from pgsd import _lib as L
        import pgsd.fl as F
        __import__("pgsd.dist").dist.init_shm(shm, rank, 2)
        f = F.open(path, "w", application="app", schema="hoomd", schema_version=[1, 4])
        f.set_partition([5, 5])
        h = f._h()
        data = np.arange(15, dtype=np.float32)
        out = []
        auto = 2 ** 64 - 1
        # frame 0: healthy on both ranks, no collective
        c0 = f.collective_count
        out.append(L.lib.pgsd_write_chunk(h, b"particles/position", 9, 5, 3, auto, 3, 0, 0, True, 0, data.ctypes.data))
        out.append(L.lib.pgsd_end_frame(h))
        out.append(f.collective_count - c0)
        # frame 1: rank 1 brings a NULL pointer for its rows -- refused there at once; rank 0 cannot know yet
        ptr = None if rank == 1 else data.ctypes.data
        out.append(L.lib.pgsd_write_chunk(h, b"particles/position", 9, 5, 3, auto, 3, 0, 0, True, 0, ptr))
        out.append(L.lib.pgsd_end_frame(h))
        out.append(L.lib.pgsd_flush(h))             # the synchronisation point: everybody hears of it
        # frame 2: healthy again -- the refused chunk kept its place, the ranks are still in step
        out.append(L.lib.pgsd_write_chunk(h, b"particles/position", 9, 5, 3, auto, 3, 0, 0, True, 0, data.ctypes.data))
        out.append(L.lib.pgsd_end_frame(h))
        out.append(L.lib.pgsd_flush(h))
        size = f.file_size
        f.set_partition(None)
        f.close()
        L.lib.pgsd_comm_finalize()
        q.put((rank, out, size))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), -1))
        raise


def test_declared_partition_a_refusal_on_one_rank_reaches_all_at_the_next_sync_and_the_file_goes_on(tmp_gsd):
    import multiprocessing as mp
    import uuid
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = "pgsddecl_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_declared_bad_rank, args=(r, shm, tmp_gsd, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    (r0, out0, size0), (r1, out1, size1) = results
    assert isinstance(out0, list) and isinstance(out1, list), (out0, out1)
    assert out0[:3] == [0, 0, 0] and out1[:3] == [0, 0, 0]           # frame 0: fine, and not one collective
    assert out1[3] == -2 and out0[3] == 0                             # the refusal: on the rank it concerns, at once
    assert out0[5] == -2 and out1[5] == -2                            # ... on every rank at the synchronisation point
    assert out0[6:] == [0, 0, 0] and out1[6:] == [0, 0, 0]            # and the file goes on
    assert size0 == size1 == 5376 + 3 * 10 * 12                       # the layouts stayed in step (three 10-row chunks)
    g = fl.open(tmp_gsd, "r")
    assert g.nframes == 3
    got = g.read_chunk(2, "particles/position")
    np.testing.assert_array_equal(got, np.tile(np.arange(15, dtype=np.float32).reshape(5, 3), (2, 1)))
    rank0_rows = g.read_chunk(1, "particles/position")[:5]            # frame 1: rank 0's rows are there, rank 1's a hole
    np.testing.assert_array_equal(rank0_rows, np.arange(15, dtype=np.float32).reshape(5, 3))
    g.close()


def test_index_relocation_costs_one_small_exchange_and_no_barrier(mirror, tmp_gsd):
    """pgsd_expand_file_index (pgsd.c:965-1091) puts the new block at the file's true end.  The ranks know where that
    will be from their own placements (the furthest byte each has written or handed to its pipeline): ONE 8-byte
    allgather, no barrier, no wait for bytes on their way -- a frame sealed asynchronously stays asynchronous when the
    index moves.  (PGSD_CHECK_EOF=1 keeps the old barrier + fstat beside it and compares: the whole suite passes so.)"""
    import os
    f = fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4])
    f.frame_exchange = True
    per_frame, eight_byte = [], []
    for i in range(40):                         # 4 entries per frame: the 128-entry block is full after frame 31
        c0, n0 = f.collective_count, len(mirror.calls)
        small_frame(f, i)
        f.write_chunk("particles/position", np.full((5, 3), i, np.float32), offset=np.array([5, 5]), rank=0)
        f.end_frame()
        per_frame.append(f.collective_count - c0)
        eight_byte.append(sum(1 for n in mirror.calls[n0:] if n == 8))
    extra = 2 if os.environ.get("PGSD_CHECK_EOF") else 1
    assert sorted(set(per_frame)) == [1, 1 + extra] and per_frame.count(1 + extra) == 1, per_frame
    k = per_frame.index(1 + extra)
    assert k == 32 and eight_byte[k] == 1 and sum(eight_byte) == 1, (k, eight_byte)
    f.close()
    with fl.open(tmp_gsd, "r") as g:
        assert g.nframes == 40
        assert float(g.read_chunk(39, "particles/position")[0, 0]) == 39.0
