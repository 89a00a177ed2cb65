"""The native harness (pgsd-sph_amd/examples/benchmark_write.hip: C ABI only, no Python in the data path) --
the counterpart of the reference's pgsd/scripts/benchmark-write.cc -- in the ways a C++ caller would run it."""
import json
import os
import subprocess
import uuid

import pytest

import product

pytestmark = pytest.mark.gpu

EXE = os.path.join(product.CSRC, "build", "benchmark_write")


def run(args, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([EXE] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-1500:]
    return p.stdout


def test_single_rank_batched_and_per_chunk(tmp_path):
    product.build()
    for mode in ("batched", "perchunk"):
        out = run([200000, 4, str(tmp_path / (mode + ".gsd")), mode], {"PGSD_RANK": "0", "PGSD_NRANKS": "1"})
        d = json.loads(out.strip().splitlines()[-1])
        assert d["ranks"] == 1 and d["frames"] == 4 and d["pack_launches"] == 5
        assert d["written_bytes_rank0"] == 5 * 200000 * 28 and d["MBps"] > 0


def test_single_rank_over_the_librarys_rccl_communicator(tmp_path):
    """the C++ bootstrap: unique id -> pgsd_comm_create_rccl + pgsd_comm_set_default, every later exchange is an ncclAllGather"""
    product.build()
    out = run([100000, 3, str(tmp_path / "rccl.gsd"), "batched", "rccl"], {"PGSD_RANK": "0", "PGSD_NRANKS": "1"})
    d = json.loads(out.strip().splitlines()[-1])
    assert d["comm"] == "rccl" and d["written_bytes_rank0"] == 4 * 100000 * 28


def test_two_ranks_share_the_gpu_one_exchange_per_frame(tmp_path):
    product.build()
    shm = "pgsdnative_%s" % uuid.uuid4().hex[:10]
    path = str(tmp_path / "two.gsd")
    procs = [subprocess.Popen([EXE, "150000", "4", path, "batched"],
                              env=dict(os.environ, PGSD_RANK=str(r), PGSD_NRANKS="2", PGSD_SHM_NAME=shm),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-1500:]
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["ranks"] == 2 and d["exchange"] == "one per frame" and d["comm"] == "shm"
    # the handle's own collectives: create/open (2) + 5 frames x ONE exchange (the harness's row-count
    # allgather for its report and its timing barriers go through the default communicator, not the handle;
    # the barrier pgsd_close makes up comes after the count was read)
    assert d["collectives_rank0"] == 2 + 5, d


READ_EXE = os.path.join(product.CSRC, "build", "benchmark_read")


def _run_ranks(exe, args, P, timeout=300):
    shm = "pgsdnative_%s" % uuid.uuid4().hex[:10]
    procs = [subprocess.Popen([exe] + [str(a) for a in args],
                              env=dict(os.environ, PGSD_RANK=str(r), PGSD_NRANKS=str(P), PGSD_SHM_NAME=shm),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(P)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o[-500:], e[-1500:])
    return json.loads(outs[0][0].strip().splitlines()[-1])


@pytest.mark.parametrize("writers,readers", [(1, 1), (2, 3), (3, 2)])
def test_read_harness_restores_what_the_write_harness_wrote(tmp_path, writers, readers):
    """benchmark_read.hip (the counterpart of scripts/benchmark-read.cc): every rank reads an even share of the rows
    of every frame straight into Scalar4 arrays in HBM and a kernel compares them with the closed-form values the
    write harness produced -- whatever partition wrote the file."""
    product.build()
    path = str(tmp_path / "native.gsd")
    per_rank, frames = 100003, 3
    d = _run_ranks(EXE, [per_rank, frames, path, "batched", "shm", "keep"], writers)
    assert d["ranks"] == writers and os.path.exists(path)
    r = _run_ranks(READ_EXE, [path, "verify"], readers)
    n_global = per_rank * writers
    assert r["ranks"] == readers and r["frames"] == frames + 1 and r["particles"] == n_global
    assert r["verified"] is True and r["mismatches"] == 0
    assert r["rows_read"] == (frames + 1) * n_global and r["bytes_read"] == (frames + 1) * n_global * 28


@pytest.mark.parametrize("writers", [1, 3])
def test_declared_partition_harness_writes_without_a_collective_per_frame(tmp_path, writers):
    """`benchmark_write ... declared`: the row counts are exchanged once before the first frame and declared with
    pgsd_set_partition; the frames then cost the handle NO collective (create/open's two are all it ever issues before
    close), and the file is the one the read harness verifies value by value."""
    product.build()
    path = str(tmp_path / "declared.gsd")
    per_rank, frames = 70001, 4
    d = _run_ranks(EXE, [per_rank, frames, path, "declared", "shm", "keep"], writers)
    assert d["ranks"] == writers and d["exchange"] == "none (declared partition)"
    assert d["collectives_rank0"] == (2 if writers > 1 else 0), d
    r = _run_ranks(READ_EXE, [path, "verify"], 2)
    assert r["verified"] is True and r["mismatches"] == 0 and r["particles"] == per_rank * writers
    assert r["frames"] == frames + 1


@pytest.mark.parametrize("writers", [1, 2])
def test_elide_harness_writes_only_what_differs_from_frame_0(tmp_path, writers):
    """`benchmark_write ... elide`: pgsd_stage_chunks_device + pgsd_copy_staged_chunks / pgsd_compare_staged_chunks +
    pgsd_write_staged_chunks from C++ alone (no Python in the data path): position, velocity and type id never
    change and are written in frame 0 only, the charge changes with the step and is written every frame."""
    import numpy as np
    import pgsd.fl as fl
    product.build()
    path = str(tmp_path / "elide.gsd")
    per_rank, frames = 60007, 4
    d = _run_ranks(EXE, [per_rank, frames, path, "elide", "shm", "keep"], writers)
    assert d["ranks"] == writers and d["chunks_written"] == 4 + frames and d["chunks_elided"] == 3 * frames, d
    assert d["pack_launches"] == frames + 1
    n_global = per_rank * writers
    with fl.open(path, "r") as f:
        assert f.nframes == frames + 1
        for k in range(frames + 1):
            assert f.chunk_exists(k, "particles/charge")
            assert f.chunk_exists(k, "particles/position") == (k == 0) == f.chunk_exists(k, "particles/typeid")
            want = (np.arange(n_global, dtype=np.uint64) % 13).astype(np.float32) + np.float32(100.0 * k)
            assert f.read_chunk(k, "particles/charge").tobytes() == want.tobytes()
    r = _run_ranks(READ_EXE, [path, "verify"], 3)         # frame 0 holds the arrays the read harness verifies
    assert r["verified"] is True and r["mismatches"] == 0 and r["rows_read"] == n_global


DUMP_EXE = os.path.join(product.CSRC, "build", "dump_writer")


@pytest.mark.parametrize("ranks,group,order", [(1, "all", "hilbert"), (1, "fluid", "hilbert"), (2, "fluid", "random"),
                                               (3, "all", "hilbert"), (1, "all", "random")])
def test_dump_writer_under_a_running_simulation(tmp_path, ranks, group, order):
    """examples/dump_writer.hip: a simulation stepping Scalar4 arrays on its own stream, snapshots gathered into tag
    order through the reverse-tag array (memory order: a Hilbert curve through the lattice of tags, or an adversarial
    uniform-like permutation; optionally a group only: pgsd_select_rows), static arrays elided against
    frame 0 in HBM, ONE collective per frame, asynchronous seals.  The program re-reads its file through the
    reference's entry points and compares every frame with a host model bit for bit (its exit code and `ok`); here
    the file is read once more through pgsd.fl and the group's size and the elision are checked from outside."""
    import numpy as np
    import pgsd.fl as fl
    product.build()
    path = str(tmp_path / "dump.gsd")
    per_rank, steps, period = 50021, 40, 10
    d = _run_ranks(DUMP_EXE, [per_rank, steps, period, path, group, "keep", order], ranks)
    assert d["memory_order"].startswith("Hilbert" if order == "hilbert" else "multiplicative")
    frames = steps // period + 1
    assert d["ok"] is True and d["failed_check"] == 0 and d["ranks"] == ranks and d["frames"] == frames == d["verified_frames"]
    assert d["pack_launches"] == frames                                  # one fused gather + pack launch per snapshot
    assert d["chunks_written"] == 5 + 3 * (frames - 1) and d["chunks_elided"] == 2 * (frames - 1), d
    # the handle's collectives: create/open's two + ONE allgather per frame (several ranks only; close's come later)
    assert d["collectives_rank0"] == ((2 + frames) if ranks > 1 else 0), d
    g = np.arange(per_rank * ranks, dtype=np.uint64)
    types = ((g * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(3)
    kept = g if group == "all" else g[types != 2]
    with fl.open(path, "r") as f:
        assert f.nframes == frames
        for k in range(frames):
            assert int(f.read_chunk(k, "configuration/step")[0]) == k * period
            assert int(f.read_chunk(k, "particles/N")[0]) == kept.size
            assert f.chunk_exists(k, "particles/typeid") == (k == 0) == f.chunk_exists(k, "particles/mass")
            dens = f.read_chunk(k, "particles/density")
            want = np.float32(1000.0) + np.float32(0.5 * k * period) + (kept % np.uint64(7)).astype(np.float32)
            assert dens.tobytes() == want.astype(np.float32).tobytes()
            assert f.read_chunk(k, "particles/position").shape == (kept.size, 3)
        assert f.read_chunk(0, "particles/typeid").tobytes() == types[np.isin(g, kept)].astype(np.uint32).tobytes()
