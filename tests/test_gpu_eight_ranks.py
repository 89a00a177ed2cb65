"""BASELINE config 3's shape at P = 8 on the one-GPU box: eight ranks pack their partitions on cuda:0 and write ONE
shared file whose offsets come from one allgather per frame.

The GPU boxes let at most six processes use the card together, so the eight ranks are eight THREADS of one child
process, each with a communicator and a handle of its own (`pgsd_comm_create_shm` / `pgsd_comm_create_rccl`,
`pgsd_create_and_open_on`): over the shm back end, and over the RCCL back end's own code with
tests/drivers/fake_rccl.cpp standing in for librccl (real RCCL refuses two ranks on one device; eight real ranks
need the driver's 8-GPU node).  The file must equal the reference-written golden `posvelid.p8.gsd` for the golden's
values, and the oracle's 8-rank file for an uneven partition with an empty and a one-row rank."""
import os
import subprocess
import sys

import numpy as np
import pytest

import product
import scenario as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "thread_ranks_worker.py")
FAKE = os.path.join(product.TBUILD, "libpgsd_fake_rccl.so")


def run_worker(kind, P, path, what):
    env = dict(os.environ)
    env.pop("PGSD_RCCL_LIBRARY", None)
    if kind == "rccl":
        product.build()
        assert os.path.exists(FAKE)
        env["PGSD_RCCL_LIBRARY"] = FAKE
        env["PGSD_FAKE_RCCL_SYNC"] = "1"          # ranks as threads of one process: see fake_rccl.cpp
    p = subprocess.run([sys.executable, WORKER, kind, str(P), path, what], env=env, capture_output=True, text=True,
                       timeout=400)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-3000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1]
    per_frame = [float(v) for v in line.split("=")[1].split(",")]
    assert per_frame == [1.0] * P, per_frame               # ONE collective per frame on every rank


@pytest.mark.parametrize("kind", ["shm", "rccl"])
def test_eight_ranks_reproduce_the_reference_written_golden(kind, tmp_path):
    mine = str(tmp_path / "p8.gsd")
    run_worker(kind, 8, mine, "posvelid")
    with open(mine, "rb") as a, open(os.path.join(S.GOLDEN, "posvelid.p8.gsd"), "rb") as b:
        assert a.read() == b.read()


@pytest.mark.parametrize("kind", ["shm", "rccl"])
def test_eight_ranks_uneven_partition_matches_the_oracle(kind, tmp_path):
    from test_gpu_file import _oracle_frames
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from thread_ranks_worker import UNEVEN
    P = 8
    mine, ref = str(tmp_path / "u8.gsd"), str(tmp_path / "ref.gsd")
    run_worker(kind, P, mine, "uneven")
    counts = UNEVEN[:P]
    row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
    frames = []
    for seed in (40, 41, 42):
        pos = [S.gen_data(9, seed, int(row0[r]), counts[r], 3) for r in range(P)]
        tid = [S.gen_data(3, seed, int(row0[r]), counts[r], 1) for r in range(P)]
        frames.append([("configuration/step", 4, 1, False, [S.gen_data(4, seed, 0, 1, 1)] * P),
                       ("particles/position", 9, 3, True, pos), ("particles/velocity", 9, 3, True, pos),
                       ("particles/typeid", 3, 1, True, tid)])
    _oracle_frames(ref, P, frames)
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()


FULL = os.path.join(ROOT, "tests", "fullsize_ranks_worker.py")


@pytest.mark.parametrize("kind", ["shm", "rccl"])
@pytest.mark.parametrize("what", ["config3", "config4"])
def test_eight_ranks_at_ten_million_particles_each_match_the_oracle(what, kind):
    """BASELINE configs 3 and 4 at their own size AND rank count: eight ranks (threads) x 10 M particles, eight
    pipelines and eight pinned rings onto one inode.  config3: position + velocity + typeid, two frames = 4.48 GB, the
    second frame beyond 4 GiB of file offset; config4: the 112 B/particle SPH set, one frame = 8.96 GB.  sha256 of
    the file == sha256 of the oracle's 8-rank file of host copies of the same values (every rank's rows at
    file_size + offset * sz, pgsd.c:2225-2249; the partition of benchmark-write.cc:33-45)."""
    import shutil
    P, n = 8, 10_000_000
    d = "/dev/shm"
    need = 20 << 30                                  # the product's file and the oracle's side by side: 2 x 8.96 GB
    if not os.path.isdir(d) or shutil.disk_usage(d).free < need:
        pytest.skip("/dev/shm has less than 20 GB free")
    env = dict(os.environ)
    env.pop("PGSD_RCCL_LIBRARY", None)
    if kind == "rccl":
        product.build()
        assert os.path.exists(FAKE)
        env["PGSD_RCCL_LIBRARY"] = FAKE
        env["PGSD_FAKE_RCCL_SYNC"] = "1"          # ranks as threads of one process: see fake_rccl.cpp
    p = subprocess.run([sys.executable, FULL, kind, str(P), str(n), what, d], env=env, capture_output=True, text=True,
                       timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
    assert lines, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
    r = dict(kv.split("=") for kv in lines[-1].split()[1:])
    payload = P * n * (28 * 2 if what == "config3" else 112)
    assert int(r["size_mine"]) == int(r["size_ref"]) > payload
    if what == "config3":
        assert int(r["size_mine"]) > (1 << 32) > int(r["size_mine"]) // 2        # frame 2 straddles the 4 GiB line
    assert r["sha_mine"] == r["sha_ref"]
    assert [float(v) for v in r["collectives"].split(",")] == [1.0] * P          # ONE collective per frame, every rank
    assert p.returncode == 0, p.stderr[-3000:]
