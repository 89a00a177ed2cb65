"""A failing write (file-size limit -> EFBIG) surfaces as OSError on EVERY rank at the same call,
nothing dead-locks and the file can still be closed. The reference has no such agreement: a rank
whose MPI_File_write_at fails returns PGSD_ERROR_IO alone (pgsd.c:2229-2236) while the others go
on to the next collective."""
import errno
import json
import os
import subprocess
import sys
import uuid

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "io_error_worker.py")


def run_ranks(path, nranks, limited_rank, device, batched=False, rows=None):
    shm = "pgsdioerr_%s" % uuid.uuid4().hex[:10]
    env = dict(os.environ) if rows is None else dict(os.environ, PGSD_IOERR_N=str(rows))
    procs = [subprocess.Popen([sys.executable, WORKER, path, str(r), str(nranks), shm,
                               "1" if r == limited_rank else "0", "1" if device else "0", "1" if batched else "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in range(nranks)]
    reports = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=240)
            assert p.returncode == 0, err.decode()[-2000:]
            reports.append(json.loads(out.decode().strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink("/dev/shm/" + shm)
        except OSError:
            pass
    return reports


def check(reports, limited_rank, device=False):
    for rep in reports:
        ev = {e[0]: e[1:] for e in rep["events"]}
        assert ev["before:write"] == ["ok"] and ev["before:end_frame"] == ["ok"], rep
        if rep["rank"] == limited_rank and not device:
            assert ev["after:write"] == ["OSError", errno.EFBIG], rep      # the failing rank knows at once
        else:
            assert ev["after:write"] == ["ok"], rep                        # device chunks are only enqueued
        assert ev["after:end_frame"][0] == "OSError", rep                  # every rank, not only the one that failed
        assert ev["after:end_frame"][1] == errno.EFBIG, rep
        assert "close" in ev, rep                                          # no dead-lock on the way out


def test_write_failure_single_rank(tmp_gsd):
    check(run_ranks(tmp_gsd, 1, 0, device=False), 0)


@pytest.mark.parametrize("limited_rank", [0, 1])
def test_write_failure_on_one_rank_is_agreed_by_all(limited_rank, tmp_gsd):
    check(run_ranks(tmp_gsd, 2, limited_rank, device=False), limited_rank)


@pytest.mark.gpu
def test_device_pipeline_write_failure(tmp_gsd):
    """the failing pwrite happens on the pipeline's writer thread; end_frame reports it"""
    check(run_ranks(tmp_gsd, 1, 0, device=True), 0, device=True)


@pytest.mark.gpu
def test_device_pipeline_write_failure_two_ranks_sharing_the_gpu(tmp_gsd):
    check(run_ranks(tmp_gsd, 2, 1, device=True), 1, device=True)


def check_batched(reports, limited_rank, device=False):
    """Batched frame exchange (pgsd_set_frame_exchange): ONE collective per frame, so the failing rank knows
    at once, the others with the next exchange -- the next frame's, or the one pgsd_close makes.  Nobody
    dead-locks, nobody misses it."""
    for rep in reports:
        ev = {e[0]: e[1:] for e in rep["events"]}
        assert ev["before:write"] == ["ok"] and ev["before:end_frame"] == ["ok"], rep
        errors = [k for k, v in ev.items() if v[0] == "OSError"]
        assert errors and all(ev[k][1] == errno.EFBIG for k in errors), rep
        if rep["rank"] == limited_rank:
            first = "after:end_frame" if device else "after:write"
            assert ev[first][0] == "OSError", rep
        assert not [k for k in errors if k.startswith("before")], rep
        assert "close" in ev, rep


@pytest.mark.parametrize("limited_rank", [0, 1])
def test_write_failure_with_batched_frame_exchange_reaches_all_ranks(limited_rank, tmp_gsd):
    check_batched(run_ranks(tmp_gsd, 2, limited_rank, device=False, batched=True), limited_rank)


@pytest.mark.gpu
def test_device_write_failure_with_batched_frame_exchange_two_ranks(tmp_gsd):
    check_batched(run_ranks(tmp_gsd, 2, 1, device=True, batched=True), 1, device=True)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,batched", [(1, False), (2, False), (2, True)])
def test_direct_path_write_failure(nranks, batched, tmp_gsd):
    """Frames small enough for the direct path (1.2 MB: packed straight into pinned host memory, pwrite()n by the
    thread that seals the frame): the failing pwrite is reported by end_frame on every rank, like the pipeline's."""
    reports = run_ranks(tmp_gsd, nranks, nranks - 1, device=True, batched=batched, rows=100_000)
    if batched:
        check_batched(reports, nranks - 1, device=True)
    else:
        check(reports, nranks - 1, device=True)
