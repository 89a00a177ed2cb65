"""Host file layer under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not available on the
pool): the scenario driver built with -fsanitize=address,undefined replays golden scenarios with one
process per rank; no report may appear and the files must still match the goldens."""
import os
import subprocess
import uuid

import pytest

import product
import scenario as S

ASAN_DRIVER = os.path.join(product.TBUILD, "scenario_driver_asan")


def _make(target):
    return product.locked_make(["-C", product.TESTS, target], check=False, capture_output=True)


@pytest.fixture(scope="module")
def asan_driver():
    r = _make("asan")
    if r.returncode != 0 or not os.path.exists(ASAN_DRIVER):
        pytest.skip("sanitizer build not available: " + r.stderr.decode()[-300:])
    return ASAN_DRIVER


@pytest.mark.parametrize("batched", [0, 1, 2, 3])
@pytest.mark.parametrize("name,P", [("sph_full", 4), ("index_expand", 2), ("names_reloc", 5), ("maxbuf", 4),
                                    ("reopen", 2), ("midflush", 3), ("zero_rank", 8), ("alltypes", 1),
                                    ("defaultargs", 3), ("vone_append", 2), ("idxbuf", 3)])
def test_scenarios_are_sanitizer_clean(asan_driver, name, P, batched, tmp_path):
    out = str(tmp_path / "out.gsd")
    script = S.scenario_path(name)
    if batched:     # the queue / frame exchange of pgsd_set_frame_exchange under the sanitizers as well; 2: with
        # pgsd_set_deferred_rows -- the driver frees a chunk's rows at the next end_frame / flush / close / dump,
        # so a row read after the exchange that should have consumed it is a use-after-free ASan reports;
        # 3: declared partitions (pgsd_set_partition): no exchange where the declaration covers a chunk
        script = product.batched_script(script, str(tmp_path / "batched.scn"), batched)
    shm = "pgsdasan_%s" % uuid.uuid4().hex[:10]
    procs = []
    for r in range(P):
        env = dict(os.environ, PGSD_RANK=str(r), PGSD_NRANKS=str(P), PGSD_SHM_NAME=shm,
                   ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        procs.append(subprocess.Popen([asan_driver, script, out], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
    reports = []
    for p in procs:
        _, err = p.communicate(timeout=180)
        reports.append((p.returncode, err.decode()))
    for rc, err in reports:
        assert rc == 0 and "ERROR" not in err and "runtime error" not in err, err[-2000:]
    with open(out, "rb") as a, open(os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P)), "rb") as b:
        assert a.read() == b.read()


def test_local_reads_are_sanitizer_clean(asan_driver, tmp_path):
    """pgsd_set_local_reads(1) under ASan + UBSan: the read-heavy golden on one rank, file == the reference's."""
    out = str(tmp_path / "out.gsd")
    script = product.local_reads_script(S.scenario_path("readback"), str(tmp_path / "local.scn"))
    env = dict(os.environ, PGSD_RANK="0", PGSD_NRANKS="1",
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([asan_driver, script, out], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=180)
    err = p.stderr.decode()
    assert p.returncode == 0 and "ERROR" not in err and "runtime error" not in err, err[-2000:]
    with open(out, "rb") as a, open(os.path.join(S.GOLDEN, "readback.p1.gsd"), "rb") as b:
        assert a.read() == b.read()


TSAN_DRIVER = os.path.join(product.TBUILD, "scenario_driver_tsan")

BIG_CHUNKS = """create app hoomd 1 4 rw 0
seed 5
chunk configuration/step u64 1 0 same:1
chunk particles/position f32 3 1 even:7000000
chunk particles/velocity f32 3 1 even:7000000
end_frame
chunk configuration/step u64 1 0 same:1
chunk particles/position f32 3 1 even:7000000
end_frame
dump
read 0 particles/position 7000000 3 0 0 7000000 3 4
read 1 particles/position 3000000 3 1000 1 3000000 3 4
close
"""


def test_writer_pool_and_read_threads_are_race_free(tmp_path):
    """ThreadSanitizer over the threads the host layer starts inside one process: 84 MB chunks are split over the
    writer pool on the way out and over the host read threads on the way in.  The file must equal the oracle's."""
    r = _make("tsan")
    if r.returncode != 0 or not os.path.exists(TSAN_DRIVER):
        pytest.skip("sanitizer build not available: " + r.stderr.decode()[-300:])
    scn = tmp_path / "big.scn"
    scn.write_text(BIG_CHUNKS)
    out, ref = str(tmp_path / "out.gsd"), str(tmp_path / "ref.gsd")
    p = subprocess.run([TSAN_DRIVER, str(scn), out], env=dict(os.environ, PGSD_RANK="0", PGSD_NRANKS="1"),
                       capture_output=True, text=True, timeout=600)
    if "FATAL: ThreadSanitizer" in p.stderr and "unexpected memory mapping" in p.stderr:
        pytest.skip("ThreadSanitizer cannot run in this container")
    assert p.returncode == 0 and "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    log = S.run_oracle(str(scn), ref, 1)
    assert [ln for ln in p.stdout.splitlines() if ln.strip()] == log
    with open(out, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()
