"""Host file layer under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not available on the
pool): the scenario driver built with -fsanitize=address,undefined replays golden scenarios with one
process per rank; no report may appear and the files must still match the goldens."""
import os
import subprocess
import uuid

import pytest

import product
import scenario as S

ASAN_DRIVER = os.path.join(product.CSRC, "build", "scenario_driver_asan")


@pytest.fixture(scope="module")
def asan_driver():
    r = subprocess.run(["make", "-C", product.CSRC, "asan"], capture_output=True)
    if r.returncode != 0 or not os.path.exists(ASAN_DRIVER):
        pytest.skip("sanitizer build not available: " + r.stderr.decode()[-300:])
    return ASAN_DRIVER


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("name,P", [("sph_full", 4), ("index_expand", 2), ("names_reloc", 5), ("maxbuf", 4),
                                    ("reopen", 2), ("midflush", 3), ("zero_rank", 8), ("alltypes", 1),
                                    ("defaultargs", 3), ("vone_append", 2), ("idxbuf", 3)])
def test_scenarios_are_sanitizer_clean(asan_driver, name, P, batched, tmp_path):
    out = str(tmp_path / "out.gsd")
    script = S.scenario_path(name)
    if batched:     # the queue / frame exchange of pgsd_set_frame_exchange under the sanitizers as well
        script = product.batched_script(script, str(tmp_path / "batched.scn"))
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
