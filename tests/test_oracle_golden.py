"""Pin the CPU oracle: byte-identical to files the compiled reference wrote (tests/golden/files)."""
import hashlib
import os

import pytest

import scenario as S


@pytest.mark.parametrize("name,P", S.golden_cases())
def test_oracle_matches_reference_file(name, P, tmp_gsd):
    log = S.run_oracle(S.scenario_path(name), tmp_gsd, P)
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        mine, ref = f.read(), g.read()
    assert len(mine) == len(ref)
    assert mine == ref
    # state trace (file_size, index/namelist location after each dump, find results)
    assert log == S.read_log(golden[:-4] + ".log")


def test_golden_checksums():
    """The committed fixtures are the ones make_golden.sh produced."""
    with open(os.path.join(S.GOLDEN, "SHA256SUMS")) as f:
        for line in f:
            digest, fn = line.split()
            with open(os.path.join(S.GOLDEN, fn), "rb") as g:
                assert hashlib.sha256(g.read()).hexdigest() == digest, fn
