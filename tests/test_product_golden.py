"""Host file layer of the product vs the reference: the SAME driver source that produced the
goldens when linked against the reference's pgsd.c is linked against libpgsd_amd.so and run
with one process per rank; the files must be byte-identical."""
import os

import pytest

import product
import scenario as S


@pytest.mark.parametrize("name,P", S.golden_cases())
def test_product_matches_reference_file(name, P, tmp_gsd):
    log = product.run_driver(S.scenario_path(name), tmp_gsd, P)
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        mine, ref = f.read(), g.read()
    assert len(mine) == len(ref)
    assert mine == ref
    assert log == S.read_log(golden[:-4] + ".log")
