"""Every reference-written golden, replayed FROM HBM.

tests/test_product_golden.py pins the HOST path to the 48 files the compiled reference wrote; here the same
scenario scripts run through the scenario driver's device build (tests/drivers/scenario_driver.c,
-DPGSD_DRIVER_DEVICE): the rows of every chunk write -- partitioned or replicated, any of the ten types, any width,
empty ranks included -- are uploaded to device memory and go through pgsd_write_chunk_device, the device twin of
pgsd_write_chunk (same arguments, pgsd.h:551-564): pack kernel, staging, device->host copies, pwrite at the offsets
MPI_File_write_at would use (pgsd.c:2225-2249).  File and state trace must equal the reference's byte for byte,
at 1-5 ranks as processes sharing the GPU and at 8 ranks as threads of one process (the boxes admit six GPU
processes).  Scenario set: tests/golden/make_golden.sh."""
import os
import re

import pytest

import product
import scenario as S
from test_product_golden import _fails_on_purpose

pytestmark = pytest.mark.gpu


def _replay(name, P, tmp_gsd, tmp_path, mode, batch):
    scn = product.device_script(S.scenario_path(name), str(tmp_path / "device.scn"), mode, batch)
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    log = product.run_driver(scn, tmp_gsd, P, allow_fail=_fails_on_purpose(golden), driver=product.DEVICE_DRIVER,
                             threads=P > 5)
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        mine, ref = f.read(), g.read()
    assert len(mine) == len(ref)
    assert mine == ref
    # the inserted `device` / `batch` lines shift the line numbers the driver prints
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines
                           if not ln.startswith("rc ") or ("cmd=batch" not in ln and "cmd=device" not in ln)]
    assert strip(log) == strip(S.read_log(golden[:-4] + ".log"))


@pytest.mark.parametrize("name,P", S.golden_cases())
def test_product_golden_device(name, P, tmp_gsd, tmp_path):
    """All 48: dense device arrays, one exchange per chunk (the call-for-call twin of the reference's sequence)."""
    _replay(name, P, tmp_gsd, tmp_path, 1, 0)


@pytest.mark.parametrize("name,P", [c for c in S.golden_cases() if c[1] in (1, 3, 8)])
def test_product_golden_device_strided_rows_batched_exchange(name, P, tmp_gsd, tmp_path):
    """... and with the rows inside wider device arrays (xyz of a Scalar4; a foreign column on either side for the
    other widths: the strided kernels) and the frame's exchange batched -- the way bench.py and pgsd.hoomd write."""
    _replay(name, P, tmp_gsd, tmp_path, 2, 1)
