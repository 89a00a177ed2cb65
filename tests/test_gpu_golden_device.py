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


def _replay(name, P, tmp_gsd, tmp_path, mode, batch, async_seal=False):
    scn = product.device_script(S.scenario_path(name), str(tmp_path / "device.scn"), mode, batch, async_seal)
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    log = product.run_driver(scn, tmp_gsd, P, allow_fail=_fails_on_purpose(golden), driver=product.DEVICE_DRIVER,
                             threads=P > 5)
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        mine, ref = f.read(), g.read()
    assert len(mine) == len(ref)
    assert mine == ref
    # the inserted `device` / `batch` lines shift the line numbers the driver prints
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines
                           if not ln.startswith("rc ") or not any(c in ln for c in ("cmd=batch", "cmd=device", "cmd=async"))]
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


@pytest.mark.parametrize("name,P", [c for c in S.golden_cases() if c[1] in (1, 2, 4, 8)])
def test_product_golden_device_asynchronous_seals(name, P, tmp_gsd, tmp_path):
    """... and with every frame sealed asynchronously (pgsd_end_frame_async: the metadata of the frame committed at
    once, copies and pwrites running on behind the caller -- through index relocations too, since round 5): the
    layout does not change by a byte, reads and re-opens in the scenarios see complete frames."""
    _replay(name, P, tmp_gsd, tmp_path, 1, 1, async_seal=True)


@pytest.mark.parametrize("name,P", [c for c in S.golden_cases() if c[0] in ("index_expand", "idxbuf", "benchlike") and c[1] <= 4])
def test_computed_end_of_file_equals_fstat_on_the_device_path(name, P, tmp_gsd, tmp_path, monkeypatch):
    """PGSD_CHECK_EOF=1 (tests/test_product_golden.py): every index relocation compares the end of file computed from
    the ranks' placements -- device chunks count from the moment they are handed to the pipeline -- with fstat's after a
    drain, and the index mirror with the block on disk; asynchronous seals, batched exchange."""
    monkeypatch.setenv("PGSD_CHECK_EOF", "1")
    _replay(name, P, tmp_gsd, tmp_path, 1, 1, async_seal=True)


FUZZ_SEEDS = int(os.environ.get("PGSD_FUZZ_SEEDS", "16"))


@pytest.mark.parametrize("seed,P", [(seed, P) for seed in range(300, 300 + FUZZ_SEEDS) for P in (1, 2, 3)])
def test_random_scenarios_from_hbm_equal_the_oracle(seed, P, tmp_path):
    """The random call sequences of tests/test_fuzz_parity.py (random names, types, widths, partitions with empty
    ranks, default-argument writes, wrong global sizes, buffer limits, mid-frame flushes, re-opens, read-backs) with
    the rows of every chunk write in HBM: file and state trace identical to the oracle's.  Odd seeds: rows inside
    wider device arrays and the frame's exchange batched."""
    from test_fuzz_parity import make_script
    scn = tmp_path / "fuzz.scn"
    scn.write_text(make_script(seed, P))
    o_path, p_path = str(tmp_path / "oracle.gsd"), str(tmp_path / "device.gsd")
    o_log = S.run_oracle(str(scn), o_path, P)
    assert not [ln for ln in o_log if ln.startswith("rc ")], o_log
    dscn = product.device_script(str(scn), str(tmp_path / "device.scn"), 2 if seed % 2 else 1, 1 if seed % 2 else 0)
    p_log = product.run_driver(dscn, p_path, P, driver=product.DEVICE_DRIVER)
    with open(o_path, "rb") as a, open(p_path, "rb") as b:
        assert a.read() == b.read()
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines]
    assert strip(p_log) == strip(o_log)


@pytest.mark.parametrize("seed,P", [(s, P) for s in range(500, 500 + max(4, FUZZ_SEEDS // 4)) for P in (1, 2, 3)])
def test_relocation_heavy_scenarios_from_hbm_with_asynchronous_seals(seed, P, tmp_path):
    """The relocation-heavy scripts of tests/test_fuzz_parity.py (one to three index relocations each) with every chunk's
    rows in HBM and EVERY frame sealed asynchronously: since round 5 a seal stays asynchronous when the index moves
    (the file's true end comes from the ranks' placements), so the new block is placed while earlier frames' copies
    and writes are still on their way -- file and state trace must be the oracle's."""
    from test_fuzz_parity import make_relocation_script
    scn = tmp_path / "reloc.scn"
    scn.write_text(make_relocation_script(seed, P))
    o_path, p_path = str(tmp_path / "oracle.gsd"), str(tmp_path / "device.gsd")
    o_log = S.run_oracle(str(scn), o_path, P)
    assert not [ln for ln in o_log if ln.startswith("rc ")], o_log
    dscn = product.device_script(str(scn), str(tmp_path / "device.scn"), 2 if seed % 2 else 1, 1, True)
    p_log = product.run_driver(dscn, p_path, P, driver=product.DEVICE_DRIVER)
    with open(o_path, "rb") as a, open(p_path, "rb") as b:
        assert a.read() == b.read()
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines
                           if not ln.startswith("rc ") or not any(c in ln for c in ("cmd=batch", "cmd=device", "cmd=async"))]
    assert strip(p_log) == strip(o_log)
