import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
sys.path.insert(0, ROOT)


def _build_if_missing():
    """A fresh checkout has no built artefacts (they are git-ignored): build the library and the
    oracle before collection imports them. The product itself still refuses to run without its
    library; this only saves the `python -c "import __graft_entry__ as g; g.build()"` step."""
    import subprocess
    import glob
    from product import locked_make
    lib = os.path.join(ROOT, "pgsd-sph_amd", "pgsd", "libpgsd_amd.so")
    drv = os.path.join(ROOT, "tests", "build", "scenario_driver")
    # always `make` (a no-op when up to date), under the session's lock: with pytest-xdist every worker passes here
    # before it runs anything, so a library or driver is never relinked while another worker executes it
    locked_make(["-C", os.path.join(ROOT, "pgsd-sph_amd", "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    locked_make(["-C", os.path.join(ROOT, "tests"), "-j8"], stdout=subprocess.DEVNULL)
    ext = glob.glob(os.path.join(ROOT, "pgsd-sph_amd", "pgsd", "_fl.*.so"))      # the Cython file layer (pgsd.fl)
    if not (os.path.exists(lib) and os.path.exists(drv) and ext):
        raise RuntimeError("make did not produce libpgsd_amd.so / scenario_driver / the pgsd.fl extension")
    if not os.path.exists(os.path.join(ROOT, "oracle", "libpgsd_oracle.so")):
        locked_make(["-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_driver")):
        locked_make(["-C", os.path.join(ROOT, "oracle"), "ref"], check=False, stdout=subprocess.DEVNULL,
                    stderr=subprocess.DEVNULL)          # only where the reference and MPICH exist


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference; build container only)")
    _build_if_missing()


@pytest.fixture
def tmp_gsd(tmp_path):
    return str(tmp_path / "out.gsd")


@pytest.fixture(autouse=True)
def _tuning_is_read_again():
    """The library reads its PGSD_* tuning variables ONCE (csrc/pgsd_private.h: pgsd_reload_tuning).  Tests that set
    them with monkeypatch (PGSD_PACK_KERNEL=tiles ...) need them read again -- before the test (its fixtures have set
    them by the time it launches something) and after it (the next test gets the restored environment)."""
    from pgsd import _lib
    _lib.lib.pgsd_reload_tuning()
    yield
    _lib.lib.pgsd_reload_tuning()
