"""The C-ABI library loads and exports every symbol include/pgsd.h declares."""
import ctypes
import os
import re

import product


def declared_symbols(path=None):
    hdr = open(path or os.path.join(product.ROOT, "include", "pgsd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(pgsd_[a-z0-9_]+)\s*\(", hdr))
    return sorted(names)


PRIVATE_HEADER = os.path.join(product.CSRC, "pgsd_private.h")


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(product.build())
    names = declared_symbols()
    # the sixteen entry points of the reference ABI (pgsd.h:362-735) must be among them
    for ref in ("pgsd_make_version", "pgsd_create_and_open", "pgsd_open", "pgsd_close", "pgsd_end_frame",
                "pgsd_flush", "pgsd_write_chunk", "pgsd_find_chunk", "pgsd_read_chunk", "pgsd_get_nframes",
                "pgsd_get_nnames", "pgsd_sizeof_type", "pgsd_find_matching_chunk_name",
                "pgsd_get_maximum_write_buffer_size", "pgsd_set_maximum_write_buffer_size",
                "pgsd_get_index_entries_to_buffer", "pgsd_set_index_entries_to_buffer",
                "pgsd_bcast_index_entry"):
        assert ref in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert 40 <= len(names) <= 58          # the reference has 18; tuning / test plumbing is not public API


def test_the_library_exports_nothing_that_no_header_declares():
    """Public API = include/pgsd.h; what tests, tools and the binding reach besides is declared in
    csrc/pgsd_private.h; nothing else is exported under the pgsd_ prefix."""
    import subprocess
    lib = product.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    exported = {ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T"
                and ln.split()[2].startswith("pgsd_")}
    public, private = set(declared_symbols()), set(declared_symbols(PRIVATE_HEADER))
    assert not (public & private)
    assert exported == public | private, sorted(exported ^ (public | private))


def test_abi_version_is_exported_and_checked_by_the_bindings():
    """entry points changed signature in place between rounds; bindings that resolve symbols at run time
    compare pgsd_abi_version() with the PGSD_ABI_VERSION they were written against."""
    from pgsd import _lib
    hdr = open(os.path.join(product.ROOT, "include", "pgsd.h")).read()
    declared = int(re.search(r"#define PGSD_ABI_VERSION (\d+)u", hdr).group(1))
    assert _lib.lib.pgsd_abi_version() == declared == _lib.ABI_VERSION
    src = open(os.path.join(product.ROOT, "pgsd-sph_amd", "pgsd", "_fl.pyx")).read()
    assert "C.pgsd_abi_version() != C.PGSD_ABI_VERSION" in src


def test_struct_layouts_match_the_format():
    from pgsd import _lib
    assert ctypes.sizeof(_lib.Header) == 256
    assert ctypes.sizeof(_lib.IndexEntry) == 32
    assert _lib.lib.pgsd_sizeof_type(9) == 4 and _lib.lib.pgsd_sizeof_type(10) == 8
    assert _lib.lib.pgsd_sizeof_type(0) == 0 and _lib.lib.pgsd_sizeof_type(11) == 0
    assert _lib.lib.pgsd_make_version(1, 4) == (1 << 16 | 4)


def test_device_entry_points_fail_loudly_without_a_gpu():
    """No CPU stand-in: on a box without a GPU the device calls return PGSD_ERROR_NO_DEVICE."""
    from pgsd import _lib
    if _lib.lib.pgsd_device_available():
        return
    job = (_lib.PackJob * 1)()
    assert _lib.lib.pgsd_pack_fields(1, job, 16, None, None) == _lib.ERROR_NO_DEVICE
    assert "no HIP device" in _lib.last_error()
    ws = ctypes.c_uint64(0)
    assert _lib.lib.pgsd_select_rows(None, 0, None, ctypes.byref(ws), None) == _lib.ERROR_NO_DEVICE
    assert _lib.lib.pgsd_device_alloc(-1, 64, None, 0) is None and "no HIP device" in _lib.last_error()


def test_device_write_without_gpu_raises(tmp_gsd):
    import numpy as np
    import pytest
    import pgsd.fl as fl
    from pgsd import _lib
    if _lib.lib.pgsd_device_available():
        return
    with fl.open(tmp_gsd, 'w', application='a', schema='s', schema_version=[1, 0]) as f:
        field = fl.DeviceField(ptr=0x1000, dtype=np.float32, N=8, M=3, stride=4)
        with pytest.raises(RuntimeError, match="no HIP device"):
            f.write_chunk('particles/position', field)


def test_rccl_library_override_is_honoured_and_fails_loudly():
    """PGSD_RCCL_LIBRARY names the RCCL build the communicator loads; a path that cannot be loaded is an error of
    the call (no silent fall-back to another librccl).  No GPU needed: the id call only opens the library."""
    import subprocess
    import sys
    code = ("import sys, ctypes; sys.path.insert(0, %r)\n"
            "from pgsd import _lib\n"
            "buf = (ctypes.c_uint8 * 128)()\n"
            "rc = _lib.lib.pgsd_comm_rccl_unique_id(buf)\n"
            "print(rc, _lib.last_error())\n" % os.path.join(product.ROOT, "pgsd-sph_amd"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PGSD_RCCL_LIBRARY="/nonexistent/librccl.so"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-1000:]
    rc, msg = p.stdout.strip().split(" ", 1)
    assert int(rc) != 0 and "PGSD_RCCL_LIBRARY" in msg and "/nonexistent/librccl.so" in msg


def test_device_buffer_fails_loudly_without_a_gpu():
    """pgsd.fl.DeviceBuffer is device memory owned by the library (pgsd_device_alloc): no host stand-in."""
    import numpy as np
    import pytest
    import pgsd.fl as fl
    from pgsd import _lib
    if _lib.lib.pgsd_device_available():
        return
    with pytest.raises(RuntimeError, match="no HIP device"):
        fl.DeviceBuffer((16,), np.float32)
    with pytest.raises(ValueError):
        fl.select_rows(np.ones(8, dtype=np.uint8))          # host memory is not a flags array on the GPU


def test_the_product_builds_without_the_tests_directory(tmp_path):
    """csrc/Makefile's `all` used to build the scenario drivers and the stand-in librccl out of
    tests/drivers.  A copy of the tree without tests/ must know how to make `all` (dry run: every prerequisite is either
    present or has a rule), and the Makefile must not name tests/ at all."""
    import shutil
    import subprocess
    mk = open(os.path.join(product.CSRC, "Makefile")).read()
    assert "$(ROOT)/tests" not in mk and "tests/drivers" not in mk
    root = tmp_path / "tree"
    shutil.copytree(os.path.join(product.ROOT, "include"), root / "include")
    shutil.copytree(os.path.join(product.ROOT, "pgsd-sph_amd"), root / "pgsd-sph_amd",
                    ignore=shutil.ignore_patterns("build", "*.so", "__pycache__"))
    p = subprocess.run(["make", "-n", "-C", str(root / "pgsd-sph_amd" / "csrc"), "all"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-800:]
    assert "pgsd_pack.hip" in p.stdout and "tests/" not in p.stdout
