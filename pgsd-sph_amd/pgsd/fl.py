"""PGSD file layer API, MI355X-native: ``pgsd.fl``.

The module is the compiled Cython file layer ``pgsd/_fl.pyx`` -- the counterpart of the reference's
``pgsd/pgsd/fl.pyx`` (same ``open`` / ``PGSDFile`` surface) bound to the C ABI of ``libpgsd_amd.so`` through
``libpgsd_amd.pxd``.  This shim only fixes the load order: ``pgsd._lib`` first (it imports torch, whose bundled
HIP runtime must be the one the process holds, and loads ``libpgsd_amd.so``), then the extension module, which
links against the same library.  There is no Python or ctypes stand-in for the file layer: without the built
extension the import fails with the build command in the message.
"""
from . import _lib  # noqa: F401  (load order: torch's HIP runtime, then libpgsd_amd.so)

try:
    from . import _fl
except ImportError as e:  # pragma: no cover - a checkout that was never built
    raise ImportError("pgsd/_fl (the Cython file layer) is not built: run `make -C pgsd-sph_amd/csrc` "
                      "(or python -c 'import __graft_entry__ as g; g.build()'): %s" % e)

from ._fl import DeviceBuffer, DeviceField, PGSDFile, open, select_rows, logger  # noqa: E402,F401
from ._fl import _is_device_tensor, _is_device_array, _pgsd_type, _NP_TO_PGSD, _PGSD_TO_NP  # noqa: E402,F401

__all__ = ["open", "PGSDFile", "DeviceField", "DeviceBuffer", "select_rows"]
