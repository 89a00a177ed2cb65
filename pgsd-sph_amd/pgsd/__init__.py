"""pgsd: MI355X-native writer/reader of PGSD (GSD v2) snapshot files.

Same import names as the reference package (``pgsd.fl``, ``pgsd.hoomd``, ``pgsd.pypgsd``,
``pgsd.version``); see DESIGN.md for what lives behind them.
"""
import signal
import sys

from . import version as _version_module
from .version import __version__

version = __version__


def _sigterm_handler(signum, frame):
    # let open files flush on SIGTERM (reference __init__.py:23-26)
    sys.exit(1)


try:
    signal.signal(signal.SIGTERM, _sigterm_handler)
except ValueError:  # not in the main thread
    pass
