"""Version of the package; tracks the reference it is a drop-in for (version.py:12)."""
__version__ = "3.2.0"
version = __version__
