"""Child of tests/test_corrupt_files.py: open every file named on the command line (mode = first
argument), walk its index and read every chunk; with 'r+' also append a frame. Exceptions are fine and counted; the point is that the process
survives (no crash, no hang, no absurd allocation)."""
import os
import resource
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))

import pgsd.fl as fl

resource.setrlimit(resource.RLIMIT_AS, (8 << 30, 8 << 30))     # a corrupt size must not eat the machine
opened = failed = chunks = 0
import numpy

mode = sys.argv[1]          # 'r': read everything; 'r+': also append a frame to whatever opened
for path in sys.argv[2:]:
    try:
        f = fl.open(path, mode)
    except (OSError, RuntimeError, MemoryError, ValueError):
        failed += 1
        continue
    opened += 1
    try:
        names = f.find_matching_chunk_names('')
        for frame in range(min(f.nframes, 50)):
            for name in names:
                try:
                    if f.chunk_exists(frame, name):
                        f.read_chunk(frame, name)
                        chunks += 1
                except (OSError, RuntimeError, MemoryError, ValueError, KeyError):
                    pass
        if mode != 'r':
            f.write_chunk('fuzz/appended', numpy.arange(7, dtype=numpy.uint32))
            f.write_chunk('particles/position', numpy.zeros((5, 3), dtype=numpy.float32))
            f.end_frame()
    except (OSError, RuntimeError, MemoryError, ValueError):
        pass
    finally:
        try:
            f.close()
        except (OSError, RuntimeError, ValueError):
            pass
print("opened %d failed %d chunks %d" % (opened, failed, chunks))
