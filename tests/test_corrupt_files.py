"""Damaged files must be rejected or read with errors, never crash the reader: truncations and byte
flips of golden files (header, index, namelist and data regions), opened read-only in a child
process. The reference guards its open path the same way (pgsd.c:1532-1703: magic, version, index and
namelist bounds, frame order)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from scenario import GOLDEN as GOLDEN_DIR

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "corrupt_worker.py")
SOURCES = ["posvelid.p2.gsd", "sph_full.p1.gsd", "index_expand.p1.gsd", "names_reloc.p1.gsd", "alltypes.p1.gsd",
           "reference_test_gsd_v1.gsd"]


def mutate(raw, rng):
    b = bytearray(raw)
    kind = rng.integers(0, 5)
    if kind == 0:                                   # truncate anywhere
        return bytes(b[:int(rng.integers(0, len(b)))])
    if kind == 1:                                   # flip bytes in the header
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, 256))] ^= int(rng.integers(1, 256))
        return bytes(b)
    if kind == 2:                                   # overwrite one header field with an extreme value
        off = int(rng.choice([8, 16, 24, 32, 40, 48, 52]))
        val = int(rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, len(b), len(b) + 1]))
        width = 4 if off >= 48 else 8
        b[off:off + width] = (val & ((1 << (8 * width)) - 1)).to_bytes(width, 'little')
        return bytes(b)
    if kind == 3:                                   # damage index / namelist blocks (right after the header)
        for _ in range(int(rng.integers(1, 40))):
            b[int(rng.integers(256, min(len(b), 256 + 4096 + 1024)))] = int(rng.integers(0, 256))
        return bytes(b)
    for _ in range(int(rng.integers(1, 200))):      # anywhere
        b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    return bytes(b)


@pytest.mark.parametrize("mode", ["r", "r+"])
@pytest.mark.parametrize("source", SOURCES)
def test_damaged_files_do_not_crash_the_reader(source, mode, tmp_path):
    path = os.path.join(GOLDEN_DIR, source)
    if not os.path.exists(path):
        pytest.fail("golden file missing: " + source)
    raw = open(path, 'rb').read()
    rng = np.random.default_rng(sum(source.encode()))
    files = []
    for i in range(120 if mode == 'r' else 60):
        name = str(tmp_path / ("m%03d.gsd" % i))
        with open(name, 'wb') as f:
            f.write(mutate(raw, rng))
        files.append(name)
    p = subprocess.run([sys.executable, WORKER, mode] + files, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, "reader died (%s): %s" % (p.returncode, p.stderr[-1500:])
    assert p.stdout.startswith("opened"), p.stdout
