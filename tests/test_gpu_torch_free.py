"""The Python device path without a tensor library.

`pgsd.hoomd.open('w').append` with GPU-resident arrays (device elision on, the default) and `read_frame_device` run in
a subprocess where `import torch` FAILS: the source arrays are hipMalloc'ed through ctypes and handed over through
``__cuda_array_interface__`` (what HOOMD's GPU snapshots, CuPy and Numba expose), everything the path keeps in HBM
itself -- frame 0's rows and the default rows of the elision test, the arrays of the device read -- is the library's
own memory (`pgsd.fl.DeviceBuffer`).  The file is the ORACLE's file of the sketched call sequence
(tests/test_hoomd_append_oracle.py: the model of hoomd.py:569-642 + 654-694 replayed through oracle/), and the frames
read back into device memory are the values that went in."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

import product

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes, os, pickle, sys
sys.modules["torch"] = None                    # `import torch` raises ImportError from here on
root, frames_path, out_path = sys.argv[1:4]
sys.path[:0] = [os.path.join(root, "pgsd-sph_amd"), os.path.join(root, "tests")]
import numpy as np
import pgsd.fl as fl
import pgsd.hoomd as hoomd
from pgsd import _lib
assert _lib._torch is None and "torch" not in [m for m in sys.modules if sys.modules[m] is not None]
try:
    import torch
    raise SystemExit("torch imported")
except ImportError:
    pass
assert _lib.lib.pgsd_device_available()

hip = None
for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
    try:
        hip = ctypes.CDLL(name)
        break
    except OSError:
        continue
assert hip is not None
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipFree.argtypes = [ctypes.c_void_p]


class Cai:
    """A bare device array: hipMalloc + __cuda_array_interface__ (no .device, no .data_ptr)."""
    def __init__(self, host):
        host = np.ascontiguousarray(host)
        self.shape, self.typestr = host.shape, host.dtype.str
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), max(host.nbytes, 16)) == 0
        self.ptr = p.value
        assert hip.hipMemcpy(self.ptr, host.ctypes.data, host.nbytes, 1) == 0          # host -> device
    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.typestr, "data": (self.ptr, False), "version": 3, "strides": None}


import test_hoomd_append_oracle as A
frames = pickle.load(open(frames_path, "rb"))
keep = []

def device_frame(g):
    fr = A.build_frame(hoomd, g, [g["n"]], 0, False)
    p = fr.particles
    n = p.N
    if p.position is not None and p.typeid is not None:            # HOOMD's Scalar4 layout: (x, y, z, typeid bits)
        pos4 = np.zeros((n, 4), np.float32)
        pos4[:, :3] = np.asarray(p.position, np.float32).reshape(n, 3)
        pos4[:, 3] = np.asarray(p.typeid, np.uint32).view(np.float32)
        a = Cai(pos4)
        keep.append(a)
        p.position = fl.DeviceField.from_device_array(a, columns=(0, 3))
        p.typeid = fl.DeviceField.from_device_array(a, columns=(3, 4), out_dtype=np.uint32, bitcast=True)
    for name, dt in A.DTYPES.items():
        v = getattr(p, name, None)
        if name in ("value", "group") or v is None or isinstance(v, fl.DeviceField):
            continue
        a = Cai(np.ascontiguousarray(v, dtype=dt))
        keep.append(a)
        setattr(p, name, a)                                         # a bare __cuda_array_interface__ object
    return fr

with hoomd.open(out_path, "w") as t:
    assert t.device_elision is True
    for g in frames:
        t.append(device_frame(g))
    refs = dict(t._dev_ref)
    assert refs and all(isinstance(r, fl.DeviceBuffer) for r in refs.values()), refs

# ---- device read, still without torch
with hoomd.open(out_path, "r") as t:
    for k, g in enumerate(frames):
        fr = t.read_frame_device(k, scalar4=True)
        n = g["n"]
        assert fr.particles.N == n
        pos = fr.particles.position
        assert isinstance(pos, fl.DeviceBuffer) and pos.shape == (n, 3)
        want_pos = np.asarray(g["particles"]["position"], np.float32).reshape(n, 3)
        assert pos.to_host().tobytes() == want_pos.tobytes(), k
        want_tid = np.asarray(g["particles"]["typeid"], np.uint32)
        assert fr.particles.typeid.to_host().tobytes() == want_tid.tobytes(), k      # frame 0's rows where elided
        pos4 = fr.particles.pos4.to_host()
        assert pos4[:, :3].tobytes() == want_pos.tobytes() and pos4[:, 3].view(np.uint32).tobytes() == want_tid.tobytes()
        dens = fr.particles.density.to_host()
        assert dens.tobytes() == np.asarray(g["particles"]["density"], np.float32).tobytes(), k
        # an attribute the file never holds: one default row, repeated through a stride of 0
        body = fr.particles.body
        assert isinstance(body, fl.DeviceBuffer) and body.shape == (n,) and body.strides == (0,)
        assert body.view(shape=(1,)).to_host().tolist() == [-1]
        assert body.__cuda_array_interface__["strides"] == (0,)
    # select_rows on a bare flags array
    flags = (np.arange(1000) % 3 == 0).astype(np.uint8)
    index, count = fl.select_rows(Cai(flags))
    assert count == 334 and isinstance(index, fl.DeviceBuffer)
    assert index.to_host().tolist() == np.nonzero(flags)[0].tolist()
print("TORCH_FREE_OK")
'''


def test_append_and_device_read_run_without_torch(tmp_path):
    import test_gpu_elision as me
    import test_hoomd_append_oracle as A
    frames = me._one_sided_frames()
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    written = A.expected_file(ref, 1, device=True, frames=frames)
    assert "particles/density" in written[1] and "particles/density" not in written[2]      # the elision is exercised
    assert "particles/typeid" not in written[1]
    fpath = str(tmp_path / "frames.pkl")
    with open(fpath, "wb") as fh:
        pickle.dump(frames, fh)
    script = str(tmp_path / "child.py")
    with open(script, "w") as fh:
        fh.write(CHILD)
    p = subprocess.run([sys.executable, script, product.ROOT, fpath, mine], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "TORCH_FREE_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()                                  # == the oracle's file


def test_device_index_comes_from_the_pipeline_not_from_torch(tmp_gsd):
    """with bare `__cuda_array_interface__` sources the elision references used to be allocated on
    torch's CURRENT device.  Now everything the path allocates sits on the GPU the file's pipeline runs on: the ticket
    names it, the references are the library's own buffers on it (one GPU here: the checks are on the plumbing)."""
    import torch
    import pgsd.fl as fl
    pos = torch.randn((4096, 4), device="cuda")

    class Bare:                                 # no .device, no .data_ptr: only the interface
        __cuda_array_interface__ = pos.__cuda_array_interface__

    with fl.open(tmp_gsd, "w", application="a", schema="hoomd", schema_version=[1, 4]) as f:
        f.configure_device(device=0)
        t = f.stage_chunks([("particles/position", fl.DeviceField.from_device_array(Bare(), columns=(0, 3)))])
        assert t[3] == 0 == f.pipeline_device()
        (ref,) = f.copy_staged(t, 0, [t[2][0]])
        assert isinstance(ref, fl.DeviceBuffer) and ref.device == 0
        assert f.compare_staged(t, 0, [ref]) == [True]
        out = f.read_chunk_device  # (bound method exists; device reads allocate on pipeline_device() too)
        assert callable(out)
        f.write_staged(t, 0, 1, offset="auto")
        f.end_frame()


def test_a_repeating_reference_must_be_whole_rows(tmp_gsd):
    """a periodic reference that is not a multiple of the chunk's row size would be compared out of phase
    from its second repetition on; it is refused (PGSD_ERROR_INVALID_ARGUMENT -> RuntimeError)."""
    import torch
    import pgsd.fl as fl
    N = 8192
    pos = torch.zeros((N, 4), device="cuda")
    pos[:, 0] = 1.0
    with fl.open(tmp_gsd, "w", application="a", schema="hoomd", schema_version=[1, 4]) as f:
        t = f.stage_chunks([("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3)))])
        row = np.array([[1.0, 0.0, 0.0]], dtype=np.float32)
        good = fl.DeviceBuffer((4096 * 12,), np.uint8, f.pipeline_device(), pattern=row)            # 4096 whole rows
        assert f.compare_staged(t, 0, [good]) == [True]
        bad = good.view(shape=(4096,))                              # 4096 bytes: a multiple of 16, not of 12
        with pytest.raises(RuntimeError, match="Invalid pgsd argument"):
            f.compare_staged(t, 0, [bad])
        f.write_staged(t, 0, 1, offset="auto")
        f.end_frame()
