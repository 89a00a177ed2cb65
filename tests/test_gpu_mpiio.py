"""PGSD_IO=mpiio together with the DEVICE pipeline: the packed chunks leave HBM through pinned slabs and the pipeline's
writer thread hands every piece to MPI_File_write_at (plugin libpgsd_amd_mpiio.so); device reads come back through
MPI_File_read_at on the reader threads.  MPI is initialised in-process through ctypes (MPI_Init_thread, singleton, with
MPI_THREAD_SERIALIZED: the library serialises its calls into the plugin).  The file must be byte-identical to the one the
POSIX back end writes for the same frames; the host-path counterpart (all 48 goldens under mpiexec) is
tests/test_product_golden.py."""
import os
import subprocess
import sys

import pytest

import product

pytestmark = pytest.mark.gpu

LIBMPI = "/opt/conda/lib/libmpi.so.12"
PLUGIN = os.path.join(product.ROOT, "pgsd-sph_amd", "pgsd", "libpgsd_amd_mpiio.so")

CHILD = r'''
import ctypes, os, sys
root, out, use_mpi = sys.argv[1], sys.argv[2], sys.argv[3] == "mpiio"
sys.path[:0] = [os.path.join(root, "pgsd-sph_amd"), os.path.join(root, "tests")]
mpi = None
if use_mpi:
    os.environ["PGSD_IO"] = "mpiio"
    mpi = ctypes.CDLL(%r, mode=ctypes.RTLD_GLOBAL)
    provided = ctypes.c_int(-1)
    assert mpi.MPI_Init_thread(None, None, 2, ctypes.byref(provided)) == 0          # MPI_THREAD_SERIALIZED
    assert provided.value >= 2, provided.value
import numpy as np
import torch
import pgsd.fl as fl
from pgsd import _lib
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(11)
N = 1 << 20                                              # 28 MB per frame: the staged road, pieces through the writer thread
pos = torch.randn((N, 4), generator=g, device="cuda")
vel = torch.randn((N, 4), generator=g, device="cuda", dtype=torch.float64)
small = torch.randn((300, 4), generator=g, device="cuda")    # a frame of <= 2 MiB: the direct road (pwritev -> one write per buffer)
f = fl.open(out, "w", application="mpiio", schema="hoomd", schema_version=[1, 4])
f.frame_exchange = True
for k in range(3):
    f.write_chunk("configuration/step", np.array([k], dtype=np.uint64), write_all=False)
    f.write_chunks([("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                    ("particles/typeid", fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True)),
                    ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3), out_dtype=np.float32))], offset="auto")
    f.end_frame(wait=(k != 1))                           # one frame sealed asynchronously
    f.wait_packed()
    if k == 2:
        want = pos[:, :3].contiguous().clone()
    pos[:, :3] += 1.0
f.write_chunks([("particles/position", fl.DeviceField.from_tensor(small, columns=(0, 3)))], offset="auto")
f.end_frame()
f.close()
r = fl.open(out, "r")
back = r.read_chunk_device(2, "particles/position")     # pread on the reader threads -> MPI_File_read_at
assert torch.equal(back.view(torch.int32), want.view(torch.int32))
assert r.read_chunk(3, "particles/position").tobytes() == small[:, :3].contiguous().cpu().numpy().tobytes()
r.close()
if mpi is not None:
    assert mpi.MPI_Finalize() == 0
print("DONE", "mpiio" if use_mpi else "posix")
''' % LIBMPI


def test_device_pipeline_over_mpiio_writes_the_posix_file(tmp_path):
    if not (os.path.exists(LIBMPI) and os.path.exists(PLUGIN)):
        pytest.skip("no MPI installation / MPI-IO plugin on this box")
    script = str(tmp_path / "child.py")
    with open(script, "w") as fh:
        fh.write(CHILD)
    files = {}
    for mode in ("posix", "mpiio"):
        files[mode] = str(tmp_path / (mode + ".gsd"))
        env = {k: v for k, v in os.environ.items() if k != "PGSD_IO"}
        p = subprocess.run([sys.executable, script, product.ROOT, files[mode], mode], capture_output=True, text=True,
                           timeout=600, env=env)
        assert p.returncode == 0 and ("DONE " + mode) in p.stdout, (mode, p.stdout[-1500:], p.stderr[-3000:])
    with open(files["posix"], "rb") as a, open(files["mpiio"], "rb") as b:
        assert a.read() == b.read()
