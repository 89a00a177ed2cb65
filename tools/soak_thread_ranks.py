#!/usr/bin/env python3
"""Eight ranks as eight threads of one process (a communicator, a file object and a device pipeline each, all on
cuda:0) appending MANY frames to one file, two of three frames sealed asynchronously, uneven partition with an empty
rank -- then the file is compared with the CPU oracle's 8-rank file of the same closed-form values.

    python tools/soak_thread_ranks.py [frames=400] [rows per rank scale=1]  [shm|rccl]

(`rccl`: the RCCL back end's code over the tests' stand-in; set PGSD_RCCL_LIBRARY and PGSD_FAKE_RCCL_SYNC=1.)"""
import os
import sys
import threading
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import pgsd.dist as pdist
import pgsd.fl as fl
import scenario as S
from test_gpu_file import _oracle_frames

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = sys.argv[3] if len(sys.argv) > 3 else "shm"
P = 8
counts = [c * scale for c in (5000, 0, 1, 77, 4096, 333, 2, 1500)]
row0 = [sum(counts[:r]) for r in range(P)]
path = "/dev/shm/pgsd_soak_threads_%d.gsd" % os.getpid()
ref = path + ".ref"
shm = "pgsdsoak_%s" % uuid.uuid4().hex[:10]
uid = pdist.rccl_unique_id() if kind == "rccl" else None
errors = []


def embed4(rows3, w):
    out = np.zeros((rows3.shape[0], 4), dtype=rows3.dtype)
    out[:, :3] = rows3
    out[:, 3] = w
    return out


def rank_main(rank):
    try:
        torch.cuda.set_device(0)
        comm = pdist.create_shm(shm, rank, P) if kind == "shm" else pdist.create_rccl(uid, rank, P, 0)
        f = fl.open(path, "w", application="app", schema="hoomd", schema_version=[1, 4], comm=comm)
        f.frame_exchange = True
        n = counts[rank]
        for k in range(frames):
            pos = S.gen_data(9, 1000 + k, row0[rank], n, 3)
            tid = S.gen_data(3, 1000 + k, row0[rank], n, 1)
            dpos = torch.from_numpy(embed4(pos, tid[:, 0].view(np.float32))).cuda()
            f.write_chunk("configuration/step", np.array([k], dtype=np.uint64), write_all=False)
            f.write_chunks([("particles/position", fl.DeviceField.from_tensor(dpos, columns=(0, 3))),
                            ("particles/typeid", fl.DeviceField.from_tensor(dpos, columns=(3, 4), out_dtype=np.uint32,
                                                                            bitcast=True))], offset="auto")
            f.end_frame(wait=(k % 3 == 0))
            f.wait_packed()
        f.close()
        pdist.release(comm)
    except Exception:  # pragma: no cover
        import traceback
        errors.append((rank, traceback.format_exc()))


t0 = time.perf_counter()
threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=900)
dt = time.perf_counter() - t0
if errors or any(t.is_alive() for t in threads):
    sys.stderr.write(repr(errors))
    os._exit(1)
out = []
for k in range(frames):
    pos = [S.gen_data(9, 1000 + k, row0[r], counts[r], 3) for r in range(P)]
    tid = [S.gen_data(3, 1000 + k, row0[r], counts[r], 1) for r in range(P)]
    out.append([("configuration/step", 4, 1, False, [np.array([[k]], dtype=np.uint64)] * P),
                ("particles/position", 9, 3, True, pos), ("particles/typeid", 3, 1, True, tid)])
_oracle_frames(ref, P, out)
same = open(path, "rb").read() == open(ref, "rb").read()
size = os.path.getsize(path)
os.unlink(path)
os.unlink(ref)
print("%s: %d ranks as threads x %d frames (%d rows in all), %.1f s, file %.1f MB, equal to the oracle's file: %s"
      % (kind, P, frames, sum(counts), dt, size / 1e6, same))
sys.exit(0 if same else 1)
