"""Child process of tests/test_io_errors.py: writes frames until the file-size limit of THIS rank
makes pwrite fail, and reports what every call raised.

usage: io_error_worker.py <path> <rank> <nranks> <shm name> <limited: 0|1> <device: 0|1> [batched: 0|1]"""
import json
import os
import resource
import signal
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))

import numpy as np

path, rank, nranks, shm, limited, device = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], \
    int(sys.argv[5]), int(sys.argv[6])
batched = len(sys.argv) > 7 and sys.argv[7] == "1"
signal.signal(signal.SIGXFSZ, signal.SIG_IGN)      # a write past the limit returns EFBIG instead of killing us

import pgsd.dist as pdist
import pgsd.fl as fl

if nranks > 1:
    pdist.init_shm(shm, rank, nranks)
N = int(os.environ.get("PGSD_IOERR_N", "300000"))      # 3.6 MB per frame: the pipeline; 100 000 -> 1.2 MB: the direct path
counts = np.array([N] * nranks, dtype=np.uint64)
report = {"rank": rank, "events": []}
f = fl.open(path, 'w', application='io-error test', schema='s', schema_version=[1, 0])
if batched:
    f.frame_exchange = True     # one exchange per frame: a failure reaches the other ranks with the next one
if device:
    import torch
    data = torch.arange(N * 3, dtype=torch.float32, device="cuda").reshape(N, 3) + rank
    field = fl.DeviceField.from_tensor(data)
else:
    field = (np.arange(N * 3, dtype=np.float32) + rank).reshape(N, 3)


def record(tag, fn):
    try:
        fn()
        report["events"].append([tag, "ok"])
    except Exception as e:  # noqa: BLE001 - the parent inspects the type
        report["events"].append([tag, type(e).__name__, getattr(e, "errno", None)])


def frame(tag):
    # the calls are collective: a rank whose write_chunk failed still seals the frame, which is where
    # the other ranks learn of the failure
    if device:
        record(tag + ":write", lambda: f.write_chunks([('particles/position', field)], offset=counts, rank=rank))
    else:
        record(tag + ":write", lambda: f.write_chunk('particles/position', field, offset=counts, rank=rank))
    record(tag + ":end_frame", f.end_frame)


frame("before")
if limited:
    hard = resource.getrlimit(resource.RLIMIT_FSIZE)[1]
    resource.setrlimit(resource.RLIMIT_FSIZE, (1 << 20, hard))          # the next frame does not fit
frame("after")
if batched:
    frame("later")
record("close", f.close)
if nranks > 1:
    pdist.finalize()
print(json.dumps(report))
