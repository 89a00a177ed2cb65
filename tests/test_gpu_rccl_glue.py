"""The RCCL back end's multi-rank glue (pgsd_comm_rccl.cpp, pgsd.dist.init_from_torch, benchmark_write.hip's
bootstrap) at MORE THAN ONE rank on a one-GPU box.

RCCL itself refuses two ranks on one device, so these runs load tests/drivers/fake_rccl.cpp through
PGSD_RCCL_LIBRARY: a stand-in with RCCL's semantics for the five entry points the communicator uses (per-rank
sendcount, rank-ordered receive buffer, stream order) and a shared-memory transport.  Everything above those five
calls is the product's own code running for real: who makes the unique id and how it reaches the others, the
exchange buffers and their growth, the barrier as a one-byte allgather, one exchange per frame, the file offsets
derived from the gathered sizes, teardown.  RCCL's transport over xGMI is NOT covered; that takes a multi-GPU node."""
import glob
import json
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

import product

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(product.CSRC, "build", "libpgsd_fake_rccl.so")
WRITE_EXE = os.path.join(product.CSRC, "build", "benchmark_write")
READ_EXE = os.path.join(product.CSRC, "build", "benchmark_read")


def _fake_logs(prefix, P):
    logs = [json.load(open("%s.%d" % (prefix, r))) for r in range(P)]
    assert [d["rank"] for d in logs] == list(range(P)) and all(d["size"] == P for d in logs)
    return logs


@pytest.mark.parametrize("P", [2, 3])
def test_native_bootstrap_and_frame_exchange_over_the_rccl_back_end(P, tmp_path):
    """benchmark_write.hip ... rccl: rank 0's ncclUniqueId travels over the shm communicator of the launch, every
    rank calls pgsd_comm_init_rccl, and from then on every exchange of the run is an ncclAllGather issued by
    pgsd_comm_rccl.cpp -- one per frame.  The file is then read back and verified by benchmark_read.hip."""
    product.build()
    assert os.path.exists(FAKE)
    shm = "pgsdrccl_%s" % uuid.uuid4().hex[:10]
    path = str(tmp_path / "rccl.gsd")
    log = str(tmp_path / "fake")
    per_rank, frames = 70001, 4
    env = dict(os.environ, PGSD_NRANKS=str(P), PGSD_SHM_NAME=shm, PGSD_RCCL_LIBRARY=FAKE, PGSD_FAKE_RCCL_LOG=log)
    procs = [subprocess.Popen([WRITE_EXE, str(per_rank), str(frames), path, "batched", "rccl", "keep"],
                              env=dict(env, PGSD_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(P)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o[-300:], e[-1500:])
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["ranks"] == P and d["comm"] == "rccl" and d["exchange"] == "one per frame"
    assert d["collectives_rank0"] == 2 + (frames + 1)             # create/open + ONE per frame
    logs = _fake_logs(log, P)
    # every rank went through ncclAllGather the same number of times: the handle's collectives, the harness's own
    # partition_rows / barriers, and pgsd_close's barrier
    assert len({l["allgathers"] for l in logs}) == 1 and logs[0]["allgathers"] >= d["collectives_rank0"]
    r = subprocess.run([READ_EXE, path, "verify"], env=dict(os.environ, PGSD_RANK="0", PGSD_NRANKS="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    v = json.loads(r.stdout.strip().splitlines()[-1])
    assert v["verified"] is True and v["particles"] == per_rank * P and v["frames"] == frames + 1


RANK = r'''
import os, sys
sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(tests)r)
import numpy as np, torch, torch.distributed as dist
rank, P = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"
dist.init_process_group(backend="gloo", rank=rank, world_size=P)
torch.cuda.set_device(0)
import pgsd.dist as pdist, pgsd.fl as fl, scenario as S
name = pdist.init_from_torch(device=0)
counts = [int(c) for c in sys.argv[4].split(",")]
n, row0 = counts[rank], sum(counts[:rank])
got, r0, ng = pdist.partition_rows(n)
assert [int(c) for c in got] == counts and r0 == row0 and ng == sum(counts), (got, r0, ng)
f = fl.open(sys.argv[3], 'w', application='app', schema='hoomd', schema_version=[1, 4])
f.frame_exchange = True
for frame in range(3):
    pos = torch.from_numpy(S.gen_data(9, 40 + frame, row0, n, 4)).cuda()
    tid = torch.from_numpy(S.gen_data(3, 40 + frame, row0, n, 1).view(np.int32)).cuda()
    f.write_chunk('configuration/step', np.array([frame], dtype=np.uint64), write_all=False)
    f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                    ('particles/typeid', fl.DeviceField.from_tensor(tid, out_dtype=np.uint32))], offset="auto")
    f.end_frame()
per_frame = (f.collective_count - 2) / 3.0
f.close()
dist.barrier()
pdist.finalize()
if rank == 0:
    print("RESULT", name, per_frame)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("counts", [[900, 1201], [0, 4097, 1, 333]])
def test_init_from_torch_builds_the_rccl_back_end_at_several_ranks(counts, tmp_path):
    """pgsd.dist.init_from_torch over a gloo group whose ranks share cuda:0: id from rank 0 by torch broadcast,
    pgsd_comm_init_rccl on every rank, self-check exchange, agreement; then a batched device write whose only
    collective per frame is the back end's allgather.  The file equals the oracle's P-rank file."""
    import scenario as S
    from test_gpu_file import _oracle_frames
    product.build()
    P = len(counts)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    log = str(tmp_path / "fake")
    port = str(29600 + os.getpid() % 300)
    env = dict(os.environ, PGSD_RCCL_LIBRARY=FAKE, PGSD_FAKE_RCCL_LOG=log, MASTER_PORT=port)
    code = RANK % {"pkg": os.path.join(ROOT, "pgsd-sph_amd"), "tests": os.path.join(ROOT, "tests")}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(P), mine, ",".join(map(str, counts))], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(P)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o[-300:], e[-2000:])
    result = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT")][-1].split()
    assert result[1] == "rccl[libpgsd_fake_rccl.so]"          # the native back end, with the stand-in named
    assert float(result[2]) == 1.0                             # ONE collective per frame
    logs = _fake_logs(log, P)
    assert len({l["allgathers"] for l in logs}) == 1 and logs[0]["allgathers"] >= 1 + 1 + 2 + 3
    row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
    frames = []
    for frame in range(3):
        pos = [S.gen_data(9, 40 + frame, int(row0[r]), counts[r], 4)[:, :3].copy() for r in range(P)]
        tid = [S.gen_data(3, 40 + frame, int(row0[r]), counts[r], 1) for r in range(P)]
        frames.append([('configuration/step', 4, 1, False, [np.array([[frame]], dtype=np.uint64)] * P),
                       ('particles/position', 9, 3, True, pos), ('particles/typeid', 3, 1, True, tid)])
    _oracle_frames(ref, P, frames)
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()
