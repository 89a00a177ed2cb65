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
FAKE = os.path.join(product.TBUILD, "libpgsd_fake_rccl.so")
WRITE_EXE = os.path.join(product.CSRC, "build", "benchmark_write")
READ_EXE = os.path.join(product.CSRC, "build", "benchmark_read")


def _fake_logs(prefix, P):
    logs = [json.load(open("%s.%d" % (prefix, r))) for r in range(P)]
    assert [d["rank"] for d in logs] == list(range(P)) and all(d["size"] == P for d in logs)
    return logs


@pytest.mark.parametrize("P", [2, 3])
def test_native_bootstrap_and_frame_exchange_over_the_rccl_back_end(P, tmp_path):
    """benchmark_write.hip ... rccl: rank 0's ncclUniqueId travels over the shm communicator of the launch, every
    rank calls pgsd_comm_create_rccl + pgsd_comm_set_default, and from then on every exchange of the run is an ncclAllGather issued by
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
    pgsd_comm_create_rccl on every rank, self-check exchange, agreement; then a batched device write whose only
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


ABSENT = r'''
import ctypes, os, sys, threading, time
sys.path.insert(0, %(pkg)r)
import torch
import pgsd.dist as pdist
from pgsd import _lib
P = 3
uid = pdist.rccl_unique_id()
out, errors = {}, []
gate = threading.Barrier(P)

def exchange(comm, value):
    send, recv = ctypes.c_uint64(value), (ctypes.c_uint64 * P)()
    t0 = time.perf_counter()
    rc = comm.allgather(comm.ctx, ctypes.byref(send), recv, 8)
    return rc, list(recv), time.perf_counter() - t0, _lib.last_error()

def rank_main(rank):
    try:
        torch.cuda.set_device(0)
        comm = pdist.create_rccl(uid, rank, P, 0)
        first = exchange(comm, 100 + rank)              # everybody is there
        assert first[0] == 0 and first[1] == [100, 101, 102], first
        gate.wait()
        if rank != 2:                                    # rank 2 never comes to the second exchange
            second = exchange(comm, 200 + rank)
            third = exchange(comm, 300 + rank)
            out[rank] = (second, third)
        gate.wait()
        pdist.release(comm)
    except Exception:
        import traceback
        errors.append((rank, traceback.format_exc()))

threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
for t in threads: t.start()
for t in threads: t.join(timeout=100)
if errors or any(t.is_alive() for t in threads):
    sys.stderr.write(repr(errors)); sys.stderr.flush(); os._exit(1)
for rank in (0, 1):
    (rc2, _, s2, msg2), (rc3, _, s3, msg3) = out[rank]
    print("RANK", rank, rc2, "%%.2f" %% s2, msg2.replace(" ", "_"), rc3, "%%.3f" %% s3, msg3.replace(" ", "_"))
'''


def test_a_rank_that_never_arrives_is_a_comm_error_inside_the_deadline(tmp_path):
    """`rccl_allgather` ended in an unbounded hipStreamSynchronize -- a rank that never enters the
    collective hung every other rank until the driver's time limit (the reference's MPI_Allgather, pgsd.c:1126, waits
    for ever too).  Three ranks (threads) on the asynchronous stand-in: the third stays away from the second exchange;
    the others come back with an error after PGSD_COMM_TIMEOUT_S, the communicator aborted (ncclCommAbort: the
    waiting kernel leaves the stream), and their next collective fails at once."""
    product.build()
    env = dict(os.environ, PGSD_RCCL_LIBRARY=FAKE, PGSD_COMM_TIMEOUT_S="2", PGSD_FAKE_RCCL_MAX_WAIT_S="30",
               PGSD_FAKE_RCCL_LOG=str(tmp_path / "fake"))
    p = subprocess.run([sys.executable, "-c", ABSENT % {"pkg": os.path.join(ROOT, "pgsd-sph_amd")}], env=env,
                       capture_output=True, text=True, timeout=200)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-3000:])
    lines = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("RANK")]
    assert [ln[1] for ln in lines] == ["0", "1"]
    for _, _, rc2, s2, msg2, rc3, s3, msg3 in lines:
        assert int(rc2) != 0 and 2.0 <= float(s2) < 12.0 and "timed_out" in msg2 and "aborted" in msg2, (rc2, s2, msg2)
        assert int(rc3) != 0 and float(s3) < 0.5 and "broken" in msg3, (rc3, s3, msg3)
    ends = [json.load(open(str(tmp_path / "fake") + ".%d" % r))["end"] for r in range(3)]
    assert ends == ["abort", "abort", "destroy"]


def test_an_exchange_rccl_gives_up_by_itself_is_not_trusted(tmp_path):
    """The other way an exchange ends without its peers: the collective library gives up on its own (here the
    stand-in's waiting kernel leaves at PGSD_FAKE_RCCL_MAX_WAIT_S, long before PGSD_COMM_TIMEOUT_S) and the stream runs
    dry with nothing gathered.  `ncclCommGetAsyncError` says so, and the call fails instead of returning stale bytes."""
    product.build()
    env = dict(os.environ, PGSD_RCCL_LIBRARY=FAKE, PGSD_COMM_TIMEOUT_S="60", PGSD_FAKE_RCCL_MAX_WAIT_S="2",
               PGSD_FAKE_RCCL_LOG=str(tmp_path / "fake"))
    p = subprocess.run([sys.executable, "-c", ABSENT % {"pkg": os.path.join(ROOT, "pgsd-sph_amd")}], env=env,
                       capture_output=True, text=True, timeout=200)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-3000:])
    lines = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("RANK")]
    assert [ln[1] for ln in lines] == ["0", "1"]
    for _, _, rc2, s2, msg2, rc3, s3, msg3 in lines:
        assert int(rc2) != 0 and 1.5 <= float(s2) < 15.0 and "asynchronously" in msg2, (rc2, s2, msg2)
        assert int(rc3) != 0 and float(s3) < 0.5 and "broken" in msg3, (rc3, s3, msg3)


STALLED = r'''
import os, sys, time
sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(tests)r)
import numpy as np, torch, torch.distributed as dist
rank, P = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"
dist.init_process_group(backend="gloo", rank=rank, world_size=P)
torch.cuda.set_device(0)
import pgsd.dist as pdist, pgsd.fl as fl
name = pdist.init_from_torch(device=0)
f = fl.open(sys.argv[3], 'w', application='app', schema='hoomd', schema_version=[1, 4])
f.frame_exchange = True
n = 5000 + rank
pos = torch.randn((n, 4), device="cuda")
done, failed_after, message = 0, None, ""
for frame in range(6):
    t0 = time.perf_counter()
    try:
        f.write_chunk('configuration/step', np.array([frame], dtype=np.uint64), write_all=False)
        f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3)))], offset="auto")
        f.end_frame()
        done += 1
    except Exception as e:
        failed_after, message = time.perf_counter() - t0, "%%s: %%s" %% (type(e).__name__, e)
        break
t0 = time.perf_counter()
try:
    f.close()
    closed = "closed"
except Exception as e:
    closed = type(e).__name__
print("RESULT", rank, done, "%%.2f" %% (failed_after or -1), "%%.2f" %% (time.perf_counter() - t0), closed, message.replace(" ", "_"))
sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
'''


def test_a_rank_hanging_inside_the_frame_exchange_fails_the_frame_on_every_rank(tmp_path):
    """The same through the write path: two ranks (processes sharing cuda:0, communicator from
    pgsd.dist.init_from_torch) append frames with one exchange per frame; the stand-in lets rank 1 hang inside one of
    its allgathers.  Both ranks' end_frame raises inside the deadline instead of hanging, close() returns, the
    processes end by themselves."""
    product.build()
    P = 2
    port = str(29300 + os.getpid() % 300)
    env = dict(os.environ, PGSD_RCCL_LIBRARY=FAKE, MASTER_PORT=port, PGSD_COMM_TIMEOUT_S="2",
               PGSD_FAKE_RCCL_MAX_WAIT_S="30", PGSD_FAKE_RCCL_STALL_RANK="1", PGSD_FAKE_RCCL_STALL_AT="6")
    code = STALLED % {"pkg": os.path.join(ROOT, "pgsd-sph_amd"), "tests": os.path.join(ROOT, "tests")}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(P), str(tmp_path / "stall.gsd")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(P)]
    outs = [p.communicate(timeout=200) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o[-300:], e[-2000:])
    for r, (o, _) in enumerate(outs):
        res = [ln for ln in o.splitlines() if ln.startswith("RESULT")][-1].split()
        done, failed_after, close_s, msg = int(res[2]), float(res[3]), float(res[4]), res[6]
        assert done < 6 and 2.0 <= failed_after < 15.0, res          # the frame failed, after the deadline, not for ever
        assert close_s < 10.0, res
        assert "RuntimeError" in msg and ("timed_out" in msg or "broken" in msg or "communicator" in msg), res


UNAVAILABLE = r'''
import os, sys
sys.path.insert(0, %(pkg)r)
import torch, torch.distributed as dist
rank, P = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"
dist.init_process_group(backend="gloo", rank=rank, world_size=P)
torch.cuda.set_device(0)
import pgsd.dist as pdist
try:
    print("RESULT", rank, pdist.init_from_torch(device=0))
except RuntimeError as e:
    print("RESULT", rank, "RuntimeError", str(e).replace(" ", "_"))
sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
'''


def test_ranks_agree_that_librccl_is_there_before_anyone_enters_the_bootstrap(tmp_path):
    """ncclCommInitRank waits for every rank, so a rank whose librccl does not load must be
    known to all BEFORE anyone enters it.  Rank 1 is given a library path that does not exist: both ranks raise the
    same error at once (nobody is left inside the bootstrap)."""
    product.build()
    P = 2
    port = str(29000 + os.getpid() % 300)
    code = UNAVAILABLE % {"pkg": os.path.join(ROOT, "pgsd-sph_amd")}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(P)],
                              env=dict(os.environ, MASTER_PORT=port,
                                       PGSD_RCCL_LIBRARY=FAKE if r == 0 else "/nonexistent/librccl.so"),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(P)]
    outs = [p.communicate(timeout=120) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (o[-300:], e[-2000:])
        res = [ln for ln in o.splitlines() if ln.startswith("RESULT")][-1].split()
        assert res[2] == "RuntimeError" and "not_available_on_at_least_one_rank" in res[3], res
        assert ("/nonexistent/librccl.so" in res[3]) == (r == 1), res
