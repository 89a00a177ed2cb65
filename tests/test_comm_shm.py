"""The /dev/shm communicator survives what a crashed earlier run leaves behind."""
import os
import subprocess
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import pgsd.dist as d
d.init_shm(sys.argv[1], int(sys.argv[2]), 2)
if sys.argv[3] == "crash":
    d.partition_rows(1)  # both ranks are attached
    os._exit(0)          # no finalize: rank 0 never unlinks the segment
c, row0, n = d.partition_rows(10 + int(sys.argv[2]))
print(int(sys.argv[2]), [int(x) for x in c], row0, n)
d.finalize()
''' % os.path.join(ROOT, "pgsd-sph_amd")


def test_stale_segment_of_a_crashed_run_is_not_joined():
    name = "pgsd_stale_%s" % uuid.uuid4().hex[:10]
    try:
        ps = [subprocess.Popen([sys.executable, "-c", CHILD, name, str(r), "crash"]) for r in (0, 1)]
        for p in ps:
            assert p.wait(timeout=120) == 0
        assert os.path.exists("/dev/shm/" + name)              # the crashed run's segment
        # the next run under the same name: rank 1 arrives a second before rank 0 and finds the old segment
        p1 = subprocess.Popen([sys.executable, "-c", CHILD, name, "1", "run"], stdout=subprocess.PIPE)
        time.sleep(1.0)
        p0 = subprocess.Popen([sys.executable, "-c", CHILD, name, "0", "run"], stdout=subprocess.PIPE)
        try:
            out0 = p0.communicate(timeout=120)[0].decode().strip()
            out1 = p1.communicate(timeout=120)[0].decode().strip()
        finally:
            for p in (p0, p1):
                if p.poll() is None:
                    p.kill()
        assert out0 == "0 [10, 11] 0 21" and out1 == "1 [10, 11] 10 21"
        assert not os.path.exists("/dev/shm/" + name)
    finally:
        try:
            os.unlink("/dev/shm/" + name)
        except OSError:
            pass


PEER_DIES = r'''
import sys, os, time
sys.path.insert(0, %r)
import pgsd.dist as d
rank = int(sys.argv[2])
d.init_shm(sys.argv[1], rank, 2)
d.partition_rows(5)                   # one good exchange
if rank == 1:
    os._exit(0)                       # dies without a word
t0 = time.time()
try:
    d.partition_rows(5)
    print("NO ERROR")
except RuntimeError as e:
    print("error after %%.1f s: %%s" %% (time.time() - t0, e))
d.finalize()
''' % os.path.join(ROOT, "pgsd-sph_amd")


def test_a_dead_peer_is_an_error_not_a_hang():
    """MPI (and a futex barrier) would wait forever for the rank that is gone; the polling barrier
    of the shm communicator checks on its peers while it waits."""
    name = "pgsd_dead_%s" % uuid.uuid4().hex[:10]
    try:
        ps = [subprocess.Popen([sys.executable, "-c", PEER_DIES, name, str(r)], stdout=subprocess.PIPE) for r in (0, 1)]
        try:
            out0 = ps[0].communicate(timeout=60)[0].decode().strip()
            ps[1].wait(timeout=60)
        finally:
            for p in ps:
                if p.poll() is None:
                    p.kill()
        assert out0.startswith("error after") and "gone" in out0, out0
    finally:
        try:
            os.unlink("/dev/shm/" + name)
        except OSError:
            pass


OUT_OF_STEP = r'''
import sys, os, ctypes
sys.path.insert(0, %r)
import pgsd.dist as d
from pgsd import _lib
rank = int(sys.argv[2])
d.init_shm(sys.argv[1], rank, 2)
d.partition_rows(5)                   # one good exchange
n = 8 if rank == 0 else 16            # rank 0 is in one collective, rank 1 in another
send = (ctypes.c_uint8 * n)()
recv = (ctypes.c_uint8 * (2 * n))()
rc = _lib.lib.pgsd_comm_allgather(send, recv, n)
print(rc, _lib.last_error())
try:
    d.partition_rows(5)               # the communicator is broken for good: no rank goes on with the other's bytes
    print("NO ERROR")
except RuntimeError as e:
    print("then:", e)
d.finalize()
''' % os.path.join(ROOT, "pgsd-sph_amd")


def test_ranks_in_different_collectives_are_told_so():
    """A collective call made by some ranks only (a collective read on one rank, say) would pair its message with a
    different exchange of the others; the shm allgather compares the message sizes and fails on EVERY rank."""
    name = "pgsd_step_%s" % uuid.uuid4().hex[:10]
    try:
        ps = [subprocess.Popen([sys.executable, "-c", OUT_OF_STEP, name, str(r)], stdout=subprocess.PIPE) for r in (0, 1)]
        outs = []
        try:
            for p in ps:
                outs.append(p.communicate(timeout=60)[0].decode().strip().splitlines())
        finally:
            for p in ps:
                if p.poll() is None:
                    p.kill()
        for out in outs:
            assert out[0].startswith("-") and "different collectives" in out[0], outs
            assert out[1].startswith("then:"), outs
    finally:
        try:
            os.unlink("/dev/shm/" + name)
        except OSError:
            pass


def test_eight_ranks_as_threads_of_one_process_on_per_handle_communicators(tmp_path):
    """`pgsd_comm_create_shm` + `pgsd_create_and_open_on`: communicators that are not the process default, one per
    thread, eight ranks in ONE process writing config 3's chunk sequence from host arrays -- the file is the
    reference-written golden posvelid.p8.gsd (the GPU twin of this test: tests/test_gpu_eight_ranks.py)."""
    import subprocess
    import sys
    import scenario as S
    mine = str(tmp_path / "p8.gsd")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "thread_ranks_worker.py")
    p = subprocess.run([sys.executable, worker, "shm", "8", mine, "posvelid", "host"], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-3000:])
    with open(mine, "rb") as a, open(os.path.join(S.GOLDEN, "posvelid.p8.gsd"), "rb") as b:
        assert a.read() == b.read()


CLOSE_AFTER_PEER_DIED = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
import pgsd.dist as d
import pgsd.fl as fl
from pgsd import _lib
rank = int(sys.argv[2])
d.init_shm(sys.argv[1], rank, 2)
f = fl.open(sys.argv[3], "w", application="a", schema="s", schema_version=[1, 0])
f.frame_exchange = True               # chunk writes queue; the frame's ONE exchange happens at end_frame / flush / close
f.write_chunk("x", np.arange(8, dtype=np.float32), offset=np.array([8, 8]), rank=rank)
f.end_frame()
if rank == 1:
    os._exit(0)                       # dies with the file open
f.write_chunk("y", np.arange(8, dtype=np.float32), write_all=False)     # queued: nothing is exchanged yet
n_fds = len(os.listdir("/proc/self/fd"))
try:
    f.close()
    print("NO ERROR")
except RuntimeError as e:
    print("close failed:", str(e)[:120].replace("\n", " "))
print("fds released:", n_fds - len(os.listdir("/proc/self/fd")))
h = f._h() if hasattr(f, "_h") else None
try:
    f.close()                         # the handle is gone: a second close is the usual "not open" complaint, no hang
    print("second close ok")
except Exception as e:
    print("second close:", type(e).__name__)
d.finalize()
''' % os.path.join(ROOT, "pgsd-sph_amd")


def test_close_on_a_broken_communicator_abandons_the_handle(tmp_path):
    """after a peer died (or an RCCL exchange timed out) the flush inside pgsd_close fails with
    PGSD_ERROR_COMM for ever; the close used to return without releasing the descriptor, the threads and the
    buffers, and `PGSDFile.close()` marked the object closed anyway -- the handle leaked.  Now the handle is
    ABANDONED: the error is returned and this rank's resources are released."""
    name = "pgsd_abandon_%s" % uuid.uuid4().hex[:10]
    path = str(tmp_path / "a.gsd")
    try:
        ps = [subprocess.Popen([sys.executable, "-c", CLOSE_AFTER_PEER_DIED, name, str(r), path], stdout=subprocess.PIPE)
              for r in (0, 1)]
        try:
            out0 = ps[0].communicate(timeout=120)[0].decode().strip().splitlines()
            ps[1].wait(timeout=60)
        finally:
            for p in ps:
                if p.poll() is None:
                    p.kill()
        assert out0[0].startswith("close failed:") and "communicator" in out0[0], out0
        assert out0[1].startswith("fds released:") and int(out0[1].split(":")[1]) >= 1, out0
    finally:
        try:
            os.unlink("/dev/shm/" + name)
        except OSError:
            pass


LOCAL_LOOKUP = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
import pgsd.dist as d
import pgsd.fl as fl
from pgsd import _lib
rank = int(sys.argv[2])
d.init_shm(sys.argv[1], rank, 2)
f = fl.open(sys.argv[3], "w", application="a", schema="s", schema_version=[1, 0])
f.write_chunk("big", np.arange(16, dtype=np.float32), offset=np.array([16, 16]), rank=rank)
f.end_frame()                                              # frame 0: a direct chunk -> flushed
f.write_chunk("small", np.arange(4, dtype=np.int32), write_all=False)
f.end_frame()                                              # frame 1: a buffered small chunk only -> NOT flushed (pgsd.c:1941-1950)
f.local_reads = True
found0 = f.chunk_exists(0, "big")
found1 = f.chunk_exists(1, "small")                        # local lookup: sees what the last collective flush committed
print(rank, found0, found1, "|", _lib.last_error())
f.local_reads = False
f.flush()
print(rank, "after flush", f.chunk_exists(1, "small"))
f.close()
d.finalize()
''' % os.path.join(ROOT, "pgsd-sph_amd")


def test_a_local_lookup_that_misses_pending_metadata_says_so(tmp_path):
    """with pgsd_set_local_reads and several ranks a lookup never starts the collective flush; a chunk of a
    frame that is sealed but not yet committed is then reported missing -- now with a note in pgsd_last_error_string()
    instead of no signal at all; after the next collective flush it is found."""
    name = "pgsd_locallook_%s" % uuid.uuid4().hex[:10]
    path = str(tmp_path / "l.gsd")
    try:
        ps = [subprocess.Popen([sys.executable, "-c", LOCAL_LOOKUP, name, str(r), path], stdout=subprocess.PIPE)
              for r in (0, 1)]
        outs = []
        try:
            for p in ps:
                outs.append(p.communicate(timeout=120)[0].decode().strip().splitlines())
        finally:
            for p in ps:
                if p.poll() is None:
                    p.kill()
        for r, out in enumerate(outs):
            assert out[0].startswith("%d True False |" % r) and "LOCAL lookup" in out[0] and "pending" in out[0], outs
            assert out[1] == "%d after flush True" % r, outs
    finally:
        try:
            os.unlink("/dev/shm/" + name)
        except OSError:
            pass
