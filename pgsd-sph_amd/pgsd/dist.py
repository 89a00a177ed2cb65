"""Install the communicator the file layer uses (the role of MPI_COMM_WORLD in the reference,
pgsd.c:106-202) from a ``torch.distributed`` process group.

* GPU ranks (backend ``nccl`` = RCCL on ROCm): the library's native RCCL back end is used.
  ``torch.distributed`` only distributes the 128-byte ncclUniqueId; afterwards every
  allgather of the write path (row counts -> file offsets) is one ``ncclAllGather`` over xGMI
  issued from C++ on a private HIP stream.
* CPU ranks (backend ``gloo``, used by the multi-process tests): a host-callback back end
  that forwards to ``torch.distributed.all_gather``.
* A single process needs nothing: the default communicator is "self".
"""
import ctypes
import os

import numpy

from . import _lib
from ._lib import lib

_keep = {}
_retired = []   # callbacks of earlier communicators: open files may still hold them (the C side keeps a
                # communicator alive until the last handle opened with it is closed)


def _retire():
    if "cb" in _keep:
        _retired.append(_keep.pop("cb"))
    _keep.clear()


def init_self():
    lib.pgsd_comm_finalize()        # back to the single-rank communicator (the state after library load)
    _retire()


def _install(comm, what):
    """Make a communicator created by pgsd_comm_create_* the process default (the default takes it over)."""
    rc = lib.pgsd_comm_set_default(ctypes.byref(comm))
    if rc != 0:
        lib.pgsd_comm_release(ctypes.byref(comm))
        raise RuntimeError("pgsd_comm_set_default(%s) failed: %s" % (what, _lib.last_error()))
    return rc


def init_shm(name, rank, size):
    """Ranks of one node meeting in /dev/shm/<name> (no torch needed), installed as the process default."""
    comm = _lib.Comm()
    rc = lib.pgsd_comm_create_shm(name.encode(), int(rank), int(size), ctypes.byref(comm))
    if rc != 0:
        raise RuntimeError("pgsd_comm_create_shm failed: " + _lib.last_error())
    _install(comm, "shm")


def init_rccl(unique_id, rank, size, device):
    """An RCCL communicator (pgsd_comm_create_rccl) installed as the process default; returns the C status (0 = ok,
    the reason in ``pgsd._lib.last_error()``)."""
    comm = _lib.Comm()
    rc = lib.pgsd_comm_create_rccl(bytes(unique_id), int(rank), int(size), int(device), ctypes.byref(comm))
    if rc != 0:
        return rc
    return lib.pgsd_comm_set_default(ctypes.byref(comm))


def init_from_torch(group=None, device=None, prefer_rccl=True, _single_rank_too=False):
    """Use an initialised ``torch.distributed`` group. Returns the back end name.

    ``_single_rank_too`` (tests): build the communicator even for a group of one rank, which
    otherwise gets the trivial "self" communicator."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        init_self()
        return "self"
    rank, size = dist.get_rank(group), dist.get_world_size(group)
    if size == 1 and not _single_rank_too:
        init_self()
        return "self"
    backend = dist.get_backend(group)
    # PGSD_RCCL_LIBRARY names a particular RCCL build -- or the tests' stand-in, which lets ranks that SHARE a GPU
    # (a gloo group on a one-GPU box) go through the native back end's code; the name then shows in the result
    chosen = os.environ.get("PGSD_RCCL_LIBRARY", "")
    if prefer_rccl and torch.cuda.is_available() and (backend == "nccl" or chosen):
        if device is None:
            device = torch.cuda.current_device()
        where = "cuda:%d" % device if backend == "nccl" else "cpu"    # where the group's own collectives run
        # First what every rank can know by itself: librccl loads, the device is there, rank 0 has an id.  The ranks
        # agree on that BEFORE anyone enters ncclCommInitRank, which waits for all of them.
        ready = lib.pgsd_comm_rccl_available(int(device))
        err = _lib.last_error() if ready != 0 else ""
        uid = torch.zeros(128, dtype=torch.uint8, device=where)
        if rank == 0 and ready == 0:
            buf = (ctypes.c_uint8 * 128)()
            ready = lib.pgsd_comm_rccl_unique_id(buf)
            if ready == 0:
                uid.copy_(torch.tensor(list(buf), dtype=torch.uint8))
            else:
                err = _lib.last_error()
        ok = torch.tensor([1 if ready == 0 else 0], dtype=torch.int32, device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise RuntimeError("the RCCL back end is not available on at least one rank: " + (err or "(another rank)"))
        dist.broadcast(uid, src=0, group=group)
        host = bytes(uid.cpu().tolist())
        rc = init_rccl(host, rank, size, int(device))
        err = _lib.last_error() if rc != 0 else ""
        if rc == 0:
            # one real exchange before the communicator is trusted with file offsets
            got = (ctypes.c_uint32 * size)()
            mine = ctypes.c_uint32(rank)
            rc = lib.pgsd_comm_allgather(ctypes.byref(mine), got, 4)
            if rc != 0 or list(got) != list(range(size)):
                err = "rccl self-check allgather returned %r: %s" % (list(got), _lib.last_error())
                rc = rc or -1
        # every rank takes the same decision, or the ranks would sit on different communicators
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            lib.pgsd_comm_finalize()
            raise RuntimeError("pgsd_comm_create_rccl failed on at least one rank: " + (err or "(another rank)"))
        return "rccl" if not chosen else "rccl[%s]" % os.path.basename(chosen)

    on_gpu = backend == "nccl"

    def _allgather(ctx, send, recv, nbytes):
        try:
            src = numpy.ctypeslib.as_array(ctypes.cast(send, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
            t = torch.from_numpy(src.copy())
            if on_gpu:
                t = t.cuda()
            out = [torch.empty_like(t) for _ in range(size)]
            dist.all_gather(out, t, group=group)
            dst = numpy.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_uint8)),
                                           shape=(nbytes * size,))
            for j, o in enumerate(out):
                dst[j * nbytes:(j + 1) * nbytes] = o.cpu().numpy()
            return 0
        except Exception:  # pragma: no cover
            return -1

    def _barrier(ctx):
        try:
            dist.barrier(group=group)
            return 0
        except Exception:  # pragma: no cover
            return -1

    cb_ag = _lib.ALLGATHER_FN(_allgather)
    cb_bar = _lib.BARRIER_FN(_barrier)
    comm = _lib.Comm()
    comm.ctx = None
    comm.rank = rank
    comm.size = size
    comm.allgather = cb_ag
    comm.barrier = cb_bar
    rc = lib.pgsd_comm_set_default(ctypes.byref(comm))
    if rc != 0:
        raise RuntimeError("pgsd_comm_set_default failed")
    _retire()
    _keep["cb"] = (cb_ag, cb_bar, comm)
    return "torch-" + backend


def create_shm(name, rank, size):
    """A shm communicator that is NOT installed as the process default (``pgsd_comm_create_shm``): pass it to
    ``pgsd.fl.open(..., comm=)``; several ranks may live in one process (one thread each).  Let it go with
    :func:`release` once every file opened on it is closed."""
    comm = _lib.Comm()
    rc = lib.pgsd_comm_create_shm(name.encode(), int(rank), int(size), ctypes.byref(comm))
    if rc != 0:
        raise RuntimeError("pgsd_comm_create_shm failed: " + _lib.last_error())
    return comm


def rccl_unique_id():
    """The 128-byte ncclUniqueId one rank creates and hands to the others (``pgsd_comm_rccl_unique_id``)."""
    buf = (ctypes.c_uint8 * 128)()
    if lib.pgsd_comm_rccl_unique_id(buf) != 0:
        raise RuntimeError("pgsd_comm_rccl_unique_id failed: " + _lib.last_error())
    return bytes(buf)


def create_rccl(unique_id, rank, size, device):
    """An RCCL communicator (one rank = one GPU; the ranks may be threads of one process) that is not the
    process default; see :func:`create_shm`."""
    comm = _lib.Comm()
    rc = lib.pgsd_comm_create_rccl(bytes(unique_id), int(rank), int(size), int(device), ctypes.byref(comm))
    if rc != 0:
        raise RuntimeError("pgsd_comm_create_rccl failed: " + _lib.last_error())
    return comm


def release(comm):
    """Destroy a communicator made by :func:`create_shm` / :func:`create_rccl` (collective for shm: a barrier)."""
    lib.pgsd_comm_release(ctypes.byref(comm))


def finalize():
    lib.pgsd_comm_finalize()
    _retire()


def partition_rows(n_local):
    """Allgather every rank's row count over the installed communicator.

    Returns ``(counts, row0, n_global)``; ``counts`` is the array ``write_chunk`` takes as
    ``offset`` (fl.pyx:594-598)."""
    size = lib.pgsd_comm_size()
    counts = (ctypes.c_uint64 * size)()
    row0 = ctypes.c_uint64(0)
    ng = ctypes.c_uint64(0)
    rc = lib.pgsd_partition_rows(int(n_local), ctypes.byref(row0), ctypes.byref(ng), counts)
    if rc != 0:
        raise RuntimeError("pgsd_partition_rows failed: " + _lib.last_error())
    return numpy.array(list(counts), dtype=numpy.uint64), int(row0.value), int(ng.value)
