"""PGSD file layer API, MI355X-native.

Mirror of the reference's Cython module ``pgsd.fl`` (/root/reference/pgsd/pgsd/fl.pyx): same
``open`` / ``PGSDFile`` surface, argument meaning, return shapes and exceptions, bound with
ctypes to the C ABI of ``libpgsd_amd.so`` (include/pgsd.h).  On top of the reference's
host-array ``write_chunk`` the same method accepts **device-resident** data (a torch tensor
on the GPU or a :class:`DeviceField`), which is packed by the HIP kernels and streamed to
the file without a host copy in Python; :meth:`PGSDFile.write_chunks` packs several
per-particle chunks with one fused launch.
"""
import ctypes
import errno as _errno
import logging
import os
from pickle import PickleError

import numpy

from . import _lib
from ._lib import lib

logger = logging.getLogger('pgsd.fl')

_NP_TO_PGSD = {
    numpy.dtype('uint8'): _lib.TYPE_UINT8, numpy.dtype('uint16'): _lib.TYPE_UINT16,
    numpy.dtype('uint32'): _lib.TYPE_UINT32, numpy.dtype('uint64'): _lib.TYPE_UINT64,
    numpy.dtype('int8'): _lib.TYPE_INT8, numpy.dtype('int16'): _lib.TYPE_INT16,
    numpy.dtype('int32'): _lib.TYPE_INT32, numpy.dtype('int64'): _lib.TYPE_INT64,
    numpy.dtype('float32'): _lib.TYPE_FLOAT, numpy.dtype('float64'): _lib.TYPE_DOUBLE,
}
_PGSD_TO_NP = {v: k for k, v in _NP_TO_PGSD.items()}
_TORCH_HAS_GPU = None       # torch.cuda.is_available(), asked once


def _pgsd_type(dtype, name=''):
    """numpy dtype / torch dtype / name -> pgsd type id (ValueError like fl.pyx:633)."""
    try:
        if not isinstance(dtype, numpy.dtype):
            s = str(dtype)
            if s.startswith('torch.'):
                s = s[6:]
            dtype = numpy.dtype(s)
        return _NP_TO_PGSD[dtype]
    except (KeyError, TypeError):
        raise ValueError("invalid type for chunk: " + name)


def _raise_on_error(retval, extra):
    """Error code -> exception, the mapping of fl.pyx:35-61 plus the device/comm codes."""
    if retval == 0:
        return
    if retval == _lib.ERROR_IO:
        err = ctypes.get_errno() or _errno.EIO
        raise IOError(err, os.strerror(err), extra)
    elif retval == _lib.ERROR_NOT_A_PGSD_FILE:
        raise RuntimeError("Not a PGSD file: " + extra)
    elif retval == _lib.ERROR_INVALID_PGSD_FILE_VERSION:
        raise RuntimeError("Unsupported PGSD file version: " + extra)
    elif retval == _lib.ERROR_FILE_CORRUPT:
        raise RuntimeError("Corrupt PGSD file: " + extra)
    elif retval == _lib.ERROR_MEMORY_ALLOCATION_FAILED:
        raise MemoryError("Memory allocation failed: " + extra)
    elif retval == _lib.ERROR_NAMELIST_FULL:
        raise RuntimeError("PGSD namelist is full: " + extra)
    elif retval == _lib.ERROR_FILE_MUST_BE_WRITABLE:
        raise RuntimeError("File must be writable: " + extra)
    elif retval == _lib.ERROR_FILE_MUST_BE_READABLE:
        raise RuntimeError("File must be readable: " + extra)
    elif retval == _lib.ERROR_INVALID_ARGUMENT:
        raise RuntimeError("Invalid pgsd argument: " + extra)
    elif retval in (_lib.ERROR_DEVICE, _lib.ERROR_NO_DEVICE, _lib.ERROR_COMM):
        raise RuntimeError("PGSD device/communicator error (%d): %s: %s"
                           % (retval, _lib.last_error(), extra))
    else:
        raise RuntimeError("Unknown error: " + extra)


class DeviceField(object):
    """Describe rows that live in GPU memory: ``chunk[i, c] = src[order[i]][col0 + c]``.

    Args:
        ptr (int): device address of the source array's first row.
        dtype: element type of the source array (numpy dtype or name).
        N (int): number of rows to write.
        M (int): number of columns of the chunk.
        stride (int): elements between consecutive source rows (4 for a HOOMD ``Scalar4``).
        col0 (int): first source column.
        out_dtype: element type of the chunk (default: ``dtype``); ``float64`` sources can be
            written as ``float32`` chunks.
        order (int): device address of an optional ``uint32[N]`` gather index (e.g. HOOMD's
            reverse-tag array) or ``None``.
        bitcast (bool): reinterpret the low bytes instead of converting the value (HOOMD
            keeps the type id in ``position.w`` as ``__int_as_scalar``).
        keepalive: any object that must stay alive until the frame is written.
    """

    def __init__(self, ptr, dtype, N, M, stride=None, col0=0, out_dtype=None, order=None,
                 bitcast=False, keepalive=None):
        self.ptr = int(ptr)
        self.dtype = numpy.dtype(dtype)
        self.N = int(N)
        self.M = int(M)
        self.stride = int(stride if stride is not None else M)
        self.col0 = int(col0)
        self.out_dtype = numpy.dtype(out_dtype) if out_dtype is not None else self.dtype
        self.order = int(order) if order else None
        self.bitcast = bool(bitcast)
        self.keepalive = keepalive

    @classmethod
    def from_tensor(cls, t, out_dtype=None, order=None, bitcast=False, columns=None):
        """Build from a torch GPU tensor of shape (N,), (N, M) or a column slice of (N, S)."""
        if columns is None and order is None and t.dim() == 2 and t.is_contiguous():
            # the common case (a whole row-major array), without the general path's dozen tensor queries
            N, M = t.shape
            return cls(t.data_ptr(), str(t.dtype)[6:], N, M, stride=max(M, 1), col0=0, out_dtype=out_dtype,
                       order=None, bitcast=bitcast, keepalive=[t])
        if t.dim() == 1:
            t2 = t.unsqueeze(1)
        elif t.dim() == 2:
            t2 = t
        else:
            raise ValueError("PGSD can only write 1 or 2 dimensional arrays")
        N, M = int(t2.shape[0]), int(t2.shape[1])
        if columns is not None:
            c0, c1 = columns
            t2 = t2[:, c0:c1]
            M = c1 - c0
        if M > 1 and t2.stride(1) != 1:
            t2 = t2.contiguous()
        stride = int(t2.stride(0)) if N > 1 else max(M, 1)
        itemsize = t2.element_size()
        col0 = 0
        ptr = t2.data_ptr()
        if stride < M:
            t2 = t2.contiguous()
            stride, ptr = M, t2.data_ptr()
        elif stride > M:
            # column slice of a wider row-major array: address whole rows so that the kernel
            # streams aligned, contiguous source tiles
            c = t2.storage_offset() % stride
            if c + M <= stride and t2.storage_offset() >= c:
                col0 = c
                ptr -= c * itemsize
        keep = [t2]
        order_ptr = None
        if order is not None:
            if str(order.dtype) not in ('torch.int32', 'torch.uint32'):
                raise ValueError("order must be a 32-bit integer tensor")
            order = order.contiguous()
            order_ptr = order.data_ptr()
            keep.append(order)
            N = int(order.shape[0])
        s = str(t2.dtype)[6:]
        return cls(ptr, s, N, M, stride=stride, col0=col0,
                   out_dtype=out_dtype, order=order_ptr, bitcast=bitcast, keepalive=keep)

    def _desc(self):
        d = _lib.FieldDesc()
        d.src = self.ptr
        d.order = self.order
        d.src_type = _pgsd_type(self.dtype)
        d.src_stride = self.stride
        d.src_col0 = self.col0
        d.bitcast = 1 if self.bitcast else 0
        return d


def _is_device_tensor(x):
    return hasattr(x, 'data_ptr') and getattr(x, 'is_cuda', False)


def select_rows(flags):
    """Stream compaction on the GPU for filtered snapshots.

    Args:
        flags: uint8 / bool torch GPU tensor of length N; non-zero = keep the particle.

    Returns:
        ``(index, count)``: ``index`` is an int32 GPU tensor whose first ``count`` entries are the
        kept rows in ascending order (usable as ``order=`` of :meth:`DeviceField.from_tensor`);
        ``count`` (int) is this rank's number of rows, i.e. what goes into the row-count allgather
        (``pgsd.dist.partition_rows``) that fixes every rank's file offsets.

    Wave-level ballot/popcount scans produce per-workgroup counts, one workgroup scans them
    into offsets, a scatter pass writes the indices (pgsd_select_rows in the C ABI).
    """
    import torch
    if not _is_device_tensor(flags):
        raise ValueError("flags must be a torch GPU tensor")
    f8 = flags.contiguous().view(torch.uint8) if flags.dtype in (torch.bool, torch.uint8, torch.int8) \
        else (flags != 0).to(torch.uint8)
    n = int(f8.numel())
    index = torch.empty((max(n, 1),), dtype=torch.int32, device=f8.device)
    count = torch.zeros((1,), dtype=torch.int64, device=f8.device)
    ws = torch.empty((max(int(lib.pgsd_select_workspace_bytes(n)), 16),), dtype=torch.uint8, device=f8.device)
    stream = torch.cuda.current_stream().cuda_stream
    retval = lib.pgsd_select_rows(f8.data_ptr(), n, index.data_ptr(), count.data_ptr(), ws.data_ptr(),
                                  ctypes.c_void_p(stream))
    _raise_on_error(retval, "select_rows")
    k = int(count.item())          # synchronises the stream
    return index[:k], k


def open(name, mode, application=None, schema=None, schema_version=None, comm=None):
    """Open a PGSD file and return a :py:class:`PGSDFile` (fl.pyx:149-228).

    Valid modes: ``'r'``, ``'r+'``, ``'w'``, ``'x'``, ``'a'``.  When creating a file
    (``'w'``, ``'x'``, ``'a'`` on a missing file) ``application``, ``schema`` and
    ``schema_version`` are required.

    ``comm`` (not in the reference, whose communicator is always ``MPI_COMM_WORLD``): a communicator made by
    ``pgsd.dist.create_shm`` / ``create_rccl`` that is NOT the process default -- several ranks in one process
    (one thread per GPU), each with a file object of its own.  It must outlive the file object.
    """
    return PGSDFile(str(name), mode, application, schema, schema_version, comm)


class PGSDFile(object):
    """PGSD file access interface (fl.pyx:231-1052)."""

    def __init__(self, name, mode, application, schema, schema_version, comm=None):
        self.__is_open = False
        self.__comm = comm
        exclusive_create = 0
        overwrite = 0
        self.__mode = mode
        # mode -> flags, fl.pyx:301-317
        if mode == 'w':
            c_flags = _lib.OPEN_READWRITE
            overwrite = 1
        elif mode == 'r':
            c_flags = _lib.OPEN_READONLY
        elif mode == 'r+':
            c_flags = _lib.OPEN_READWRITE
        elif mode == 'x':
            c_flags = _lib.OPEN_READWRITE
            overwrite = 1
            exclusive_create = 1
        elif mode == 'a':
            c_flags = _lib.OPEN_READWRITE
            if not os.path.exists(name):
                overwrite = 1
        else:
            raise ValueError("Invalid mode: " + mode)

        # One process per rank under torchrun: make sure the library knows about the other ranks
        # (a rank that believes it is alone would overwrite its neighbours' rows).
        if comm is None and lib.pgsd_comm_size() == 1 and _lib._torch is not None:
            tdist = _lib._torch.distributed
            if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
                from . import dist as _dist
                _dist.init_from_torch()

        self.__name = name
        self.__handle = _lib.Handle()
        self.__keepalive = []
        self.__explicit_stream = False
        self.__source_stream = -1       # what the pipeline was last told (-1: nothing yet)
        self.__deferred_rows = False
        self.__async_keep = []

        if overwrite:
            if application is None:
                raise ValueError("Provide application when creating a file")
            if schema is None:
                raise ValueError("Provide schema when creating a file")
            if schema_version is None:
                raise ValueError("Provide schema_version when creating a file")
            logger.info('overwriting file: ' + name + ' with mode: ' + mode
                        + ', application: ' + application + ', schema: ' + schema
                        + ', and schema_version: ' + str(schema_version))
            ctypes.set_errno(0)
            args = (ctypes.byref(self.__handle), name.encode('utf-8'),
                    application.encode('utf-8'), schema.encode('utf-8'),
                    lib.pgsd_make_version(schema_version[0], schema_version[1]), c_flags, exclusive_create)
            retval = lib.pgsd_create_and_open(*args) if comm is None \
                else lib.pgsd_create_and_open_on(ctypes.byref(comm), *args)
        else:
            logger.info('opening file: ' + name + ' with mode: ' + mode)
            ctypes.set_errno(0)
            args = (ctypes.byref(self.__handle), name.encode('utf-8'), c_flags)
            retval = lib.pgsd_open(*args) if comm is None else lib.pgsd_open_on(ctypes.byref(comm), *args)
        _raise_on_error(retval, name)
        self.__is_open = True

        # validate schema, fl.pyx:371-378
        if schema is not None:
            schema_truncated = schema
            if len(schema_truncated) > 64:
                schema_truncated = schema_truncated[0:63]
            if self.schema != schema_truncated:
                raise RuntimeError('file ' + name + ' has incorrect schema: ' + self.schema)

    # ------------------------------------------------------------------ helpers
    def _h(self):
        return ctypes.byref(self.__handle)

    def _check_open(self):
        if not self.__is_open:
            raise ValueError("File is not open")

    # ------------------------------------------------------------------ lifecycle
    def close(self, write_all=True):
        """Close the file (fl.pyx:382-419). May be called more than once."""
        if self.__is_open:
            logger.info('closing file: ' + self.__name)
            retval = lib.pgsd_close(self._h())
            self.__is_open = False
            self.__keepalive = []
            self.__async_keep = []
            _raise_on_error(retval, self.__name)

    def end_frame(self, write_all=True, wait=True):
        """Complete the current frame (fl.pyx:460-506).

        With ``wait=True`` (default, the reference's behaviour) the frame is in the file on
        return.  ``wait=False`` seals the frame but lets its device chunks finish in the
        background (``pgsd_end_frame_async``): call :meth:`wait_packed` before overwriting the
        source arrays and :meth:`frame_sync` (or any later synchronous call) before relying on the
        file contents.
        """
        self._check_open()
        logger.debug('end frame: ' + self.__name)
        if wait:
            retval = lib.pgsd_end_frame(self._h())
            # a synchronous seal drains the pipeline: nothing sealed earlier still reads its sources
            self.__keepalive = []
            self.__async_keep = []
        else:
            retval = lib.pgsd_end_frame_async(self._h())
            self.__async_keep.append(self.__keepalive)
            self.__keepalive = []
            # The pack kernels read the source arrays, the copies and writes read the staging arena: once
            # a frame's kernels are done its sources may go.  Only the newest frames can still be packing;
            # keep two, sync-free, so a long run of append(wait=False) does not pin every frame's tensors.
            if len(self.__async_keep) > 2:
                _raise_on_error(lib.pgsd_device_wait_packed(self._h()), self.__name)
                self.__async_keep = self.__async_keep[-1:]
        _raise_on_error(retval, self.__name)

    def frame_sync(self):
        """Wait until every asynchronously sealed frame of this rank is in the file."""
        self._check_open()
        retval = lib.pgsd_frame_sync(self._h())
        self.__async_keep = []
        _raise_on_error(retval, self.__name)

    def flush(self, write_all=True):
        """Flush all buffered frames to the file (fl.pyx:508-524)."""
        self._check_open()
        logger.debug('flush: ' + self.__name)
        retval = lib.pgsd_flush(self._h())
        self.__async_keep = []
        _raise_on_error(retval, self.__name)

    @property
    def frame_exchange(self):
        """bool: batch the exchange between the ranks per frame (``pgsd_set_frame_exchange``).

        Off (default): every chunk write exchanges the ranks' sizes at once, like the reference's per-chunk
        collectives.  On: replicated small chunks and all device chunks are queued and ONE allgather at
        :meth:`end_frame` carries their sizes and the ranks' status -- a frame of small chunks, fused device
        chunks (``offset='auto'``) and ``end_frame`` costs one collective.  The file is byte-identical
        either way."""
        self._check_open()
        return bool(lib.pgsd_get_frame_exchange(self._h()))

    @frame_exchange.setter
    def frame_exchange(self, on):
        self._check_open()
        _raise_on_error(lib.pgsd_set_frame_exchange(self._h(), 1 if on else 0), self.__name)

    @property
    def deferred_rows(self):
        """bool: with :attr:`frame_exchange` on, host arrays written with ``write_all=True`` wait for the frame's
        exchange like every other chunk instead of forcing one at once (``pgsd_set_deferred_rows``): the file object
        keeps the arrays alive until then, and the CALLER must not change them before :meth:`end_frame` /
        :meth:`flush` / :meth:`exchange_now` (the rule device tensors follow anyway).  A frame then costs one
        exchange whatever it holds."""
        return self.__deferred_rows

    @deferred_rows.setter
    def deferred_rows(self, on):
        self._check_open()
        _raise_on_error(lib.pgsd_set_deferred_rows(self._h(), 1 if on else 0), self.__name)
        self.__deferred_rows = bool(on)

    def exchange_now(self):
        """Perform the pending frame exchange now (collective; nothing is flushed)."""
        self._check_open()
        _raise_on_error(lib.pgsd_frame_exchange(self._h()), self.__name)

    @property
    def collective_count(self):
        """int: allgathers / barriers this handle has issued on its communicator."""
        self._check_open()
        return int(lib.pgsd_get_collective_count(self._h()))

    def exchange_stats(self, reset=False):
        """dict ``count``, ``total_us``, ``max_us``, ``min_us``: the allgathers this handle issued and their wall time on this
        rank (transport latency + the wait for the slowest rank)."""
        self._check_open()
        st = _lib.ExchangeStats()
        _raise_on_error(lib.pgsd_get_exchange_stats(self._h(), ctypes.byref(st), 1 if reset else 0), self.__name)
        return {"count": int(st.count), "total_us": float(st.total_us), "max_us": float(st.max_us),
                "min_us": float(st.min_us)}

    # ------------------------------------------------------------------ writing
    def write_chunk(self, name, data, offset=None, rank=0, write_all=True):
        """Write a data chunk to the current frame (fl.pyx:526-654).

        Args:
            name (str): chunk name.
            data: numpy array / array-like with <= 2 dimensions (host path, as in the
                reference), or a torch GPU tensor / :class:`DeviceField` (device path).
            offset: ``None`` or the integer array of every rank's row count; with ``rank`` it
                gives ``N_global = offset.sum()`` and this rank's first row
                ``offset[:rank].sum()`` (fl.pyx:594-598).  ``'auto'``: rows partitioned in rank order,
                counts taken from the library's own size exchange.
            rank (int): this rank.
            write_all (bool): ``True``: every rank writes its rows of a per-particle chunk;
                ``False``: replicated small chunk.
        """
        self._check_open()
        if isinstance(data, DeviceField) or _is_device_tensor(data):
            return self._write_chunk_device(name, data, offset, rank, write_all)

        data_array = numpy.ascontiguousarray(data)
        if data_array is not data:
            logger.warning('implicit data copy when writing chunk: ' + name)
        data_array = data_array.view()
        if len(data_array.shape) > 2:
            raise ValueError("PGSD can only write 1 or 2 dimensional arrays: " + name)
        if len(data_array.shape) == 1:
            data_array = data_array.reshape([data_array.shape[0], 1])
        N = data_array.shape[0]
        M = data_array.shape[1]
        N_global, stride = self._partition_args(offset, rank, N, M)
        pgsd_type = _pgsd_type(data_array.dtype, name)
        ptr = data_array.ctypes.data if data_array.size else None
        if self.__deferred_rows and write_all:
            self.__keepalive.append(data_array)     # the rows are read at the frame's exchange, not now
        logger.debug('write chunk: ' + self.__name + ' - ' + name)
        ctypes.set_errno(0)
        retval = lib.pgsd_write_chunk(self._h(), name.encode('utf-8'), pgsd_type, N, M, N_global, M,
                                      stride, (N_global * M) % 2 ** 64, bool(write_all), 0, ptr)
        _raise_on_error(retval, self.__name)

    @staticmethod
    def _partition_args(offset, rank, N, M):
        """``offset`` of :meth:`write_chunk` -> (N_global, element offset of this rank), fl.pyx:594-598.
        ``'auto'``: the library derives both from its own size exchange (PGSD_PARTITION_AUTO)."""
        if isinstance(offset, str):
            if offset != 'auto':
                raise ValueError("offset must be None, 'auto' or the array of every rank's row count")
            return _lib.PARTITION_AUTO, 0
        if offset is None:
            return N, 0
        offset = numpy.asarray(offset)
        return int(offset.sum()), M * int(offset[0:rank].sum())

    def _sync_source_stream(self):
        """Tell the pipeline which stream produced the arrays: PyTorch's current stream.  (The raw-handle
        query and the remembered last value keep this at ~1 us per call: `torch.cuda.current_stream()` builds
        a Stream object, 10-15 us, once per device write of a small frame.)"""
        global _TORCH_HAS_GPU
        torch = _lib._torch
        if torch is None:
            return
        if _TORCH_HAS_GPU is None:
            _TORCH_HAS_GPU = bool(torch.cuda.is_available())
        if not _TORCH_HAS_GPU:
            return                      # the device call itself reports the missing GPU
        try:
            stream = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
        except AttributeError:  # pragma: no cover - other torch versions
            stream = torch.cuda.current_stream().cuda_stream
        if stream != self.__source_stream:
            _raise_on_error(lib.pgsd_device_set_source_stream(self._h(), ctypes.c_void_p(stream)), self.__name)
            self.__source_stream = stream

    def set_source_stream(self, stream):
        """Name the HIP stream (integer handle) on which the particle arrays are produced;
        device writes are ordered after the work already enqueued there."""
        self._check_open()
        _raise_on_error(lib.pgsd_device_set_source_stream(self._h(), ctypes.c_void_p(stream)), self.__name)
        self.__explicit_stream = True
        self.__source_stream = stream

    def _write_chunk_device(self, name, data, offset, rank, write_all):
        if not self.__explicit_stream:
            self._sync_source_stream()
        f = data if isinstance(data, DeviceField) else DeviceField.from_tensor(data)
        N, M = f.N, f.M
        N_global, stride = self._partition_args(offset, rank, N, M)
        desc = f._desc()
        self.__keepalive.append(f)
        ctypes.set_errno(0)
        retval = lib.pgsd_write_chunk_device(self._h(), name.encode('utf-8'), _pgsd_type(f.out_dtype, name),
                                             N, M, N_global, M, stride, (N_global * M) % 2 ** 64, bool(write_all), 0,
                                             ctypes.byref(desc))
        _raise_on_error(retval, self.__name)

    def write_chunks(self, fields, offset=None, rank=0):
        """Write several per-particle chunks of the same N with ONE fused pack launch.

        Args:
            fields: list of ``(name, data)`` with ``data`` a torch GPU tensor or a
                :class:`DeviceField`; all must have the same number of rows.
            offset, rank: as in :meth:`write_chunk` (``write_all`` is implied); ``offset='auto'`` lets the
                library derive the partition from its own size exchange, so no row-count allgather of the
                caller is needed.
        """
        self._check_open()
        if not self.__explicit_stream:
            self._sync_source_stream()
        specs = []
        for name, data in fields:
            f = data if isinstance(data, DeviceField) else DeviceField.from_tensor(data)
            specs.append((name, f))
        if not specs:
            return
        N = specs[0][1].N
        if any(f.N != N for _, f in specs):
            raise ValueError("all fields of a fused write must have the same number of rows")
        N_global, row0 = self._partition_args(offset, rank, N, 1)
        reqs = (_lib.ChunkReq * len(specs))()
        names = []
        for i, (name, f) in enumerate(specs):
            names.append(name.encode('utf-8'))
            reqs[i].name = names[-1]
            reqs[i].type = _pgsd_type(f.out_dtype, name)
            reqs[i].M = f.M
            reqs[i].src = f._desc()
            self.__keepalive.append(f)
        ctypes.set_errno(0)
        retval = lib.pgsd_write_chunks_device(self._h(), len(specs), reqs, N, N_global, row0)
        _raise_on_error(retval, self.__name)

    def wait_packed(self):
        """Block until the pack kernels of the open frame are done (sources may be reused)."""
        self._check_open()
        _raise_on_error(lib.pgsd_device_wait_packed(self._h()), self.__name)

    def configure_device(self, device=-1, slab_bytes=0, n_slabs=0, n_writers=0, profile=False):
        """(Re)create the device pipeline of this file with explicit staging parameters."""
        self._check_open()
        cfg = _lib.DeviceConfig(device, slab_bytes, n_slabs, n_writers, 1 if profile else 0, 0)
        _raise_on_error(lib.pgsd_device_configure(self._h(), ctypes.byref(cfg)), self.__name)
        self.__source_stream = -1       # a new pipeline: it has to be told again

    def device_stats(self, reset=False):
        """dict of pipeline counters (pack launches/ms/bytes, D2H and write bytes/ms)."""
        self._check_open()
        st = _lib.DeviceStats()
        _raise_on_error(lib.pgsd_device_get_stats(self._h(), ctypes.byref(st), 1 if reset else 0),
                        self.__name)
        return {k: getattr(st, k) for k, _ in st._fields_}

    # ------------------------------------------------------------------ reading
    def chunk_exists(self, frame, name, write_all=False):
        """Test if a chunk exists (fl.pyx:656-715)."""
        self._check_open()
        logger.debug('chunk exists: ' + self.__name + ' - ' + name)
        e = lib.pgsd_find_chunk(self._h(), int(frame), name.encode('utf-8'))
        return bool(e)

    def read_chunk(self, frame, name, N=0, M=0, offset=0, r_all=False):
        """Read a data chunk and return it as a numpy array (fl.pyx:717-874).

        ``(N,)`` for Nx1 chunks, ``(N, M)`` otherwise.  With ``r_all=True`` only ``N`` rows of
        ``M`` columns starting at row ``offset`` are read (every rank reads its partition).
        """
        self._check_open()
        e = lib.pgsd_find_chunk(self._h(), int(frame), name.encode('utf-8'))
        if not e:
            raise KeyError("frame " + str(frame) + " / chunk " + name + " not found in: " + self.__name)
        entry = e.contents
        eN, eM, etype = int(entry.N), int(entry.M), int(entry.type)
        if etype not in _PGSD_TO_NP:
            raise ValueError("invalid type for chunk: " + name)
        data_array = numpy.empty(dtype=_PGSD_TO_NP[etype], shape=[eN, eM])
        logger.debug('read chunk: ' + self.__name + ' - ' + str(frame) + ' - ' + name)
        if eN != 0 and eM != 0:
            retval = lib.pgsd_read_chunk(self._h(), data_array.ctypes.data, e, int(N), int(M), int(offset),
                                         bool(r_all))
            _raise_on_error(retval, self.__name)
        if eM == 1:
            return data_array.reshape([eN])
        return data_array

    def read_chunk_device(self, frame, name, out=None, N=None, offset=0, columns=None, order=None,
                          bitcast=False, wait=True, fill=None):
        """Read rows ``[offset, offset + N)`` of a chunk straight into GPU memory.

        The rows are ``pread`` into pinned slabs, copied to HBM and unpacked by a HIP kernel
        (device twin of :meth:`read_chunk` with ``r_all=True``; every rank reads its own
        partition).

        Args:
            frame (int), name (str): the chunk.
            out: destination torch GPU tensor of shape ``(N,)``, ``(N, M)`` or wider ``(N, S)``
                (e.g. a ``Scalar4`` array); ``None`` allocates a dense ``(N, M)`` tensor of the
                chunk's type on the current device.
            N (int): number of rows (default: all rows after ``offset``).
            offset (int): first row.
            columns (tuple): ``(c0, c1)`` columns of ``out`` that receive the chunk's M columns.
            order: optional int32 GPU tensor; row ``i`` goes to ``out[order[i]]``.
            bitcast (bool): reinterpret equal-sized elements (uint32 type id -> float ``w`` slot).
            wait (bool): block until the data is in ``out`` (else call :meth:`wait_read`).
            fill: value for the columns of ``out``'s rows that no chunk read before the same :meth:`wait_read`
                writes (``pgsd_field_dst.fill_rest``): velocity into a ``Scalar4`` array with ``fill=1.0`` gives
                ``(vx, vy, vz, 1.0)`` rows, stored whole.  ``None``: those columns keep what they hold.

        Returns:
            the destination tensor.
        """
        self._check_open()
        import torch
        e = lib.pgsd_find_chunk(self._h(), int(frame), name.encode('utf-8'))
        if not e:
            raise KeyError("frame " + str(frame) + " / chunk " + name + " not found in: " + self.__name)
        entry = e.contents
        eN, eM, etype = int(entry.N), int(entry.M), int(entry.type)
        if etype not in _PGSD_TO_NP:
            raise ValueError("invalid type for chunk: " + name)
        if N is None:
            N = eN - int(offset)
        if N < 0 or int(offset) + N > eN:
            raise ValueError("row range outside the chunk: " + name)
        np_dt = _PGSD_TO_NP[etype]
        if out is None:
            tdt = getattr(torch, np_dt.name)
            out = torch.empty((N, eM) if eM > 1 else (N,), dtype=tdt, device='cuda')
        if not _is_device_tensor(out):
            raise ValueError("out must be a torch GPU tensor")
        t2 = out.unsqueeze(1) if out.dim() == 1 else out
        if t2.dim() != 2 or (t2.shape[1] > 1 and t2.stride(1) != 1):
            raise ValueError("out must be 1-D or row-major 2-D")
        stride = int(t2.stride(0)) if t2.shape[0] > 1 else int(t2.shape[1])
        c0 = 0 if columns is None else int(columns[0])
        if columns is not None and int(columns[1]) - c0 != eM:
            raise ValueError("columns must span the chunk's %d columns" % eM)
        if c0 + eM > max(stride, int(t2.shape[1])):
            raise ValueError("chunk does not fit the destination rows")
        if order is None and int(t2.shape[0]) < N:
            raise ValueError("destination has fewer rows than requested")
        dst = _lib.FieldDst()
        dst.dst = t2.data_ptr()
        dst.order = order.data_ptr() if order is not None else None
        dst.dst_type = _pgsd_type(t2.dtype, name)
        dst.dst_stride = stride
        dst.dst_col0 = c0
        dst.bitcast = 1 if bitcast else 0
        if fill is not None:
            np_out = numpy.dtype(str(t2.dtype)[6:])
            dst.fill_rest = 1
            dst.fill_bits = int(numpy.array([fill], dtype=np_out).view(numpy.dtype('u%d' % np_out.itemsize))[0])
        self.__keepalive.append((out, order))
        if not self.__explicit_stream:
            self._sync_source_stream()      # the unpack is ordered behind this stream's use of `out`
        retval = lib.pgsd_read_chunk_device(self._h(), e, int(N), int(offset), ctypes.byref(dst))
        _raise_on_error(retval, self.__name)
        if wait:
            self.wait_read()
        return out

    def wait_read(self):
        """Block until every :meth:`read_chunk_device` issued so far has landed in GPU memory."""
        self._check_open()
        retval = lib.pgsd_device_wait_read(self._h())
        if self.__mode == 'r':
            self.__keepalive = []
        _raise_on_error(retval, self.__name)

    def find_matching_chunk_names(self, match, write_all=False):
        """All chunk names in the file that start with ``match`` (fl.pyx:876-945)."""
        self._check_open()
        retval = []
        c_match = match.encode('utf-8')
        found = lib.pgsd_find_matching_chunk_name(self._h(), c_match, None)
        while found:
            retval.append(ctypes.string_at(found).decode('utf-8'))
            found = lib.pgsd_find_matching_chunk_name(self._h(), c_match, found)
        return retval

    # ------------------------------------------------------------------ protocol
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()

    def __reduce__(self):
        """Allows filehandles to be pickled when in read only mode (fl.pyx:971-978)."""
        if self.__mode not in ['rb', 'r']:
            raise PickleError("Only read only GSDFiles can be pickled.")
        return (PGSDFile, (self.__name, self.__mode, self.application, self.schema, self.schema_version))

    def __del__(self):
        try:
            if self.__is_open:
                logger.info('closing file: ' + self.__name)
                lib.pgsd_close(self._h())
                self.__is_open = False
        except Exception:
            pass

    # ------------------------------------------------------------------ properties
    @property
    def name(self):
        return self.__name

    @property
    def mode(self):
        return self.__mode

    @property
    def pgsd_version(self):
        v = self.__handle.header.pgsd_version
        return (v >> 16, v & 0xffff)

    @property
    def schema_version(self):
        v = self.__handle.header.schema_version
        return (v >> 16, v & 0xffff)

    @property
    def schema(self):
        return self.__handle.header.schema.decode('utf-8')

    @property
    def application(self):
        return self.__handle.header.application.decode('utf-8')

    @property
    def rank(self):
        """int: this process's (or thread's) rank in the communicator the file was opened on."""
        return int(self.__handle.rank)

    @property
    def nprocs(self):
        """int: number of ranks of the communicator the file was opened on."""
        return int(self.__handle.nprocs)

    def allgather(self, send):
        """Allgather the bytes of the 1-D uint8 array ``send`` over the file's communicator
        (``pgsd_handle_allgather``); returns a ``(nprocs, len(send))`` uint8 array."""
        self._check_open()
        send = numpy.ascontiguousarray(send, dtype=numpy.uint8)
        out = numpy.zeros((self.nprocs, send.size), dtype=numpy.uint8)
        retval = lib.pgsd_handle_allgather(self._h(), send.ctypes.data_as(ctypes.c_void_p),
                                           out.ctypes.data_as(ctypes.c_void_p), send.size)
        _raise_on_error(retval, self.__name)
        return out

    @property
    def nframes(self):
        self._check_open()
        return lib.pgsd_get_nframes(self._h())

    @property
    def nnames(self):
        self._check_open()
        return lib.pgsd_get_nnames(self._h())

    @property
    def file_size(self):
        """Logical end of the file as the writer tracks it (bytes)."""
        return int(self.__handle.file_size)

    @property
    def maximum_write_buffer_size(self):
        self._check_open()
        return lib.pgsd_get_maximum_write_buffer_size(self._h())

    @maximum_write_buffer_size.setter
    def maximum_write_buffer_size(self, size):
        self._check_open()
        _raise_on_error(lib.pgsd_set_maximum_write_buffer_size(self._h(), size), self.__name)

    @property
    def index_entries_to_buffer(self):
        self._check_open()
        return lib.pgsd_get_index_entries_to_buffer(self._h())

    @index_entries_to_buffer.setter
    def index_entries_to_buffer(self, number):
        self._check_open()
        _raise_on_error(lib.pgsd_set_index_entries_to_buffer(self._h(), number), self.__name)
