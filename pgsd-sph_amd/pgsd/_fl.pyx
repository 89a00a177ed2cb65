# cython: language_level=3, binding=True
"""PGSD file layer API, MI355X-native (compiled part; import it as ``pgsd.fl``).

Mirror of the reference's Cython module ``pgsd.fl`` (/root/reference/pgsd/pgsd/fl.pyx): same
``open`` / ``PGSDFile`` surface, argument meaning, return shapes and exceptions, bound -- like the
reference -- with Cython to the C ABI of the library underneath (``libpgsd_amd.so``, declared in
``libpgsd_amd.pxd`` from ``include/pgsd.h``); the handle is embedded in the object (fl.pyx:284) and every
C call runs without the GIL (fl.pyx:352-860).  On top of the reference's host-array ``write_chunk`` the
same method accepts **device-resident** data (a torch tensor on the GPU or a :class:`DeviceField`), which
is packed by the HIP kernels and streamed to the file without a host copy in Python;
:meth:`PGSDFile.write_chunks` packs several per-particle chunks with one fused launch.
"""
import ctypes
import errno as _errno
import logging
import os
from pickle import PickleError

import numpy

from libc.errno cimport errno
from libc.stdint cimport uint8_t, uint32_t, uint64_t, uintptr_t
from libc.stdlib cimport calloc, free
from libc.string cimport memset
from cpython.buffer cimport PyObject_GetBuffer, PyBuffer_Release, PyBUF_ANY_CONTIGUOUS, PyBUF_SIMPLE

from . cimport libpgsd_amd as C
from . import _lib

logger = logging.getLogger('pgsd.fl')

if C.pgsd_abi_version() != C.PGSD_ABI_VERSION:
    raise ImportError("pgsd/_fl was built against ABI version %d of include/pgsd.h, libpgsd_amd.so has %d: rebuild "
                      "(`make -C pgsd-sph_amd/csrc`)" % (C.PGSD_ABI_VERSION, C.pgsd_abi_version()))

_NP_TO_PGSD = {
    numpy.dtype('uint8'): 1, numpy.dtype('uint16'): 2, numpy.dtype('uint32'): 3, numpy.dtype('uint64'): 4,
    numpy.dtype('int8'): 5, numpy.dtype('int16'): 6, numpy.dtype('int32'): 7, numpy.dtype('int64'): 8,
    numpy.dtype('float32'): 9, numpy.dtype('float64'): 10,
}
_PGSD_TO_NP = {v: k for k, v in _NP_TO_PGSD.items()}
_TORCH_DTYPE_TO_PGSD = {}   # torch.dtype -> pgsd type id, filled on first use
_TORCH_HAS_GPU = None       # torch.cuda.is_available(), asked once


def _pgsd_type(dtype, name=''):
    """numpy dtype / torch dtype / name -> pgsd type id (ValueError like fl.pyx:633)."""
    try:
        if not isinstance(dtype, numpy.dtype):
            t = _TORCH_DTYPE_TO_PGSD.get(dtype)
            if t is not None:
                return t
            s = str(dtype)
            is_torch = s.startswith('torch.')
            if is_torch:
                s = s[6:]
            t = _NP_TO_PGSD[numpy.dtype(s)]
            if is_torch:
                _TORCH_DTYPE_TO_PGSD[dtype] = t
            return t
        return _NP_TO_PGSD[dtype]
    except (KeyError, TypeError):
        raise ValueError("invalid type for chunk: " + name)


cdef _raise_on_error(int retval, extra, int err=0):
    """Error code -> exception, the mapping of fl.pyx:35-61 plus the device/comm codes."""
    if retval == 0:
        return
    if retval == C.PGSD_ERROR_IO:
        err = err or _errno.EIO
        raise IOError(err, os.strerror(err), extra)
    elif retval == C.PGSD_ERROR_NOT_A_PGSD_FILE:
        raise RuntimeError("Not a PGSD file: " + extra)
    elif retval == C.PGSD_ERROR_INVALID_PGSD_FILE_VERSION:
        raise RuntimeError("Unsupported PGSD file version: " + extra)
    elif retval == C.PGSD_ERROR_FILE_CORRUPT:
        raise RuntimeError("Corrupt PGSD file: " + extra)
    elif retval == C.PGSD_ERROR_MEMORY_ALLOCATION_FAILED:
        raise MemoryError("Memory allocation failed: " + extra)
    elif retval == C.PGSD_ERROR_NAMELIST_FULL:
        raise RuntimeError("PGSD namelist is full: " + extra)
    elif retval == C.PGSD_ERROR_FILE_MUST_BE_WRITABLE:
        raise RuntimeError("File must be writable: " + extra)
    elif retval == C.PGSD_ERROR_FILE_MUST_BE_READABLE:
        raise RuntimeError("File must be readable: " + extra)
    elif retval == C.PGSD_ERROR_INVALID_ARGUMENT:
        raise RuntimeError("Invalid pgsd argument: " + extra)
    elif retval in (C.PGSD_ERROR_DEVICE, C.PGSD_ERROR_NO_DEVICE, C.PGSD_ERROR_COMM):
        msg = C.pgsd_last_error_string()
        raise RuntimeError("PGSD device/communicator error (%d): %s: %s"
                           % (retval, msg.decode('utf-8', 'replace') if msg != NULL else '', extra))
    else:
        raise RuntimeError("Unknown error: " + extra)


def _device_index_of(field):
    """GPU index of the array behind a field (a DeviceField made from a tensor keeps it alive; or a tensor), or None."""
    keep = field.keepalive if isinstance(field, DeviceField) else field
    for t in (keep if isinstance(keep, (list, tuple)) else [keep]):
        index = getattr(getattr(t, 'device', None), 'index', None)
        if index is not None:
            return index
    return None


cdef class DeviceField:
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
    cdef public object dtype, out_dtype, keepalive
    cdef public object ptr, order
    cdef public Py_ssize_t N, M, stride, col0
    cdef public bint bitcast
    cdef uint32_t _src_type, _out_type

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
        self._src_type = _pgsd_type(self.dtype)
        self._out_type = 0      # resolved (with the chunk's name in the error) when written

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

    @classmethod
    def from_device_array(cls, a, out_dtype=None, bitcast=False, columns=None, order=None):
        """Build from any object that describes GPU memory through ``__cuda_array_interface__`` (version 2 or 3; the
        protocol ROCm builds of HOOMD's GPU snapshots, CuPy, Numba and PyTorch share): shape ``(N,)``, ``(N, M)`` or a
        column range ``columns=(c0, c1)`` of ``(N, S)`` -- a ``Scalar4`` array --, row-major, rows possibly strided.
        ``order``: an optional 32-bit integer array of the same kind (gather index).  The arrays must be complete when
        the chunk is written (the protocol's ``stream`` entry is not waited on: order the producer against
        :meth:`PGSDFile.set_source_stream`, or synchronise it)."""
        if _is_device_tensor(a):
            return cls.from_tensor(a, out_dtype=out_dtype, order=order, bitcast=bitcast, columns=columns)
        iface = getattr(a, '__cuda_array_interface__', None)
        if not isinstance(iface, dict) or 'data' not in iface or 'shape' not in iface or 'typestr' not in iface:
            raise ValueError("not a GPU array: no __cuda_array_interface__")
        dt = numpy.dtype(iface['typestr'])
        if dt.byteorder == '>':
            raise ValueError("big-endian device arrays are not supported")
        shape = tuple(int(x) for x in iface['shape'])
        if len(shape) == 1:
            shape = (shape[0], 1)
        elif len(shape) != 2:
            raise ValueError("PGSD can only write 1 or 2 dimensional arrays")
        N, S = shape
        strides = iface.get('strides')
        if strides is None:
            row = S
        else:
            strides = tuple(int(x) for x in strides) if len(iface['shape']) == 2 else (int(strides[0]), dt.itemsize)
            if (S > 1 and strides[1] != dt.itemsize) or strides[0] % dt.itemsize != 0:
                raise ValueError("device arrays must be row-major with element-aligned rows")
            row = strides[0] // dt.itemsize if N > 1 else S
        c0, c1 = (0, S) if columns is None else (int(columns[0]), int(columns[1]))
        if not (0 <= c0 < c1 <= S) or row < S:
            raise ValueError("columns outside the array's rows")
        ptr = int(iface['data'][0]) if N * S > 0 else 0
        keep = [a]
        order_ptr = None
        if order is not None:
            if _is_device_tensor(order):
                o_ptr, o_dt, o_n = order.data_ptr(), numpy.dtype(str(order.dtype)[6:]), int(order.shape[0])
                dense = order.dim() == 1 and order.is_contiguous()
            else:
                oi = order.__cuda_array_interface__
                o_ptr, o_dt, o_n = int(oi['data'][0]), numpy.dtype(oi['typestr']), int(oi['shape'][0])
                dense = len(oi['shape']) == 1 and oi.get('strides') in (None, (4,))
            if o_dt not in (numpy.dtype('<i4'), numpy.dtype('<u4')) or not dense:
                raise ValueError("order must be a contiguous 32-bit integer device array")
            order_ptr, N = o_ptr, o_n
            keep.append(order)
        return cls(ptr, dt, N, c1 - c0, stride=max(row, 1), col0=c0, out_dtype=out_dtype, order=order_ptr,
                   bitcast=bitcast, keepalive=keep)

    cdef void fill_desc(self, C.pgsd_field_desc* d):
        d.src = <const void*><uintptr_t>self.ptr
        d.order = <const uint32_t*><uintptr_t>(self.order if self.order is not None else 0)
        d.src_type = self._src_type
        d.src_stride = <uint32_t>self.stride
        d.src_col0 = <uint32_t>self.col0
        d.bitcast = 1 if self.bitcast else 0

    cdef uint32_t out_type(self, name) except 0:
        if self._out_type == 0:
            self._out_type = _pgsd_type(self.out_dtype, name)
        return self._out_type

    def _desc(self):
        """ctypes ``FieldDesc`` of this field (tests and tools that call the C ABI directly)."""
        d = _lib.FieldDesc()
        d.src = self.ptr
        d.order = self.order
        d.src_type = self._src_type
        d.src_stride = self.stride
        d.src_col0 = self.col0
        d.bitcast = 1 if self.bitcast else 0
        return d


def _is_device_tensor(x):
    return hasattr(x, 'data_ptr') and getattr(x, 'is_cuda', False)


def _is_device_array(x):
    """A torch GPU tensor, or any other object that describes GPU memory through ``__cuda_array_interface__``."""
    if _is_device_tensor(x):
        return True
    if hasattr(x, 'data_ptr'):
        return False                    # a torch tensor in host memory (its __cuda_array_interface__ raises)
    try:
        return isinstance(getattr(x, '__cuda_array_interface__', None), dict)
    except Exception:
        return False


cdef class DeviceBuffer:
    """Device memory owned by ``libpgsd_amd.so`` (``pgsd_device_alloc``), described through
    ``__cuda_array_interface__`` (version 3) so that PyTorch, CuPy, Numba or HOOMD take it without a copy
    (``torch.as_tensor(buf, device='cuda')``).  What :mod:`pgsd.fl` / :mod:`pgsd.hoomd` keep in HBM themselves lives
    here -- references of the GPU-side elision, rows of device reads when no tensor library is importable, index
    lists of :func:`select_rows` -- so the Python device path needs no tensor library: torch is an optional
    PRODUCER of source arrays, never a requirement.

    Args:
        shape: tuple of ints (or an int).
        dtype: numpy dtype of an element.
        device (int): HIP device ordinal (-1: the current one).
        pattern: optional host array / bytes repeated over the whole buffer (rows of a default value).
    """
    cdef public object dtype, shape, strides, base
    cdef public Py_ssize_t nbytes
    cdef public int device
    cdef uintptr_t _ptr
    cdef bint _owner

    def __init__(self, shape, dtype=numpy.uint8, device=-1, pattern=None):
        self.dtype = numpy.dtype(dtype)
        self.shape = (int(shape),) if isinstance(shape, (int, numpy.integer)) else tuple(int(x) for x in shape)
        self.strides = None
        self.base = None
        self.device = int(device)
        n = 1
        for x in self.shape:
            n *= x
        self.nbytes = n * self.dtype.itemsize
        cdef size_t c_bytes = self.nbytes, c_pat = 0
        cdef const void* c_ptr = NULL
        cdef int c_dev = self.device
        cdef void* got
        pat = None
        if pattern is not None:
            pat = numpy.ascontiguousarray(pattern).view(numpy.uint8).reshape(-1)
            if pat.size > 0:
                c_pat = pat.size
                c_ptr = <const void*><uintptr_t>pat.ctypes.data
        with nogil:
            got = C.pgsd_device_alloc(c_dev, c_bytes, c_ptr, c_pat)
        if got == NULL:
            msg = C.pgsd_last_error_string()
            raise RuntimeError("pgsd_device_alloc(%d bytes) failed: %s"
                               % (self.nbytes, msg.decode('utf-8', 'replace') if msg != NULL else ''))
        self._ptr = <uintptr_t>got
        self._owner = True

    def __dealloc__(self):
        if self._owner and self._ptr != 0:
            C.pgsd_device_free(self.device, <void*>self._ptr)
            self._ptr = 0

    @property
    def ptr(self):
        return int(self._ptr)

    def data_ptr(self):
        return int(self._ptr)

    def numel(self):
        n = 1
        for x in self.shape:
            n *= x
        return n

    def element_size(self):
        return self.dtype.itemsize

    def is_contiguous(self):
        return self.strides is None

    def __len__(self):
        return self.shape[0] if self.shape else 1

    @property
    def __cuda_array_interface__(self):
        return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (int(self._ptr), False), 'version': 3,
                'strides': self.strides}

    def view(self, dtype=None, shape=None, strides=None, offset_bytes=0):
        """Another description of (a part of) the same memory; the view keeps this buffer alive.  ``strides`` in
        bytes (0 repeats a row: the default rows of :meth:`pgsd.hoomd.HOOMDTrajectory.read_frame_device`)."""
        cdef DeviceBuffer v = DeviceBuffer.__new__(DeviceBuffer)
        v.dtype = numpy.dtype(dtype) if dtype is not None else self.dtype
        if shape is None:
            if self.nbytes % v.dtype.itemsize != 0:
                raise ValueError("the buffer is not a whole number of such elements")
            shape = ((self.nbytes - int(offset_bytes)) // v.dtype.itemsize,)
        v.shape = (int(shape),) if isinstance(shape, (int, numpy.integer)) else tuple(int(x) for x in shape)
        v.strides = tuple(int(x) for x in strides) if strides is not None else None
        n = 1
        for x in v.shape:
            n *= x
        # the bytes the view can reach must lie inside this buffer
        if v.strides is None:
            reach = n * v.dtype.itemsize
        else:
            reach = v.dtype.itemsize if n > 0 else 0
            for x, st in zip(v.shape, v.strides):
                reach += (x - 1) * st if x > 0 else 0
        if offset_bytes < 0 or int(offset_bytes) + reach > self.nbytes:
            raise ValueError("the view does not fit the buffer")
        v.nbytes = n * v.dtype.itemsize
        v.device = self.device
        v.base = self
        v._ptr = self._ptr + <uintptr_t>int(offset_bytes)
        v._owner = False
        return v

    def clone(self):
        """A buffer of its own with the same (dense) contents: one device-to-device copy."""
        if self.strides is not None:
            raise ValueError("only dense buffers are cloned")
        cdef DeviceBuffer c = DeviceBuffer(self.shape, self.dtype, self.device)
        cdef int rc, dev = self.device
        cdef size_t n = self.nbytes
        cdef uintptr_t d = c._ptr, s = self._ptr
        with nogil:
            rc = C.pgsd_device_copy(dev, <void*>d, <const void*>s, n)
        _raise_on_error(rc, "DeviceBuffer.clone")
        return c

    def to_host(self):
        """The contents as a numpy array (one device-to-host copy; dense buffers only)."""
        if self.strides is not None:
            raise ValueError("only dense buffers are copied to the host")
        out = numpy.empty(self.shape, dtype=self.dtype)
        cdef int rc, dev = self.device
        cdef size_t n = self.nbytes
        cdef uintptr_t d = out.ctypes.data, s = self._ptr
        if n:
            with nogil:
                rc = C.pgsd_device_copy(dev, <void*>d, <const void*>s, n)
            _raise_on_error(rc, "DeviceBuffer.to_host")
        return out


def _device_memory(x, what="array"):
    """(address, bytes) of a DENSE array in GPU memory: a :class:`DeviceBuffer`, a torch GPU tensor or any object with
    ``__cuda_array_interface__``."""
    if isinstance(x, DeviceBuffer):
        if not x.is_contiguous():
            raise ValueError("%s must be dense" % what)
        return x.ptr, int(x.nbytes)
    if _is_device_tensor(x):
        if not x.is_contiguous():
            raise ValueError("%s must be a contiguous GPU tensor" % what)
        return int(x.data_ptr()), int(x.numel() * x.element_size())
    iface = getattr(x, '__cuda_array_interface__', None) if not hasattr(x, 'data_ptr') else None
    if not isinstance(iface, dict):
        raise ValueError("%s must live in GPU memory (DeviceBuffer, torch GPU tensor or __cuda_array_interface__)" % what)
    dt = numpy.dtype(iface['typestr'])
    shape = tuple(int(v) for v in iface['shape'])
    n = 1
    for v in shape:
        n *= v
    strides = iface.get('strides')
    if strides is not None:
        expect, acc = [], dt.itemsize
        for v in reversed(shape):
            expect.append(acc)
            acc *= v
        if n > 0 and tuple(int(v) for v in strides) != tuple(reversed(expect)):
            raise ValueError("%s must be dense (C-contiguous)" % what)
    return (int(iface['data'][0]) if n > 0 else 0), n * dt.itemsize


def select_rows(flags):
    """Stream compaction on the GPU for filtered snapshots.

    Args:
        flags: uint8 / bool array of length N in GPU memory (torch GPU tensor, :class:`DeviceBuffer` or any
            ``__cuda_array_interface__`` object of one-byte elements); non-zero = keep the particle.

    Returns:
        ``(index, count)``: ``index`` holds the kept rows in ascending order as 32-bit integers (usable as
        ``order=`` of :meth:`DeviceField.from_tensor` / :meth:`DeviceField.from_device_array`) -- an int32 GPU tensor
        of ``count`` entries when ``flags`` is a torch tensor, a :class:`DeviceBuffer` view otherwise; ``count``
        (int) is this rank's number of rows, i.e. what goes into the row-count allgather
        (``pgsd.dist.partition_rows``) that fixes every rank's file offsets.

    Wave-level ballot/popcount scans produce per-workgroup counts, one workgroup scans them
    into offsets, a scatter pass writes the indices (pgsd_select_rows in the C ABI; the scratch space is the
    library's).
    """
    cdef uintptr_t stream = 0, p_flags, p_index
    cdef uint64_t n, k = 0
    cdef int retval
    if _is_device_tensor(flags):
        torch = _lib._torch
        f8 = flags.contiguous().view(torch.uint8) if flags.dtype in (torch.bool, torch.uint8, torch.int8) \
            else (flags != 0).to(torch.uint8)
        n = int(f8.numel())
        index = torch.empty((max(n, 1),), dtype=torch.int32, device=f8.device)
        stream = torch.cuda.current_stream(f8.device).cuda_stream
        p_flags, p_index = f8.data_ptr(), index.data_ptr()
        keep = (f8, index)
    else:
        ptr, nbytes = _device_memory(flags, "flags")
        iface = flags.__cuda_array_interface__
        if numpy.dtype(iface['typestr']).itemsize != 1:
            raise ValueError("flags must have one-byte elements")
        n = nbytes
        index = DeviceBuffer((max(n, 1),), numpy.int32, getattr(flags, 'device', -1) if isinstance(flags, DeviceBuffer) else -1)
        p_flags, p_index = ptr, index.ptr
        keep = (flags, index)
    with nogil:
        retval = C.pgsd_select_rows(<const uint8_t*>p_flags, n, <uint32_t*>p_index, &k, <void*>stream)
    _raise_on_error(retval, "select_rows")
    if isinstance(index, DeviceBuffer):
        return index.view(shape=(int(k),)), int(k)
    return index[:int(k)], int(k)


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


cdef class PGSDFile:
    """PGSD file access interface (fl.pyx:231-1052)."""
    cdef C.pgsd_handle _handle          # embedded, by value (fl.pyx:284)
    cdef bint _is_open
    cdef object _mode, _name, _comm
    cdef list _keepalive, _async_keep
    cdef bint _explicit_stream, _deferred_rows, _local_reads, _frame_exchange_on
    cdef object _source_stream

    def __init__(self, name, mode, application, schema, schema_version, comm=None):
        cdef C.pgsd_open_flag c_flags
        cdef int exclusive_create = 0
        cdef int overwrite = 0
        cdef int retval, err
        self._is_open = False
        self._comm = comm
        self._mode = mode
        # mode -> flags, fl.pyx:301-317
        if mode == 'w':
            c_flags = C.PGSD_OPEN_READWRITE
            overwrite = 1
        elif mode == 'r':
            c_flags = C.PGSD_OPEN_READONLY
        elif mode == 'r+':
            c_flags = C.PGSD_OPEN_READWRITE
        elif mode == 'x':
            c_flags = C.PGSD_OPEN_READWRITE
            overwrite = 1
            exclusive_create = 1
        elif mode == 'a':
            c_flags = C.PGSD_OPEN_READWRITE
            if not os.path.exists(name):
                overwrite = 1
        else:
            raise ValueError("Invalid mode: " + mode)

        # One process per rank under torchrun: make sure the library knows about the other ranks
        # (a rank that believes it is alone would overwrite its neighbours' rows).
        if comm is None and C.pgsd_comm_size() == 1 and _lib._torch is not None:
            tdist = _lib._torch.distributed
            if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
                from . import dist as _dist
                _dist.init_from_torch()

        self._name = name
        self._keepalive = []
        self._explicit_stream = False
        self._frame_exchange_on = False
        self._source_stream = -1       # what the pipeline was last told (-1: nothing yet)
        self._deferred_rows = False
        self._local_reads = False
        self._async_keep = []
        memset(&self._handle, 0, sizeof(self._handle))

        cdef const C.pgsd_comm* c_comm = NULL
        if comm is not None:
            c_comm = <const C.pgsd_comm*><uintptr_t>ctypes.addressof(comm)
        name_b = name.encode('utf-8')
        cdef const char* c_name = name_b
        cdef const char* c_app
        cdef const char* c_schema
        cdef uint32_t c_version
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
            app_b, schema_b = application.encode('utf-8'), schema.encode('utf-8')
            c_app, c_schema = app_b, schema_b
            c_version = C.pgsd_make_version(schema_version[0], schema_version[1])
            with nogil:
                if c_comm == NULL:
                    retval = C.pgsd_create_and_open(&self._handle, c_name, c_app, c_schema, c_version, c_flags,
                                                    exclusive_create)
                else:
                    retval = C.pgsd_create_and_open_on(c_comm, &self._handle, c_name, c_app, c_schema, c_version,
                                                       c_flags, exclusive_create)
                err = errno
        else:
            logger.info('opening file: ' + name + ' with mode: ' + mode)
            with nogil:
                if c_comm == NULL:
                    retval = C.pgsd_open(&self._handle, c_name, c_flags)
                else:
                    retval = C.pgsd_open_on(c_comm, &self._handle, c_name, c_flags)
                err = errno
        _raise_on_error(retval, name, err)
        self._is_open = True

        # validate schema, fl.pyx:371-378
        if schema is not None:
            schema_truncated = schema
            if len(schema_truncated) > 64:
                schema_truncated = schema_truncated[0:63]
            if self.schema != schema_truncated:
                raise RuntimeError('file ' + name + ' has incorrect schema: ' + self.schema)

    # ------------------------------------------------------------------ helpers
    def _h(self):
        """ctypes pointer to the embedded handle: for tests and tools that call the C ABI directly."""
        return ctypes.cast(<uintptr_t>&self._handle, _lib.HP)

    def _async_frames_kept(self):
        """Number of asynchronously sealed frames whose source arrays are still held (tests)."""
        return len(self._async_keep)

    cdef int _check_open(self) except -1:
        if not self._is_open:
            raise ValueError("File is not open")
        return 0

    # ------------------------------------------------------------------ lifecycle
    def close(self, write_all=True):
        """Close the file (fl.pyx:382-419). May be called more than once."""
        cdef int retval, err
        if self._is_open:
            logger.info('closing file: ' + self._name)
            with nogil:
                retval = C.pgsd_close(&self._handle)
                err = errno
            self._is_open = False
            self._keepalive = []
            self._async_keep = []
            _raise_on_error(retval, self._name, err)

    def end_frame(self, write_all=True, wait=True):
        """Complete the current frame (fl.pyx:460-506).

        With ``wait=True`` (default, the reference's behaviour) the frame is in the file on
        return.  ``wait=False`` seals the frame but lets its device chunks finish in the
        background (``pgsd_end_frame_async``): call :meth:`wait_packed` before overwriting the
        source arrays and :meth:`frame_sync` (or any later synchronous call) before relying on the
        file contents.
        """
        cdef int retval, err, prc = 0
        self._check_open()
        if wait:
            with nogil:
                retval = C.pgsd_end_frame(&self._handle)
                err = errno
            # a synchronous seal drains the pipeline: nothing sealed earlier still reads its sources
            self._keepalive = []
            self._async_keep = []
        else:
            with nogil:
                retval = C.pgsd_end_frame_async(&self._handle)
                err = errno
            self._async_keep.append(self._keepalive)
            self._keepalive = []
            # The pack kernels read the source arrays, the copies and writes read the staging arena: once
            # a frame's kernels are done its sources may go.  Only the newest frames can still be packing;
            # keep two, sync-free, so a long run of append(wait=False) does not pin every frame's tensors.
            if len(self._async_keep) > 2:
                with nogil:
                    prc = C.pgsd_device_wait_packed(&self._handle)
                _raise_on_error(prc, self._name)
                self._async_keep = self._async_keep[-1:]
        _raise_on_error(retval, self._name, err)

    def frame_sync(self):
        """Wait until every asynchronously sealed frame of this rank is in the file."""
        cdef int retval, err
        self._check_open()
        with nogil:
            retval = C.pgsd_frame_sync(&self._handle)
            err = errno
        self._async_keep = []
        _raise_on_error(retval, self._name, err)

    def flush(self, write_all=True):
        """Flush all buffered frames to the file (fl.pyx:508-524)."""
        cdef int retval, err
        self._check_open()
        with nogil:
            retval = C.pgsd_flush(&self._handle)
            err = errno
        self._async_keep = []
        _raise_on_error(retval, self._name, err)

    @property
    def frame_exchange(self):
        """bool: batch the exchange between the ranks per frame (``pgsd_set_frame_exchange``).

        Off (default): every chunk write exchanges the ranks' sizes at once, like the reference's per-chunk
        collectives.  On: replicated small chunks and all device chunks are queued and ONE allgather at
        :meth:`end_frame` carries their sizes and the ranks' status -- a frame of small chunks, fused device
        chunks (``offset='auto'``) and ``end_frame`` costs one collective.  The file is byte-identical
        either way."""
        self._check_open()
        return bool(self._frame_exchange_on)

    @frame_exchange.setter
    def frame_exchange(self, on):
        cdef int retval, err, flag = 1 if on else 0
        self._check_open()
        with nogil:
            retval = C.pgsd_set_frame_exchange(&self._handle, flag)
            err = errno
        _raise_on_error(retval, self._name, err)
        self._frame_exchange_on = flag != 0

    @property
    def deferred_rows(self):
        """bool: with :attr:`frame_exchange` on, host arrays written with ``write_all=True`` wait for the frame's
        exchange like every other chunk instead of forcing one at once (``pgsd_set_deferred_rows``): the file object
        keeps the arrays alive until then, and the CALLER must not change them before :meth:`end_frame` /
        :meth:`flush` / :meth:`exchange_now` (the rule device tensors follow anyway).  A frame then costs one
        exchange whatever it holds."""
        return bool(self._deferred_rows)

    @deferred_rows.setter
    def deferred_rows(self, on):
        cdef int retval, err, flag = 1 if on else 0
        self._check_open()
        with nogil:
            retval = C.pgsd_set_deferred_rows(&self._handle, flag)
            err = errno
        _raise_on_error(retval, self._name, err)
        self._deferred_rows = bool(on)

    @property
    def local_reads(self):
        """bool: reads on a writable file take no part in a collective flush (``pgsd_set_local_reads``): for rows the
        caller knows to be in the file -- this rank's own rows of a sealed frame, or a file opened after they were
        written.  Default False: a read flushes first, collectively, like the reference's (pgsd.c:2436-2537).

        With several ranks the LOOKUP in front of a read (:meth:`chunk_exists`, :meth:`read_chunk`,
        :meth:`find_matching_chunk_names`) is local too while this is on: it sees what the last collective flush
        committed.  A frame that held buffered small chunks only is not flushed by :meth:`end_frame`
        (pgsd.c:1941-1950), so its chunks are reported missing until the next :meth:`flush`; the library then leaves
        a note in ``pgsd_last_error_string()``.  On ONE rank the flush concerns nobody else and runs as ever."""
        return bool(self._local_reads)

    @local_reads.setter
    def local_reads(self, on):
        cdef int retval, err, flag = 1 if on else 0
        self._check_open()
        with nogil:
            retval = C.pgsd_set_local_reads(&self._handle, flag)
            err = errno
        _raise_on_error(retval, self._name, err)
        self._local_reads = bool(on)

    def set_partition(self, rows):
        """Declare every rank's row count (``pgsd_set_partition``): while declared, chunk writes exchange nothing --
        chunks written with ``offset='auto'`` are partitioned by ``rows`` (this rank must bring ``rows[rank]``),
        every other chunk must have the same size on every rank -- and a frame costs no collective at all.
        ``None`` clears the declaration.  Every rank must declare the same vector."""
        cdef int retval, err
        cdef uint32_t n = 0
        cdef const uint64_t[::1] view
        cdef const uint64_t* ptr = NULL
        self._check_open()
        if rows is not None:
            arr = numpy.ascontiguousarray(rows, dtype=numpy.uint64)
            if arr.ndim != 1 or arr.shape[0] != self._handle.nprocs:
                raise ValueError("the partition must have one row count per rank")
            view = arr
            ptr = &view[0]
            n = arr.shape[0]
        with nogil:
            retval = C.pgsd_set_partition(&self._handle, ptr, n)
            err = errno
        _raise_on_error(retval, self._name, err)

    def exchange_now(self):
        """Perform the pending frame exchange now (collective; nothing is flushed)."""
        cdef int retval, err
        self._check_open()
        with nogil:
            retval = C.pgsd_frame_exchange(&self._handle)
            err = errno
        _raise_on_error(retval, self._name, err)

    @property
    def collective_count(self):
        """int: allgathers / barriers this handle has issued on its communicator."""
        cdef C.pgsd_exchange_stats st
        self._check_open()
        _raise_on_error(C.pgsd_get_exchange_stats(&self._handle, &st, 0), self._name)
        return int(st.collectives)

    def exchange_stats(self, reset=False):
        """dict ``count``, ``total_us``, ``max_us``, ``min_us``: the allgathers this handle issued and their wall
        time on this rank (transport latency + the wait for the slowest rank)."""
        cdef C.pgsd_exchange_stats st
        self._check_open()
        _raise_on_error(C.pgsd_get_exchange_stats(&self._handle, &st, 1 if reset else 0), self._name)
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
        if isinstance(data, DeviceField) or _is_device_array(data):
            return self._write_chunk_device(name, data, offset, rank, write_all)

        data_array = numpy.ascontiguousarray(data)
        if data_array is not data:
            logger.warning('implicit data copy when writing chunk: ' + name)
        if data_array.ndim > 2:
            raise ValueError("PGSD can only write 1 or 2 dimensional arrays: " + name)
        cdef uint64_t N, N_global, stride
        cdef uint32_t M
        if data_array.ndim == 1:
            N, M = data_array.shape[0], 1
        elif data_array.ndim == 2:
            N, M = data_array.shape[0], data_array.shape[1]
        else:
            data_array = data_array.reshape([1, 1])
            N, M = 1, 1
        py_ng, py_stride = self._partition_args(offset, rank, N, M)
        N_global, stride = py_ng, py_stride
        cdef C.pgsd_type pgsd_type = <C.pgsd_type><int>_pgsd_type(data_array.dtype, name)
        cdef Py_buffer view
        cdef const void* ptr = NULL
        cdef bint have_view = False
        if data_array.size:
            PyObject_GetBuffer(data_array, &view, PyBUF_ANY_CONTIGUOUS)
            ptr = view.buf
            have_view = True
        if self._deferred_rows and write_all:
            self._keepalive.append(data_array)     # the rows are read at the frame's exchange, not now
        name_b = name.encode('utf-8')
        cdef const char* c_name = name_b
        cdef bint c_all = bool(write_all)
        cdef uint64_t global_size = N_global * M     # wraps like the C expression
        cdef int retval, err
        with nogil:
            retval = C.pgsd_write_chunk(&self._handle, c_name, pgsd_type, N, M, N_global, M, stride, global_size,
                                        c_all, 0, ptr)
            err = errno
        if have_view:
            PyBuffer_Release(&view)
        _raise_on_error(retval, self._name, err)

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

    cdef _sync_source_stream(self):
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
        cdef uintptr_t s
        if stream != self._source_stream:
            s = stream
            _raise_on_error(C.pgsd_device_set_source_stream(&self._handle, <void*>s), self._name)
            self._source_stream = stream

    def set_source_stream(self, stream):
        """Name the HIP stream (integer handle) on which the particle arrays are produced;
        device writes are ordered after the work already enqueued there."""
        self._check_open()
        cdef uintptr_t s = int(stream) if stream else 0
        _raise_on_error(C.pgsd_device_set_source_stream(&self._handle, <void*>s), self._name)
        self._explicit_stream = True
        self._source_stream = stream

    def _write_chunk_device(self, name, data, offset, rank, write_all):
        if not self._explicit_stream:
            self._sync_source_stream()
        cdef DeviceField f = data if isinstance(data, DeviceField) else DeviceField.from_device_array(data)
        cdef uint64_t N = f.N, N_global, stride
        cdef uint32_t M = f.M
        py_ng, py_stride = self._partition_args(offset, rank, N, M)
        N_global, stride = py_ng, py_stride
        cdef C.pgsd_field_desc desc
        f.fill_desc(&desc)
        cdef C.pgsd_type t = <C.pgsd_type><int>f.out_type(name)
        self._keepalive.append(f)
        name_b = name.encode('utf-8')
        cdef const char* c_name = name_b
        cdef bint c_all = bool(write_all)
        cdef uint64_t global_size = N_global * M
        cdef int retval, err
        with nogil:
            retval = C.pgsd_write_chunk_device(&self._handle, c_name, t, N, M, N_global, M, stride, global_size, c_all, 0,
                                               &desc)
            err = errno
        _raise_on_error(retval, self._name, err)

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
        if not self._explicit_stream:
            self._sync_source_stream()
        cdef Py_ssize_t n = len(fields), i
        if n == 0:
            return
        cdef C.pgsd_chunk_req* reqs = <C.pgsd_chunk_req*>calloc(n, sizeof(C.pgsd_chunk_req))
        if reqs == NULL:
            raise MemoryError()
        cdef DeviceField f
        cdef uint64_t N = 0, N_global, row0
        cdef int retval, err
        names = []
        try:
            for i in range(n):
                name, data = fields[i]
                f = data if isinstance(data, DeviceField) else DeviceField.from_device_array(data)
                if i == 0:
                    N = f.N
                elif <uint64_t>f.N != N:
                    raise ValueError("all fields of a fused write must have the same number of rows")
                name_b = name.encode('utf-8')
                names.append(name_b)
                reqs[i].name = name_b
                reqs[i].type = f.out_type(name)
                reqs[i].M = <uint32_t>f.M
                f.fill_desc(&reqs[i].src)
                self._keepalive.append(f)
            py_ng, py_row0 = self._partition_args(offset, rank, N, 1)
            N_global, row0 = py_ng, py_row0
            with nogil:
                retval = C.pgsd_write_chunks_device(&self._handle, <uint32_t>n, reqs, N, N_global, row0)
                err = errno
        finally:
            free(reqs)
        _raise_on_error(retval, self._name, err)

    def stage_chunks(self, fields):
        """Launch the fused pack of ``fields`` (as in :meth:`write_chunks`) NOW and return a ticket; the chunks get
        their place in the frame later, with :meth:`write_staged` (``pgsd_stage_chunks_device``).  The kernel -- and
        for small frames the PCIe crossing -- then runs under whatever the caller does in between."""
        self._check_open()
        if not self._explicit_stream:
            self._sync_source_stream()
        cdef Py_ssize_t n = len(fields), i
        if n == 0:
            raise ValueError("no fields to stage")
        cdef C.pgsd_chunk_req* reqs = <C.pgsd_chunk_req*>calloc(n, sizeof(C.pgsd_chunk_req))
        if reqs == NULL:
            raise MemoryError()
        cdef DeviceField f
        cdef uint64_t N = 0, ticket = 0
        cdef int retval, err
        names = []
        sizes = []
        try:
            for i in range(n):
                name, data = fields[i]
                f = data if isinstance(data, DeviceField) else DeviceField.from_device_array(data)
                if i == 0:
                    N = f.N
                elif <uint64_t>f.N != N:
                    raise ValueError("all fields of a fused write must have the same number of rows")
                name_b = name.encode('utf-8')
                names.append(name_b)
                reqs[i].name = name_b
                reqs[i].type = f.out_type(name)
                reqs[i].M = <uint32_t>f.M
                f.fill_desc(&reqs[i].src)
                self._keepalive.append(f)
                sizes.append(int(f.N) * int(f.M) * f.out_dtype.itemsize)
            with nogil:
                retval = C.pgsd_stage_chunks_device(&self._handle, <uint32_t>n, reqs, N, &ticket)
                err = errno
        finally:
            free(reqs)
        _raise_on_error(retval, self._name, err)
        # (ticket, rows, packed bytes of every chunk, the GPU the pipeline -- and with it the staging -- lives on: the
        # handle's own answer, whatever kind of array the sources are and whatever a tensor library's "current device" is)
        return (int(ticket), int(N), tuple(sizes), self.pipeline_device())

    def pipeline_device(self):
        """int: the HIP device this file's pipeline runs on (created on the device given to :meth:`configure_device`,
        else on the device that is current at the first device call)."""
        cdef int dev
        self._check_open()
        with nogil:
            dev = C.pgsd_device_of(&self._handle)
        if dev < 0:
            _raise_on_error(dev, self._name)
        return dev

    def write_staged(self, ticket, first, count, offset=None, rank=0):
        """Write chunks ``[first, first + count)`` of a :meth:`stage_chunks` ticket at this point of the frame
        (``offset`` / ``rank`` as in :meth:`write_chunks`)."""
        self._check_open()
        cdef uint64_t c_ticket = ticket[0], N = ticket[1], N_global, row0
        cdef uint32_t c_first = first, c_count = count
        py_ng, py_row0 = self._partition_args(offset, rank, N, 1)
        N_global, row0 = py_ng, py_row0
        cdef int retval, err
        with nogil:
            retval = C.pgsd_write_staged_chunks(&self._handle, c_ticket, c_first, c_count, N_global, row0)
            err = errno
        _raise_on_error(retval, self._name, err)

    def compare_staged(self, ticket, first, refs):
        """Do the packed rows of staged chunks ``first, first + 1, ...`` of a :meth:`stage_chunks` ticket equal
        ``refs[i]`` -- dense arrays in GPU memory (:class:`DeviceBuffer`, torch GPU tensors, ``__cuda_array_interface__``
        objects) holding the same rows of another frame as the chunk stores them (:meth:`read_chunk_device`,
        :meth:`copy_staged`), or ``None`` (not compared: ``False``)?  A reference SHORTER
        than the chunk repeats (rows of a default value: at least 4096 bytes, a multiple of 16 bytes and of whole
        rows).  One kernel behind the pack, one stream wait (``pgsd_compare_staged_chunks``).  Equality is
        ``numpy.array_equal``'s: integer chunks by their bytes, float chunks by value (a NaN equals nothing, +0.0 equals
        -0.0).  Returns a list of bool."""
        self._check_open()
        if not self._explicit_stream:
            self._sync_source_stream()      # the comparison is ordered behind this stream's writes to the references
        cdef Py_ssize_t n = len(refs), i
        if n == 0:
            return []
        cdef uint64_t c_ticket = ticket[0], rows = ticket[1]
        cdef uint32_t c_first = first, c_count = n
        cdef const void** ptrs = <const void**>calloc(n, sizeof(void*))
        cdef uint64_t* sizes = <uint64_t*>calloc(n, sizeof(uint64_t))
        cdef uint8_t* eq = <uint8_t*>calloc(n, 1)
        cdef uintptr_t p
        cdef int retval, err
        if ptrs == NULL or eq == NULL or sizes == NULL:
            free(ptrs)
            free(sizes)
            free(eq)
            raise MemoryError()
        try:
            for i in range(n):
                r = refs[i]
                if r is None:
                    continue
                addr, have = _device_memory(r, "a reference")
                if first + i >= len(ticket[2]):
                    raise ValueError("the ticket has no chunk %d" % (first + i))
                if have > ticket[2][first + i]:
                    # the kernel reads as many bytes of the reference as the packed chunk has: never fewer at hand
                    # (a shorter one repeats; the library checks its shape)
                    raise ValueError("reference %d holds more than the %d bytes of the packed chunk" % (i, ticket[2][first + i]))
                sizes[i] = have
                p = addr
                # an empty tensor has no address: any non-null one says "there is a reference" (no byte is read)
                ptrs[i] = <const void*>p if p != 0 else <const void*>ptrs
            with nogil:
                retval = C.pgsd_compare_staged_chunks(&self._handle, c_ticket, c_first, c_count, ptrs, sizes, eq)
                err = errno
            _raise_on_error(retval, self._name, err)
            return [bool(eq[i]) for i in range(n)]
        finally:
            free(ptrs)
            free(sizes)
            free(eq)

    def copy_staged(self, ticket, first, sizes):
        """Keep the packed bytes of staged chunks ``first, first + 1, ...`` of a ticket: returns one :class:`DeviceBuffer`
        of ``sizes[i]`` bytes per chunk (``None`` where ``sizes[i]`` is ``None``), filled asynchronously behind the
        pack (``pgsd_copy_staged_chunks``) -- references for :meth:`compare_staged` in later frames.  The buffers are
        the library's own, on the GPU the staging lives on (which need not be a tensor library's current device)."""
        self._check_open()
        cdef Py_ssize_t n = len(sizes), i
        if n == 0:
            return []
        for i in range(n):
            if sizes[i] is not None and (first + i >= len(ticket[2]) or int(sizes[i]) != ticket[2][first + i]):
                raise ValueError("chunk %d of the ticket has %d packed bytes" % (first + i, ticket[2][first + i]
                                                                                 if first + i < len(ticket[2]) else -1))
        device = ticket[3] if len(ticket) > 3 and ticket[3] is not None else self.pipeline_device()
        cdef uint64_t c_ticket = ticket[0]
        cdef uint32_t c_first = first, c_count = n
        cdef void** ptrs = <void**>calloc(n, sizeof(void*))
        cdef uintptr_t p
        cdef int retval, err
        if ptrs == NULL:
            raise MemoryError()
        out = []
        try:
            for i in range(n):
                if sizes[i] is None:
                    out.append(None)
                    continue
                t = DeviceBuffer((int(sizes[i]),), numpy.uint8, device)
                out.append(t)
                p = t.ptr
                ptrs[i] = <void*>p
            with nogil:
                retval = C.pgsd_copy_staged_chunks(&self._handle, c_ticket, c_first, c_count, ptrs)
                err = errno
            _raise_on_error(retval, self._name, err)
            return out
        finally:
            free(ptrs)

    def wait_packed(self):
        """Block until the pack kernels of the open frame are done (sources may be reused)."""
        cdef int retval
        self._check_open()
        with nogil:
            retval = C.pgsd_device_wait_packed(&self._handle)
        _raise_on_error(retval, self._name)

    def configure_device(self, device=-1, slab_bytes=0, n_slabs=0, n_writers=0, profile=False, prealloc_mib=0):
        """(Re)create the device pipeline of this file with explicit staging parameters.  ``prealloc_mib`` > 0: that
        much HBM staging and the whole ring of pinned slabs are allocated by this call instead of when a frame first
        needs them -- a simulation pays for its allocations before the run, not during a snapshot."""
        cdef C.pgsd_device_config cfg
        cdef int retval
        self._check_open()
        memset(&cfg, 0, sizeof(cfg))
        cfg.device = device
        cfg.slab_bytes = slab_bytes
        cfg.n_slabs = n_slabs
        cfg.n_writers = n_writers
        cfg.profile = 1 if profile else 0
        cfg.prealloc_mib = prealloc_mib
        with nogil:
            retval = C.pgsd_device_configure(&self._handle, &cfg)
        _raise_on_error(retval, self._name)
        self._source_stream = -1       # a new pipeline: it has to be told again

    def device_stats(self, reset=False):
        """dict of pipeline counters (pack launches/ms/bytes, D2H and write bytes/ms)."""
        cdef C.pgsd_device_stats st
        self._check_open()
        _raise_on_error(C.pgsd_device_get_stats(&self._handle, &st, 1 if reset else 0), self._name)
        return {"pack_launches": st.pack_launches, "pack_ms": st.pack_ms, "pack_rows": st.pack_rows,
                "pack_bytes_out": st.pack_bytes_out, "pack_bytes_in": st.pack_bytes_in, "d2h_bytes": st.d2h_bytes,
                "written_bytes": st.written_bytes, "d2h_ms": st.d2h_ms, "write_ms": st.write_ms}

    # ------------------------------------------------------------------ reading
    cdef const C.pgsd_index_entry* _find(self, frame, name) except? NULL:
        name_b = name.encode('utf-8')
        cdef const char* c_name = name_b
        cdef uint64_t c_frame = int(frame)
        cdef const C.pgsd_index_entry* e
        with nogil:
            e = C.pgsd_find_chunk(&self._handle, c_frame, c_name)
        return e

    def chunk_exists(self, frame, name, write_all=False):
        """Test if a chunk exists (fl.pyx:656-715)."""
        self._check_open()
        return self._find(frame, name) != NULL

    def read_chunk(self, frame, name, N=0, M=0, offset=0, r_all=False):
        """Read a data chunk and return it as a numpy array (fl.pyx:717-874).

        ``(N,)`` for Nx1 chunks, ``(N, M)`` otherwise.  With ``r_all=True`` only ``N`` rows of
        ``M`` columns starting at row ``offset`` are read (every rank reads its partition).
        """
        self._check_open()
        cdef const C.pgsd_index_entry* e = self._find(frame, name)
        if e == NULL:
            raise KeyError("frame " + str(frame) + " / chunk " + name + " not found in: " + self._name)
        cdef uint64_t eN = e.N
        cdef uint32_t eM = e.M
        cdef int etype = e.type
        if etype not in _PGSD_TO_NP:
            raise ValueError("invalid type for chunk: " + name)
        data_array = numpy.empty(dtype=_PGSD_TO_NP[etype], shape=[eN, eM])
        cdef Py_buffer view
        cdef uint64_t c_N = int(N)
        cdef uint32_t c_M = int(M), c_off = int(offset)
        cdef bint c_all = bool(r_all)
        cdef int retval, err
        if eN != 0 and eM != 0:
            PyObject_GetBuffer(data_array, &view, PyBUF_ANY_CONTIGUOUS)
            with nogil:
                retval = C.pgsd_read_chunk(&self._handle, view.buf, e, c_N, c_M, c_off, c_all)
                err = errno
            PyBuffer_Release(&view)
            _raise_on_error(retval, self._name, err)
        if eM == 1:
            return data_array.reshape([eN])
        return data_array

    def read_rows(self, frame, name, row0, n):
        """Rows ``[row0, row0 + n)`` of a chunk as an ``(n,)`` / ``(n, M)`` numpy array: `pgsd_read_chunk` with
        ``all == true`` (pgsd.c:2498-2534) into an array of THAT height -- :meth:`read_chunk` with ``r_all=True``
        allocates the chunk's full height like the reference's binding (fl.pyx:838-860), every rank the global array."""
        self._check_open()
        cdef const C.pgsd_index_entry* e = self._find(frame, name)
        if e == NULL:
            raise KeyError("frame " + str(frame) + " / chunk " + name + " not found in: " + self._name)
        cdef uint64_t eN = e.N
        cdef uint32_t eM = e.M
        cdef int etype = e.type
        if etype not in _PGSD_TO_NP:
            raise ValueError("invalid type for chunk: " + name)
        if int(row0) < 0 or int(n) < 0 or int(row0) + int(n) > eN or int(row0) >= (1 << 32):
            raise ValueError("row range outside the chunk: " + name)
        data_array = numpy.empty(dtype=_PGSD_TO_NP[etype], shape=[int(n), eM])
        cdef Py_buffer view
        cdef uint64_t c_N = int(n)
        cdef uint32_t c_off = int(row0)
        cdef int retval, err
        if c_N != 0 and eM != 0:
            PyObject_GetBuffer(data_array, &view, PyBUF_ANY_CONTIGUOUS)
            with nogil:
                retval = C.pgsd_read_chunk(&self._handle, view.buf, e, c_N, eM, c_off, True)
                err = errno
            PyBuffer_Release(&view)
            _raise_on_error(retval, self._name, err)
        if eM == 1:
            return data_array.reshape([int(n)])
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
        torch = _lib._torch
        cdef const C.pgsd_index_entry* e = self._find(frame, name)
        if e == NULL:
            raise KeyError("frame " + str(frame) + " / chunk " + name + " not found in: " + self._name)
        cdef C.pgsd_index_entry entry = e[0]      # a later flush may move the index storage
        eN, eM, etype = int(entry.N), int(entry.M), int(entry.type)
        if etype not in _PGSD_TO_NP:
            raise ValueError("invalid type for chunk: " + name)
        if N is None:
            N = eN - int(offset)
        if N < 0 or int(offset) + N > eN:
            raise ValueError("row range outside the chunk: " + name)
        np_dt = _PGSD_TO_NP[etype]
        if out is None:
            # on the GPU the pipeline runs on (not a tensor library's "current device"); a torch tensor where torch
            # is importable, the library's own memory otherwise
            device = self.pipeline_device()
            if torch is not None:
                out = torch.empty((N, eM) if eM > 1 else (N,), dtype=getattr(torch, np_dt.name),
                                  device=torch.device('cuda', device))
            else:
                out = DeviceBuffer((N, eM) if eM > 1 else (N,), np_dt, device)
        cdef uintptr_t p_dst, p_order = 0
        if _is_device_tensor(out):
            t2 = out.unsqueeze(1) if out.dim() == 1 else out
            if t2.dim() != 2 or (t2.shape[1] > 1 and t2.stride(1) != 1):
                raise ValueError("out must be 1-D or row-major 2-D")
            rows, width = int(t2.shape[0]), int(t2.shape[1])
            stride = int(t2.stride(0)) if rows > 1 else width
            out_dtype = t2.dtype
            p_dst = t2.data_ptr()
        else:
            iface = getattr(out, '__cuda_array_interface__', None) if not hasattr(out, 'data_ptr') or isinstance(out, DeviceBuffer) else None
            if not isinstance(iface, dict):
                raise ValueError("out must live in GPU memory (torch GPU tensor, DeviceBuffer or __cuda_array_interface__)")
            out_dtype = numpy.dtype(iface['typestr'])
            shp = tuple(int(v) for v in iface['shape'])
            if len(shp) == 1:
                shp = (shp[0], 1)
            if len(shp) != 2:
                raise ValueError("out must be 1-D or row-major 2-D")
            rows, width = shp
            st = iface.get('strides')
            if st is not None and len(iface['shape']) == 2:
                if (width > 1 and int(st[1]) != out_dtype.itemsize) or int(st[0]) % out_dtype.itemsize != 0:
                    raise ValueError("out must be 1-D or row-major 2-D")
                stride = int(st[0]) // out_dtype.itemsize if rows > 1 else width
            elif st is not None:
                if int(st[0]) % out_dtype.itemsize != 0:
                    raise ValueError("out must be 1-D or row-major 2-D")
                stride = int(st[0]) // out_dtype.itemsize if rows > 1 else 1
            else:
                stride = width
            p_dst = int(iface['data'][0]) if rows * width > 0 else 0
        c0 = 0 if columns is None else int(columns[0])
        if columns is not None and int(columns[1]) - c0 != eM:
            raise ValueError("columns must span the chunk's %d columns" % eM)
        if c0 + eM > max(stride, width):
            raise ValueError("chunk does not fit the destination rows")
        if order is None and rows < N:
            raise ValueError("destination has fewer rows than requested")
        if order is not None:
            p_order = _device_memory(order, "order")[0]
        cdef C.pgsd_field_dst dst
        memset(&dst, 0, sizeof(dst))
        dst.dst = <void*>p_dst
        dst.order = <const uint32_t*>p_order
        dst.dst_type = _pgsd_type(out_dtype, name)
        dst.dst_stride = stride
        dst.dst_col0 = c0
        dst.bitcast = 1 if bitcast else 0
        if fill is not None:
            np_out = numpy.dtype(str(out_dtype)[6:]) if str(out_dtype).startswith('torch.') else numpy.dtype(out_dtype)
            dst.fill_rest = 1
            dst.fill_bits = int(numpy.array([fill], dtype=np_out).view(numpy.dtype('u%d' % np_out.itemsize))[0])
        self._keepalive.append((out, order))
        if not self._explicit_stream:
            self._sync_source_stream()      # the unpack is ordered behind this stream's use of `out`
        cdef uint64_t c_N = N, c_off = int(offset)
        cdef int retval, err
        with nogil:
            retval = C.pgsd_read_chunk_device(&self._handle, &entry, c_N, c_off, &dst)
            err = errno
        _raise_on_error(retval, self._name, err)
        if wait:
            self.wait_read()
        return out

    def wait_read(self):
        """Block until every :meth:`read_chunk_device` issued so far has landed in GPU memory."""
        cdef int retval, err
        self._check_open()
        with nogil:
            retval = C.pgsd_device_wait_read(&self._handle)
            err = errno
        if self._mode == 'r':
            self._keepalive = []
        _raise_on_error(retval, self._name, err)

    def find_matching_chunk_names(self, match, write_all=False):
        """All chunk names in the file that start with ``match`` (fl.pyx:876-945)."""
        self._check_open()
        retval = []
        match_b = match.encode('utf-8')
        cdef const char* c_match = match_b
        cdef const char* found
        with nogil:
            found = C.pgsd_find_matching_chunk_name(&self._handle, c_match, NULL)
        while found != NULL:
            retval.append(found.decode('utf-8'))
            with nogil:
                found = C.pgsd_find_matching_chunk_name(&self._handle, c_match, found)
        return retval

    def allgather(self, send):
        """Allgather the bytes of the 1-D uint8 array ``send`` over the file's communicator
        (``pgsd_handle_allgather``); returns a ``(nprocs, len(send))`` uint8 array."""
        self._check_open()
        send = numpy.ascontiguousarray(send, dtype=numpy.uint8)
        out = numpy.zeros((self.nprocs, send.size), dtype=numpy.uint8)
        cdef const uint8_t[::1] s = send
        cdef uint8_t[:, ::1] o = out
        cdef size_t nbytes = send.size
        cdef int retval
        if nbytes == 0:
            return out
        with nogil:
            retval = C.pgsd_handle_allgather(&self._handle, &s[0], &o[0, 0], nbytes)
        _raise_on_error(retval, self._name)
        return out

    # ------------------------------------------------------------------ protocol
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()

    def __reduce__(self):
        """Allows filehandles to be pickled when in read only mode (fl.pyx:971-978)."""
        if self._mode not in ['rb', 'r']:
            raise PickleError("Only read only GSDFiles can be pickled.")
        return (PGSDFile, (self._name, self._mode, self.application, self.schema, self.schema_version))

    def __dealloc__(self):
        if self._is_open:
            with nogil:
                C.pgsd_close(&self._handle)
            self._is_open = False

    # ------------------------------------------------------------------ properties
    @property
    def name(self):
        return self._name

    @property
    def mode(self):
        return self._mode

    @property
    def pgsd_version(self):
        cdef uint32_t v = self._handle.header.pgsd_version
        return (v >> 16, v & 0xffff)

    @property
    def schema_version(self):
        cdef uint32_t v = self._handle.header.schema_version
        return (v >> 16, v & 0xffff)

    @property
    def schema(self):
        return self._handle.header.schema.decode('utf-8')

    @property
    def application(self):
        return self._handle.header.application.decode('utf-8')

    @property
    def rank(self):
        """int: this process's (or thread's) rank in the communicator the file was opened on."""
        return int(self._handle.rank)

    @property
    def nprocs(self):
        """int: number of ranks of the communicator the file was opened on."""
        return int(self._handle.nprocs)

    @property
    def nframes(self):
        self._check_open()
        return C.pgsd_get_nframes(&self._handle)

    @property
    def nnames(self):
        self._check_open()
        return C.pgsd_get_nnames(&self._handle)

    @property
    def file_size(self):
        """Logical end of the file as the writer tracks it (bytes)."""
        return int(self._handle.file_size)

    @property
    def maximum_write_buffer_size(self):
        self._check_open()
        return C.pgsd_get_maximum_write_buffer_size(&self._handle)

    @maximum_write_buffer_size.setter
    def maximum_write_buffer_size(self, size):
        cdef uint64_t c = size
        cdef int retval
        self._check_open()
        with nogil:
            retval = C.pgsd_set_maximum_write_buffer_size(&self._handle, c)
        _raise_on_error(retval, self._name)

    @property
    def index_entries_to_buffer(self):
        self._check_open()
        return C.pgsd_get_index_entries_to_buffer(&self._handle)

    @index_entries_to_buffer.setter
    def index_entries_to_buffer(self, number):
        cdef uint64_t c = number
        cdef int retval
        self._check_open()
        with nogil:
            retval = C.pgsd_set_index_entries_to_buffer(&self._handle, c)
        _raise_on_error(retval, self._name)
