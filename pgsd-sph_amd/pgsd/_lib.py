"""ctypes view of libpgsd_amd.so (the C ABI declared in include/pgsd.h).

This is the only place the shared library is loaded.  There is no Python or CPU stand-in
for it: if the library is missing the import fails, and the device entry points return
PGSD_ERROR_NO_DEVICE when no MI355X is visible.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpgsd_amd.so")

# PyTorch bundles its own libamdhip64.so.7; a process must only ever hold ONE HIP runtime,
# so when torch is installed it is imported first and the library binds to that runtime.
try:  # pragma: no cover - depends on the environment
    import torch as _torch  # noqa: F401
except Exception:  # torch is optional for host-only use
    _torch = None

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libpgsd_amd.so not found at %s: build it with `make -C pgsd-sph_amd/csrc` "
        "(or python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL, use_errno=True)

c_u8, c_u16, c_u32, c_u64 = ctypes.c_uint8, ctypes.c_uint16, ctypes.c_uint32, ctypes.c_uint64
c_i32, c_i64, c_vp, c_cp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_char_p

# enum pgsd_type / pgsd_open_flag / pgsd_error (include/pgsd.h)
TYPE_UINT8, TYPE_UINT16, TYPE_UINT32, TYPE_UINT64 = 1, 2, 3, 4
TYPE_INT8, TYPE_INT16, TYPE_INT32, TYPE_INT64 = 5, 6, 7, 8
TYPE_FLOAT, TYPE_DOUBLE = 9, 10
OPEN_READWRITE, OPEN_READONLY, OPEN_APPEND = 1, 2, 3
SUCCESS = 0
ERROR_IO = -1
ERROR_INVALID_ARGUMENT = -2
ERROR_NOT_A_PGSD_FILE = -3
ERROR_INVALID_PGSD_FILE_VERSION = -4
ERROR_FILE_CORRUPT = -5
ERROR_MEMORY_ALLOCATION_FAILED = -6
ERROR_NAMELIST_FULL = -7
ERROR_FILE_MUST_BE_WRITABLE = -8
ERROR_FILE_MUST_BE_READABLE = -9
ERROR_DEVICE = -20
ERROR_COMM = -21
ERROR_NO_DEVICE = -22
PARTITION_AUTO = 2 ** 64 - 1      # PGSD_PARTITION_AUTO


class Header(ctypes.Structure):
    _fields_ = [("magic", c_u64), ("index_location", c_u64), ("index_allocated_entries", c_u64),
                ("namelist_location", c_u64), ("namelist_allocated_entries", c_u64),
                ("schema_version", c_u32), ("pgsd_version", c_u32),
                ("application", ctypes.c_char * 64), ("schema", ctypes.c_char * 64),
                ("reserved", ctypes.c_char * 80)]


class IndexEntry(ctypes.Structure):
    _fields_ = [("frame", c_u64), ("N", c_u64), ("location", c_i64), ("M", c_u32), ("id", c_u16),
                ("type", c_u8), ("flags", c_u8)]


class IndexBuffer(ctypes.Structure):
    _fields_ = [("data", ctypes.POINTER(IndexEntry)), ("size", ctypes.c_size_t),
                ("reserved", ctypes.c_size_t)]


class ByteBuffer(ctypes.Structure):
    _fields_ = [("data", c_vp), ("size", ctypes.c_size_t), ("reserved", ctypes.c_size_t)]


class NameBuffer(ctypes.Structure):
    _fields_ = [("data", ByteBuffer), ("n_names", ctypes.c_size_t)]


class Handle(ctypes.Structure):
    _fields_ = [("fd", c_i32), ("header", Header), ("file_index", IndexBuffer),
                ("file_names", NameBuffer), ("cur_frame", c_u64), ("file_size", ctypes.c_longlong),
                ("open_flags", c_i32), ("pending_index_entries", c_u64),
                ("maximum_write_buffer_size", c_u64), ("index_entries_to_buffer", c_u64),
                ("rank", c_i32), ("nprocs", c_i32), ("impl", c_vp)]


class FieldDesc(ctypes.Structure):
    _fields_ = [("src", c_vp), ("order", c_vp), ("src_type", c_u32), ("src_stride", c_u32),
                ("src_col0", c_u32), ("bitcast", c_u32)]


class ChunkReq(ctypes.Structure):
    _fields_ = [("name", c_cp), ("type", c_u32), ("M", c_u32), ("src", FieldDesc)]


class PackJob(ctypes.Structure):
    _fields_ = [("dst", c_vp), ("dst_type", c_u32), ("M", c_u32), ("src", FieldDesc)]


class FieldDst(ctypes.Structure):
    _fields_ = [("dst", c_vp), ("order", c_vp), ("dst_type", c_u32), ("dst_stride", c_u32),
                ("dst_col0", c_u32), ("bitcast", c_u32), ("fill_rest", c_u32), ("reserved", c_u32),
                ("fill_bits", c_u64)]


class UnpackJob(ctypes.Structure):
    _fields_ = [("src", c_vp), ("src_type", c_u32), ("M", c_u32), ("dst", FieldDst)]


class DeviceConfig(ctypes.Structure):
    _fields_ = [("device", c_i32), ("slab_bytes", c_u64), ("n_slabs", c_u32), ("n_writers", c_u32),
                ("profile", c_u32), ("prealloc_mib", c_u32)]


class DeviceStats(ctypes.Structure):
    _fields_ = [("pack_launches", c_u64), ("pack_ms", ctypes.c_double), ("pack_rows", c_u64),
                ("pack_bytes_out", c_u64), ("pack_bytes_in", c_u64), ("d2h_bytes", c_u64),
                ("written_bytes", c_u64), ("d2h_ms", ctypes.c_double), ("write_ms", ctypes.c_double)]


class ExchangeStats(ctypes.Structure):
    _fields_ = [("collectives", c_u64), ("count", c_u64), ("total_us", ctypes.c_double), ("max_us", ctypes.c_double),
                ("min_us", ctypes.c_double)]


ALLGATHER_FN = ctypes.CFUNCTYPE(c_i32, c_vp, c_vp, c_vp, ctypes.c_size_t)
BARRIER_FN = ctypes.CFUNCTYPE(c_i32, c_vp)
DESTROY_FN = ctypes.CFUNCTYPE(None, c_vp)


class Comm(ctypes.Structure):
    _fields_ = [("ctx", c_vp), ("rank", c_i32), ("size", c_i32), ("allgather", ALLGATHER_FN),
                ("barrier", BARRIER_FN), ("destroy", DESTROY_FN)]


HP = ctypes.POINTER(Handle)


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


ABI_VERSION = 5     # PGSD_ABI_VERSION of the include/pgsd.h these signatures were written against

# every symbol include/pgsd.h declares (tests/test_abi.py checks the list against the header); the entry points of
# csrc/pgsd_private.h (bare kernels, queue plumbing, housekeeping: tests, tools, bench legs) follow at the end
_sig("pgsd_make_version", c_u32, ctypes.c_uint, ctypes.c_uint)
_sig("pgsd_create_and_open", c_i32, HP, c_cp, c_cp, c_cp, c_u32, c_i32, c_i32)
_sig("pgsd_open", c_i32, HP, c_cp, c_i32)
_sig("pgsd_close", c_i32, HP)
_sig("pgsd_end_frame", c_i32, HP)
_sig("pgsd_flush", c_i32, HP)
_sig("pgsd_end_frame_async", c_i32, HP)
_sig("pgsd_frame_sync", c_i32, HP)
_sig("pgsd_write_chunk", c_i32, HP, c_cp, c_i32, c_u64, c_u32, c_u64, c_u32, c_u64, c_u64,
     ctypes.c_bool, c_u8, c_vp)
_sig("pgsd_find_chunk", ctypes.POINTER(IndexEntry), HP, c_u64, c_cp)
_sig("pgsd_read_chunk", c_i32, HP, c_vp, ctypes.POINTER(IndexEntry), c_u64, c_u32, c_u32, ctypes.c_bool)
_sig("pgsd_get_nframes", c_u64, HP)
_sig("pgsd_get_nnames", c_u64, HP)
_sig("pgsd_sizeof_type", ctypes.c_size_t, c_i32)
_sig("pgsd_find_matching_chunk_name", c_vp, HP, c_cp, c_vp)
_sig("pgsd_get_maximum_write_buffer_size", c_u64, HP)
_sig("pgsd_set_maximum_write_buffer_size", c_i32, HP, c_u64)
_sig("pgsd_get_index_entries_to_buffer", c_u64, HP)
_sig("pgsd_set_index_entries_to_buffer", c_i32, HP, c_u64)
_sig("pgsd_bcast_index_entry", None, ctypes.POINTER(IndexEntry))
_sig("pgsd_last_error_string", c_cp)
_sig("pgsd_comm_set_default", c_i32, ctypes.POINTER(Comm))
_sig("pgsd_comm_init_from_env", c_i32)
_sig("pgsd_comm_rccl_unique_id", c_i32, c_vp)
_sig("pgsd_comm_rccl_available", c_i32, c_i32)
_sig("pgsd_device_release_parked", c_i32)
_sig("pgsd_comm_finalize", c_i32)
_sig("pgsd_comm_rank", c_i32)
_sig("pgsd_comm_size", c_i32)
_sig("pgsd_comm_allgather", c_i32, c_vp, c_vp, ctypes.c_size_t)
_sig("pgsd_comm_barrier", c_i32)
_sig("pgsd_comm_create_shm", c_i32, c_cp, c_i32, c_i32, ctypes.POINTER(Comm))
_sig("pgsd_comm_create_rccl", c_i32, c_vp, c_i32, c_i32, c_i32, ctypes.POINTER(Comm))
_sig("pgsd_comm_release", None, ctypes.POINTER(Comm))
_sig("pgsd_create_and_open_on", c_i32, ctypes.POINTER(Comm), HP, c_cp, c_cp, c_cp, c_u32, c_i32, c_i32)
_sig("pgsd_open_on", c_i32, ctypes.POINTER(Comm), HP, c_cp, c_i32)
_sig("pgsd_handle_allgather", c_i32, HP, c_vp, c_vp, ctypes.c_size_t)
_sig("pgsd_partition_rows", c_i32, c_u64, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64))
_sig("pgsd_write_chunk_device", c_i32, HP, c_cp, c_i32, c_u64, c_u32, c_u64, c_u32, c_u64, c_u64,
     ctypes.c_bool, c_u8, ctypes.POINTER(FieldDesc))
_sig("pgsd_write_chunks_device", c_i32, HP, c_u32, ctypes.POINTER(ChunkReq), c_u64, c_u64, c_u64)
_sig("pgsd_stage_chunks_device", c_i32, HP, c_u32, ctypes.POINTER(ChunkReq), c_u64, ctypes.POINTER(c_u64))
_sig("pgsd_write_staged_chunks", c_i32, HP, c_u64, c_u32, c_u32, c_u64, c_u64)
_sig("pgsd_compare_staged_chunks", c_i32, HP, c_u64, c_u32, c_u32, ctypes.POINTER(c_vp), ctypes.POINTER(c_u64),
     ctypes.POINTER(c_u8))
_sig("pgsd_copy_staged_chunks", c_i32, HP, c_u64, c_u32, c_u32, ctypes.POINTER(c_vp))
_sig("pgsd_device_wait_packed", c_i32, HP)
_sig("pgsd_device_set_source_stream", c_i32, HP, c_vp)
_sig("pgsd_device_configure", c_i32, HP, ctypes.POINTER(DeviceConfig))
_sig("pgsd_device_get_stats", c_i32, HP, ctypes.POINTER(DeviceStats), c_i32)
_sig("pgsd_pack_fields", c_i32, c_u32, ctypes.POINTER(PackJob), c_u64, c_vp, ctypes.POINTER(ctypes.c_float))
_sig("pgsd_set_frame_exchange", c_i32, HP, c_i32)
_sig("pgsd_frame_exchange", c_i32, HP)
_sig("pgsd_set_deferred_rows", c_i32, HP, c_i32)
_sig("pgsd_set_local_reads", c_i32, HP, c_i32)
_sig("pgsd_set_partition", c_i32, HP, ctypes.POINTER(c_u64), c_u32)
_sig("pgsd_get_exchange_stats", c_i32, HP, ctypes.POINTER(ExchangeStats), c_i32)
_sig("pgsd_unpack_fields", c_i32, c_u32, ctypes.POINTER(UnpackJob), c_u64, c_vp)
_sig("pgsd_read_chunk_device", c_i32, HP, ctypes.POINTER(IndexEntry), c_u64, c_u64, ctypes.POINTER(FieldDst))
_sig("pgsd_device_wait_read", c_i32, HP)
_sig("pgsd_select_rows", c_i32, c_vp, c_u64, c_vp, ctypes.POINTER(c_u64), c_vp)
_sig("pgsd_device_alloc", c_vp, c_i32, ctypes.c_size_t, c_vp, ctypes.c_size_t)
_sig("pgsd_device_free", c_i32, c_i32, c_vp)
_sig("pgsd_abi_version", c_u32)
_sig("pgsd_device_available", c_i32)


# csrc/pgsd_private.h
_sig("pgsd_reload_tuning", None)
_sig("pgsd_device_of", c_i32, HP)
_sig("pgsd_device_copy", c_i32, c_i32, c_vp, c_vp, ctypes.c_size_t)

if lib.pgsd_abi_version() != ABI_VERSION:
    raise ImportError("libpgsd_amd.so at %s has ABI version %d, these bindings were written against %d: rebuild "
                      "(`make -C pgsd-sph_amd/csrc`)" % (LIB_PATH, lib.pgsd_abi_version(), ABI_VERSION))


def last_error():
    s = lib.pgsd_last_error_string()
    return s.decode("utf-8", "replace") if s else ""
