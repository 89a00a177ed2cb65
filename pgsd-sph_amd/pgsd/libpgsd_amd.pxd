# libpgsd_amd.pxd -- Cython view of include/pgsd.h (the C ABI of libpgsd_amd.so).
#
# Counterpart of the reference's pgsd/pgsd/libpgsd.pxd (lines 51-144): the same sixteen entry points and
# on-disk structs, minus `mpi4py.MPI` (the reference cimports it for the MPI_File member of its handle,
# libpgsd.pxd:6; this library's handle starts with a POSIX fd), plus the device path.  The declarations are
# checked by the C compiler against the real header when pgsd/_fl.pyx is built: a prototype that drifts
# from include/pgsd.h does not compile.
from libc.stdint cimport uint8_t, uint16_t, uint32_t, uint64_t, int64_t


cdef extern from "pgsd.h" nogil:
    cdef enum pgsd_type:
        PGSD_TYPE_UINT8
        PGSD_TYPE_UINT16
        PGSD_TYPE_UINT32
        PGSD_TYPE_UINT64
        PGSD_TYPE_INT8
        PGSD_TYPE_INT16
        PGSD_TYPE_INT32
        PGSD_TYPE_INT64
        PGSD_TYPE_FLOAT
        PGSD_TYPE_DOUBLE

    cdef enum pgsd_open_flag:
        PGSD_OPEN_READWRITE
        PGSD_OPEN_READONLY
        PGSD_OPEN_APPEND

    cdef enum pgsd_error:
        PGSD_SUCCESS
        PGSD_ERROR_IO
        PGSD_ERROR_INVALID_ARGUMENT
        PGSD_ERROR_NOT_A_PGSD_FILE
        PGSD_ERROR_INVALID_PGSD_FILE_VERSION
        PGSD_ERROR_FILE_CORRUPT
        PGSD_ERROR_MEMORY_ALLOCATION_FAILED
        PGSD_ERROR_NAMELIST_FULL
        PGSD_ERROR_FILE_MUST_BE_WRITABLE
        PGSD_ERROR_FILE_MUST_BE_READABLE
        PGSD_ERROR_DEVICE
        PGSD_ERROR_COMM
        PGSD_ERROR_NO_DEVICE

    cdef uint64_t PGSD_PARTITION_AUTO

    cdef struct pgsd_header:
        uint64_t magic
        uint64_t index_location
        uint64_t index_allocated_entries
        uint64_t namelist_location
        uint64_t namelist_allocated_entries
        uint32_t schema_version
        uint32_t pgsd_version
        char application[64]
        char schema[64]
        char reserved[80]

    cdef struct pgsd_index_entry:
        uint64_t frame
        uint64_t N
        int64_t location
        uint32_t M
        uint16_t id
        uint8_t type
        uint8_t flags

    cdef struct pgsd_handle:
        int fd
        pgsd_header header
        uint64_t cur_frame
        long long file_size
        pgsd_open_flag open_flags
        uint64_t pending_index_entries
        uint64_t maximum_write_buffer_size
        uint64_t index_entries_to_buffer
        int rank
        int nprocs
        void* impl

    cdef struct pgsd_comm:
        void* ctx
        int rank
        int size

    # ---- part 1: the reference's entry points (pgsd.h:362-735 there)
    uint32_t pgsd_make_version(unsigned int major, unsigned int minor)
    int pgsd_create_and_open(pgsd_handle* handle, const char* fname, const char* application, const char* schema,
                             uint32_t schema_version, pgsd_open_flag flags, int exclusive_create)
    int pgsd_open(pgsd_handle* handle, const char* fname, pgsd_open_flag flags)
    int pgsd_close(pgsd_handle* handle)
    int pgsd_end_frame(pgsd_handle* handle)
    int pgsd_flush(pgsd_handle* handle)
    int pgsd_write_chunk(pgsd_handle* handle, const char* name, pgsd_type type, uint64_t N, uint32_t M,
                         uint64_t N_global, uint32_t M_global, uint64_t offset, uint64_t global_size, bint all,
                         uint8_t flags, const void* data)
    const pgsd_index_entry* pgsd_find_chunk(pgsd_handle* handle, uint64_t frame, const char* name)
    int pgsd_read_chunk(pgsd_handle* handle, void* data, const pgsd_index_entry* chunk, uint64_t N, uint32_t M,
                        uint32_t offset, bint all)
    uint64_t pgsd_get_nframes(pgsd_handle* handle)
    uint64_t pgsd_get_nnames(pgsd_handle* handle)
    size_t pgsd_sizeof_type(pgsd_type type)
    const char* pgsd_find_matching_chunk_name(pgsd_handle* handle, const char* match, const char* prev)
    uint64_t pgsd_get_maximum_write_buffer_size(pgsd_handle* handle)
    int pgsd_set_maximum_write_buffer_size(pgsd_handle* handle, uint64_t size)
    uint64_t pgsd_get_index_entries_to_buffer(pgsd_handle* handle)
    int pgsd_set_index_entries_to_buffer(pgsd_handle* handle, uint64_t number)
    const char* pgsd_last_error_string()

    # ---- frame exchange, communicators
    int pgsd_set_frame_exchange(pgsd_handle* handle, int batched)
    int pgsd_set_local_reads(pgsd_handle* handle, int on)
    int pgsd_set_partition(pgsd_handle* handle, const uint64_t* rows, uint32_t n_ranks)
    cdef struct pgsd_exchange_stats:
        uint64_t collectives
        uint64_t count
        double total_us
        double max_us
        double min_us
    int pgsd_get_exchange_stats(pgsd_handle* handle, pgsd_exchange_stats* out, int reset)
    int pgsd_comm_size()
    int pgsd_create_and_open_on(const pgsd_comm* comm, pgsd_handle* handle, const char* fname, const char* application,
                                const char* schema, uint32_t schema_version, pgsd_open_flag flags, int exclusive_create)
    int pgsd_open_on(const pgsd_comm* comm, pgsd_handle* handle, const char* fname, pgsd_open_flag flags)
    int pgsd_handle_allgather(pgsd_handle* handle, const void* send, void* recv, size_t bytes)

    # ---- part 3: device path
    cdef struct pgsd_field_desc:
        const void* src
        const uint32_t* order
        uint32_t src_type
        uint32_t src_stride
        uint32_t src_col0
        uint32_t bitcast

    cdef struct pgsd_chunk_req:
        const char* name
        uint32_t type
        uint32_t M
        pgsd_field_desc src

    cdef struct pgsd_field_dst:
        void* dst
        const uint32_t* order
        uint32_t dst_type
        uint32_t dst_stride
        uint32_t dst_col0
        uint32_t bitcast
        uint32_t fill_rest
        uint32_t reserved
        uint64_t fill_bits

    cdef struct pgsd_device_config:
        int device
        uint64_t slab_bytes
        uint32_t n_slabs
        uint32_t n_writers
        uint32_t profile
        uint32_t prealloc_mib

    cdef struct pgsd_device_stats:
        uint64_t pack_launches
        double pack_ms
        uint64_t pack_rows
        uint64_t pack_bytes_out
        uint64_t pack_bytes_in
        uint64_t d2h_bytes
        uint64_t written_bytes
        double d2h_ms
        double write_ms

    int pgsd_write_chunk_device(pgsd_handle* handle, const char* name, pgsd_type type, uint64_t N, uint32_t M,
                                uint64_t N_global, uint32_t M_global, uint64_t offset, uint64_t global_size, bint all,
                                uint8_t flags, const pgsd_field_desc* src)
    int pgsd_write_chunks_device(pgsd_handle* handle, uint32_t n_chunks, const pgsd_chunk_req* chunks, uint64_t N,
                                 uint64_t N_global, uint64_t offset_rows)
    int pgsd_stage_chunks_device(pgsd_handle* handle, uint32_t n_chunks, const pgsd_chunk_req* chunks, uint64_t N,
                                 uint64_t* ticket)
    int pgsd_write_staged_chunks(pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                 uint64_t N_global, uint64_t offset_rows)
    int pgsd_compare_staged_chunks(pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                   const void* const* ref, const uint64_t* ref_bytes, uint8_t* equal)
    int pgsd_copy_staged_chunks(pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                void* const* dst)
    int pgsd_end_frame_async(pgsd_handle* handle)
    int pgsd_frame_sync(pgsd_handle* handle)
    int pgsd_device_wait_packed(pgsd_handle* handle)
    int pgsd_device_configure(pgsd_handle* handle, const pgsd_device_config* cfg)
    int pgsd_device_set_source_stream(pgsd_handle* handle, void* stream)
    int pgsd_device_get_stats(pgsd_handle* handle, pgsd_device_stats* out, int reset)
    int pgsd_select_rows(const uint8_t* flags, uint64_t N, uint32_t* out_index, uint64_t* out_count, void* stream)
    void* pgsd_device_alloc(int device, size_t bytes, const void* pattern, size_t pattern_bytes)
    int pgsd_device_free(int device, void* ptr)
    uint32_t PGSD_ABI_VERSION
    uint32_t pgsd_abi_version()
    int pgsd_read_chunk_device(pgsd_handle* handle, const pgsd_index_entry* chunk, uint64_t N, uint64_t row_offset,
                               const pgsd_field_dst* dst)
    int pgsd_device_wait_read(pgsd_handle* handle)


# entry points that are not part of the public C ABI (csrc/pgsd_private.h): queue plumbing and binding helpers
cdef extern from "pgsd_private.h" nogil:
    int pgsd_set_deferred_rows(pgsd_handle* handle, int on)
    int pgsd_frame_exchange(pgsd_handle* handle)
    int pgsd_device_of(pgsd_handle* handle)
    int pgsd_device_copy(int device, void* dst, const void* src, size_t bytes)
