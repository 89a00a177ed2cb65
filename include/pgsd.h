/* pgsd.h -- C ABI of the MI355X-native PGSD snapshot writer (libpgsd_amd.so).
 *
 * Part 1 is the drop-in boundary: the same eighteen entry points, enums and on-disk structs
 * as the reference's /root/reference/pgsd/pgsd/pgsd.h (each prototype cites the line it
 * replaces), so a caller of the reference (pgsd/pgsd/fl.pyx, pgsd/scripts/benchmark-write.cc,
 * HOOMD-SPH's dump writer) re-links without source changes.  Files written through it are
 * byte-identical to the reference's MPI-IO output for the same calls on the same partition.
 *
 * What changed underneath:
 *   - no <mpi.h>: ranks talk through a small communicator vtable (part 2); back ends are
 *     "self", a single-node shared-memory segment, RCCL over xGMI, or host callbacks
 *     (torch.distributed).  The reference's ~10 collectives per chunk become one small
 *     allgather per chunk -- or ONE per frame (pgsd_set_frame_exchange), or none at all
 *     (pgsd_set_partition).
 *   - all metadata (names, index, file size) is replicated deterministically on every
 *     rank; only rank 0 writes it.  pgsd_find_chunk() is therefore valid on every rank.
 *   - part 3 adds the device path: chunks are packed from HBM-resident particle arrays
 *     by HIP kernels, streamed to pinned host slabs with hipMemcpyAsync and written with
 *     pwrite at the offsets the reference's MPI_File_write_at would use -- or, with PGSD_IO=mpiio
 *     in the environment, by MPI_File_write_at itself (plugin libpgsd_amd_mpiio.so).
 */
#ifndef PGSD_H
#define PGSD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C"
    {
#endif

    /* ------------------------------------------------------------------ part 1: drop-in */

    /* element types; reference pgsd.h:38-69 */
    enum pgsd_type
        {
        PGSD_TYPE_UINT8 = 1,
        PGSD_TYPE_UINT16,
        PGSD_TYPE_UINT32,
        PGSD_TYPE_UINT64,
        PGSD_TYPE_INT8,
        PGSD_TYPE_INT16,
        PGSD_TYPE_INT32,
        PGSD_TYPE_INT64,
        PGSD_TYPE_FLOAT,
        PGSD_TYPE_DOUBLE
        };

    /* reference pgsd.h:72-82 */
    enum pgsd_open_flag
        {
        PGSD_OPEN_READWRITE = 1,
        PGSD_OPEN_READONLY,
        PGSD_OPEN_APPEND
        };

    /* reference pgsd.h:85-120 */
    enum pgsd_error
        {
        PGSD_SUCCESS = 0,
        PGSD_ERROR_IO = -1,
        PGSD_ERROR_INVALID_ARGUMENT = -2,
        PGSD_ERROR_NOT_A_PGSD_FILE = -3,
        PGSD_ERROR_INVALID_PGSD_FILE_VERSION = -4,
        PGSD_ERROR_FILE_CORRUPT = -5,
        PGSD_ERROR_MEMORY_ALLOCATION_FAILED = -6,
        PGSD_ERROR_NAMELIST_FULL = -7,
        PGSD_ERROR_FILE_MUST_BE_WRITABLE = -8,
        PGSD_ERROR_FILE_MUST_BE_READABLE = -9,
        /* additions (never produced by the reference) */
        PGSD_ERROR_DEVICE = -20,      /* HIP runtime / kernel failure, see pgsd_last_error_string() */
        PGSD_ERROR_COMM = -21,        /* communicator failure or ranks disagree */
        PGSD_ERROR_NO_DEVICE = -22    /* device entry point called without a usable GPU */
        };

    enum
        {
        PGSD_NAME_SIZE = 64, /* reference pgsd.h:122-128 */
        PGSD_RESERVED_BYTES = 80
        };

    /* 256-byte file header, on-disk layout; reference pgsd.h:143-174 */
    struct pgsd_header
        {
        uint64_t magic;
        uint64_t index_location;
        uint64_t index_allocated_entries;
        uint64_t namelist_location;
        uint64_t namelist_allocated_entries;
        uint32_t schema_version;
        uint32_t pgsd_version;
        char application[PGSD_NAME_SIZE];
        char schema[PGSD_NAME_SIZE];
        char reserved[PGSD_RESERVED_BYTES];
        };

    /* 32-byte index entry, on-disk layout; reference pgsd.h:182-204 */
    struct pgsd_index_entry
        {
        uint64_t frame;
        uint64_t N;
        int64_t location;
        uint32_t M;
        uint16_t id;
        uint8_t type;
        uint8_t flags;
        };

    /* read-only views into library-owned storage; reference pgsd.h:239-286 */
    struct pgsd_index_buffer
        {
        struct pgsd_index_entry* data;
        size_t size;
        size_t reserved;
        };

    struct pgsd_byte_buffer
        {
        char* data;
        size_t size;
        size_t reserved;
        };

    struct pgsd_name_buffer
        {
        struct pgsd_byte_buffer data;
        size_t n_names;
        };

    /* File handle: caller-allocated, by value, zeroed by the library (reference pgsd.h:297-353).
       The first member of the reference's struct is an MPI_File; here it is a POSIX fd and
       the library state lives behind `impl`.  Every member is read-only to the caller and
       is refreshed on return from each pgsd_* call.  The views (file_index, file_names) are
       valid on EVERY rank until the next call that flushes. */
    struct pgsd_handle
        {
        int fd;
        struct pgsd_header header;
        struct pgsd_index_buffer file_index;
        struct pgsd_name_buffer file_names;
        uint64_t cur_frame;
        long long int file_size;
        enum pgsd_open_flag open_flags;
        uint64_t pending_index_entries;
        uint64_t maximum_write_buffer_size;
        uint64_t index_entries_to_buffer;
        int rank;
        int nprocs;
        void* impl;
        };

    /* reference pgsd.h:362 */
    uint32_t pgsd_make_version(unsigned int major, unsigned int minor);

    /* reference pgsd.h:412-418. Collective over the default communicator (part 2). */
    int pgsd_create_and_open(struct pgsd_handle* handle,
                             const char* fname,
                             const char* application,
                             const char* schema,
                             uint32_t schema_version,
                             enum pgsd_open_flag flags,
                             int exclusive_create);

    /* reference pgsd.h:440 */
    int pgsd_open(struct pgsd_handle* handle, const char* fname, enum pgsd_open_flag flags);

    /* reference pgsd.h:480 */
    int pgsd_close(struct pgsd_handle* handle);

    /* reference pgsd.h:498. Waits for every device chunk of the frame to reach the file. */
    int pgsd_end_frame(struct pgsd_handle* handle);

    /* reference pgsd.h:517 */
    int pgsd_flush(struct pgsd_handle* handle);

    /* reference pgsd.h:551-564.  `data` is HOST memory, borrowed for the call.
       N/M: this rank's rows/columns; N_global/M_global: what the index entry records;
       offset: element offset of this rank's first row inside the chunk;
       global_size: accepted and ignored, exactly as in the reference (pgsd.c:2147-2151 scales it
       and never reads it again): the file advances by the SUM of the ranks' N*M*sizeof(type)
       (MPI_Allreduce SUM, pgsd.c:2240-2246), here one 16-byte allgather per chunk that also
       carries each rank's argument-check status, so a bad argument on one rank fails the call on
       every rank instead of leaving the others inside a collective. */
    int pgsd_write_chunk(struct pgsd_handle* handle,
                         const char* name,
                         enum pgsd_type type,
                         uint64_t N,
                         uint32_t M,
                         uint64_t N_global,
                         uint32_t M_global,
                         uint64_t offset,
                         uint64_t global_size,
                         bool all,
                         uint8_t flags,
                         const void* data);

    /* N_global == PGSD_PARTITION_AUTO (pgsd_write_chunk, pgsd_write_chunk_device, pgsd_write_chunks_device):
       the chunk is partitioned over the ranks in rank order and the library derives the global row
       count and this rank's first row (`offset` / `offset_rows` are then ignored) from the size exchange
       the chunk write performs anyway -- the caller-side MPI_Allgather of the reference's callers
       (benchmark-write.cc:39-45, fl.pyx:596-598) is not needed. */
#define PGSD_PARTITION_AUTO UINT64_MAX

    /* Frame-batched exchange (off after open).
       Off: every chunk write exchanges the ranks' sizes at once -- the reference's per-chunk Allreduces
       (pgsd.c:2157, 2242) as one 16-byte allgather -- so handle->file_size is current after each call,
       and pgsd_end_frame ends with a status exchange that doubles as the barrier after which every
       rank's rows of the frame are in the file.
       On: chunk writes that do not need their file offset at once are QUEUED: replicated host chunks
       (all == false; copied) and all device chunks (packed into the staging arena at once,
       the kernel needs no offset).  ONE allgather at the next pgsd_end_frame (or pgsd_flush,
       pgsd_close, a read, a host chunk with all == true, a buffer-limit setter) carries every rank's
       status word and the byte counts of all queued chunks (a fixed 512-byte message per rank whatever is
       queued -- equal send counts, as ncclAllGather requires; only a frame of more than 62 queued chunks
       takes further rounds); placement then replays the reference's
       decisions in call order, so the file is byte-identical to the unbatched one.  A frame of
       small chunks + device chunks + pgsd_end_frame costs ONE collective (one ncclAllGather on the
       RCCL back end, issued after the pack launch and overlapping it).  Consequences: the handle's
       file_size / pending_index_entries mirror lags until the queue is resolved; a failure on one
       rank is returned there at once and reaches the others with the next exchange; the barrier
       covering a sealed frame is made up at the next pgsd_flush / pgsd_close / read or by the next
       frame's exchange.  Collective like pgsd_end_frame when it turns batching off. */
    int pgsd_set_frame_exchange(struct pgsd_handle* handle, int batched);
    /* Declared partition: NO exchange per chunk and none per frame.  The caller tells the library every rank's
       row count -- what a size exchange would tell it -- because it has exchanged the counts itself (pgsd.hoomd's
       one allgather per frame carries them next to its write/skip votes) or because they have not changed since
       the last snapshot (a run without particle migration writes frame after frame with ZERO collectives).  While a
       partition is declared:
         - a chunk written with N_global == PGSD_PARTITION_AUTO is partitioned by it: N must be rows[rank]; the
           global row count is the sum, this rank's first row the prefix;
         - any other chunk must have the same byte size on every rank (replicated data, or data every rank writes
           alike: the reference binding's default-argument call shape);
       and every chunk is placed at once from that knowledge, byte for byte where the exchanges would have put it
       (every golden and fuzz scenario is replayed this way).  Errors: a call that fails this rank's checks returns
       the code at once; the other ranks learn of it at the next synchronisation point (pgsd_flush / pgsd_close / a
       read), like a failed write; the frame's barrier is made up there too, as with the batched exchange.  That
       exchange also compares the ranks' views of the file (size, frame counter, numbers of names and index entries):
       ranks that brought different sizes for a chunk that is not partitioned -- the one thing the declaration takes
       on trust -- all get PGSD_ERROR_COMM there instead of going on with different layouts of one file.
       rows == NULL clears the declaration.  Every rank must declare the same vector (n_ranks = the communicator's
       size). */
    int pgsd_set_partition(struct pgsd_handle* handle, const uint64_t* rows, uint32_t n_ranks);

    /* pgsd_read_chunk / pgsd_read_chunk_device on a writable handle flush first like the reference's (pgsd.c:2436
       -2537): collective, because a rank may read rows another rank wrote.  With `on`, reads are LOCAL: they wait
       for this rank's own asynchronous copies only and take no part in a collective -- for rows the caller knows to
       be in the file: its own rows of a sealed frame, or any rows of a file that was opened after they were
       written.  (pgsd.hoomd reads each rank's rows of frame 0 this way when it first compares an array with them;
       which arrays a rank compares need not be the same on every rank.)  Chunks of the open frame are not flushed
       by a local read.  The lookup in front of the read (pgsd_find_chunk, pgsd_find_matching_chunk_name) is local
       too while this is on and there are several ranks: it sees what the last flush committed -- a frame that held
       buffered small chunks only is not flushed by pgsd_end_frame (pgsd.c:1941-1950), and its chunks are not found
       by a local lookup until the next collective flush. */
    int pgsd_set_local_reads(struct pgsd_handle* handle, int on);
    /* Collectives of a handle: `collectives` = allgathers and barriers it has issued on its communicator since it
       was opened (never reset: the evidence behind "one collective per frame"); the rest = what its allgathers
       cost the calling thread (wall clock around the communicator's allgather: transport latency + the wait for
       the slowest rank): bench.py's `exchange_us`. */
    struct pgsd_exchange_stats
        {
        uint64_t collectives; /* since open */
        uint64_t count;   /* allgathers since open / last reset */
        double total_us;
        double max_us;
        double min_us;    /* the quickest one: a rank that arrived last waits for nobody */
        };
    int pgsd_get_exchange_stats(struct pgsd_handle* handle, struct pgsd_exchange_stats* out, int reset);

    /* reference pgsd.h:581-582.  The reference flushes first (collective).  Here the index is replicated:
       the call flushes (collectively) only while written chunks / names / index entries are still pending --
       replicated state, every rank decides alike -- and is LOCAL otherwise, so a lookup made by one rank
       only (after pgsd_end_frame) cannot strand the other ranks in a collective.  Same for
       pgsd_find_matching_chunk_name. */
    const struct pgsd_index_entry*
    pgsd_find_chunk(struct pgsd_handle* handle, uint64_t frame, const char* name);

    /* reference pgsd.h:604-610 */
    int pgsd_read_chunk(struct pgsd_handle* handle,
                        void* data,
                        const struct pgsd_index_entry* chunk,
                        uint64_t N,
                        uint32_t M,
                        uint32_t offset,
                        bool all);

    /* reference pgsd.h:620, 630, 638 */
    uint64_t pgsd_get_nframes(struct pgsd_handle* handle);
    uint64_t pgsd_get_nnames(struct pgsd_handle* handle);
    size_t pgsd_sizeof_type(enum pgsd_type type);

    /* reference pgsd.h:659-660 */
    const char*
    pgsd_find_matching_chunk_name(struct pgsd_handle* handle, const char* match, const char* prev);

    /* reference pgsd.h:686, 701, 711, 729 */
    uint64_t pgsd_get_maximum_write_buffer_size(struct pgsd_handle* handle);
    int pgsd_set_maximum_write_buffer_size(struct pgsd_handle* handle, uint64_t size);
    uint64_t pgsd_get_index_entries_to_buffer(struct pgsd_handle* handle);
    int pgsd_set_index_entries_to_buffer(struct pgsd_handle* handle, uint64_t number);

    /* reference pgsd.h:735: broadcast an index entry from rank 0 (unused by the library
       itself, kept for callers). */
    void pgsd_bcast_index_entry(struct pgsd_index_entry* e);

    /* Text of the most recent failure on this thread (HIP error strings, comm mismatches). */
    const char* pgsd_last_error_string(void);

    /* ------------------------------------------------------------ part 2: communicator */

    /* The process-wide default communicator plays the role of the reference's hard-coded
       MPI_COMM_WORLD (pgsd.c:106-202, 1748).  A handle captures it when it is opened.
       All ranks must call the collective pgsd_* functions in the same order. */
    struct pgsd_comm
        {
        void* ctx;
        int rank;
        int size;
        /* gather `bytes` bytes from every rank into recv[rank*bytes ...]; 0 on success */
        int (*allgather)(void* ctx, const void* send, void* recv, size_t bytes);
        /* optional; NULL = a 1-byte allgather */
        int (*barrier)(void* ctx);
        /* optional */
        void (*destroy)(void* ctx);
        };

    /* Install a back end as the process default: a caller-provided one (MPI, torch.distributed callbacks), or one
       made by pgsd_comm_create_shm / pgsd_comm_create_rccl below.  The struct is copied and the default OWNS the
       context from then on (its `destroy` runs when the default is replaced and the last handle opened on it is
       closed): a communicator that was installed is not passed to pgsd_comm_release. */
    int pgsd_comm_set_default(const struct pgsd_comm* comm);
    /* a /dev/shm communicator installed from the environment: PGSD_RANK / PGSD_NRANKS / PGSD_SHM_NAME, else
       RANK / WORLD_SIZE / MASTER_PORT (torchrun), else single rank */
    int pgsd_comm_init_from_env(void);
    /* RCCL over xGMI (pgsd_comm_create_rccl): `unique_id` is the 128-byte ncclUniqueId created by
       pgsd_comm_rccl_unique_id() on one rank and distributed by the caller.  Allgathers run
       as ncclAllGather on device buffers on a private HIP stream. */
    int pgsd_comm_rccl_unique_id(void* unique_id_128);
    /* What one rank can know BEFORE the collective bootstrap above: librccl loads with the entry points used and
       `device` (-1: the current one) exists.  Callers let the ranks agree on it first, so that nobody waits inside
       ncclCommInitRank for a rank that could never have come.  An exchange on an RCCL communicator waits at most
       PGSD_COMM_TIMEOUT_S seconds (environment, default 120) for the other ranks; then -- or when RCCL reports an
       asynchronous error -- the communicator is aborted (ncclCommAbort), the call returns PGSD_ERROR_COMM and so
       does every later collective on it. */
    int pgsd_comm_rccl_available(int device);
    /* back to a single rank (the state after library load) */
    int pgsd_comm_finalize(void);
    int pgsd_comm_rank(void);
    int pgsd_comm_size(void);
    int pgsd_comm_allgather(const void* send, void* recv, size_t bytes);
    int pgsd_comm_barrier(void);

    /* Making communicators.  pgsd_comm_create_shm: ranks of ONE node through a /dev/shm segment (process-shared
       barrier + slots); pgsd_comm_create_rccl: RCCL over xGMI.  Both fill in `out` without installing anything.
       Either the caller installs it as the process default (pgsd_comm_set_default, which takes it over), or it
       stays a communicator that is NOT the default: several ranks in ONE process (one thread per GPU, each
       with a handle of its own), or independent groups of ranks side by side.  pgsd_create_and_open_on /
       pgsd_open_on open a file on such a communicator (the struct is copied; the caller keeps the communicator
       alive until every handle opened on it is closed, then lets it go with pgsd_comm_release).  Everything else
       is as with the default communicator: the pgsd_* calls on the handle are collective over ITS communicator. */
    int pgsd_comm_create_shm(const char* name, int rank, int size, struct pgsd_comm* out);
    int pgsd_comm_create_rccl(const void* unique_id_128, int rank, int size, int device, struct pgsd_comm* out);
    void pgsd_comm_release(struct pgsd_comm* comm);
    int pgsd_create_and_open_on(const struct pgsd_comm* comm,
                                struct pgsd_handle* handle,
                                const char* fname,
                                const char* application,
                                const char* schema,
                                uint32_t schema_version,
                                enum pgsd_open_flag flags,
                                int exclusive_create);
    int pgsd_open_on(const struct pgsd_comm* comm, struct pgsd_handle* handle, const char* fname,
                     enum pgsd_open_flag flags);
    /* An allgather over the communicator `handle` was opened on (counted among its collectives): what a
       caller-side exchange that belongs to the file -- pgsd.hoomd's row counts and write/skip votes -- uses
       instead of the process default. */
    int pgsd_handle_allgather(struct pgsd_handle* handle, const void* send, void* recv, size_t bytes);

    /* The per-frame exchange of the reference's callers (benchmark-write.cc:39-45,
       fl.pyx:596-598): allgather every rank's local row count; returns the exclusive
       prefix (this rank's first row) and the total.  counts may be NULL. */
    int pgsd_partition_rows(uint64_t n_local, uint64_t* row0, uint64_t* n_global, uint64_t* counts);

    /* --------------------------------------------------------------- part 3: device path */

    /* One output chunk fed from an HBM-resident source array:
         chunk[i][c] = convert(src[(order ? order[i] : i) * src_stride + src_col0 + c]),
         i < N, c < M, converted from src_type to the chunk's pgsd_type.
       Supported conversions: same type; integer<->integer (wrap / extend by source
       signedness); f64->f32 (round to nearest even); f32->f64; <=32-bit integer->float;
       bitcast (low bytes of the element unchanged, e.g. HOOMD's typeid kept in position.w). */
    struct pgsd_field_desc
        {
        const void* src;       /* device pointer */
        const uint32_t* order; /* device pointer or NULL */
        uint32_t src_type;     /* enum pgsd_type */
        uint32_t src_stride;   /* elements per source row (4 for float4 / double4) */
        uint32_t src_col0;     /* first source column */
        uint32_t bitcast;      /* 0 = value conversion, 1 = reinterpret low bytes */
        };

    /* Device twin of pgsd_write_chunk: same arguments, but the rows come from device
       memory described by `src`.  Host bookkeeping (name, index entry, file location) is
       done before returning; the pack kernel, the device->pinned-host copy and the pwrite
       run asynchronously and are complete when pgsd_end_frame() returns.  The source
       arrays must stay unmodified until pgsd_device_wait_packed() or pgsd_end_frame(). */
    int pgsd_write_chunk_device(struct pgsd_handle* handle,
                                const char* name,
                                enum pgsd_type type,
                                uint64_t N,
                                uint32_t M,
                                uint64_t N_global,
                                uint32_t M_global,
                                uint64_t offset,
                                uint64_t global_size,
                                bool all,
                                uint8_t flags,
                                const struct pgsd_field_desc* src);

    /* Several per-particle chunks (all==true, same N / N_global / row offset) packed by ONE
       fused kernel launch, e.g. position+velocity+typeid. offset_rows is this rank's first
       row (elements = offset_rows * M per chunk). */
    struct pgsd_chunk_req
        {
        const char* name;
        uint32_t type; /* enum pgsd_type of the chunk */
        uint32_t M;
        struct pgsd_field_desc src;
        };
    int pgsd_write_chunks_device(struct pgsd_handle* handle,
                                 uint32_t n_chunks,
                                 const struct pgsd_chunk_req* chunks,
                                 uint64_t N,
                                 uint64_t N_global,
                                 uint64_t offset_rows);

    /* pgsd_write_chunks_device in two steps, for callers that know their device fields before they know the
       frame's other chunks: pgsd_stage_chunks_device launches the fused pack AT ONCE (kernel and, for small
       frames, the PCIe crossing then run under whatever the caller does next) and returns a ticket;
       pgsd_write_staged_chunks writes chunks [first, first + count) of the ticket at THIS point of the frame's
       chunk order -- same arguments, same exchange, same placement as pgsd_write_chunks_device, minus the
       launch.  Chunks of a ticket that were never written are dropped by the next pgsd_end_frame / pgsd_close.
       (For a simulation that can stage right behind its last kernel and has other work before the frame is
       sealed, and for the elision test below, which needs the packed bytes before it is known what is written.
       Staging merely to hide the kernel behind pgsd.hoomd's schema bookkeeping was measured and bought nothing --
       the extra call cost what the hidden kernel wait saved) */
    int pgsd_stage_chunks_device(struct pgsd_handle* handle, uint32_t n_chunks, const struct pgsd_chunk_req* chunks,
                                 uint64_t N, uint64_t* ticket);
    int pgsd_write_staged_chunks(struct pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                 uint64_t N_global, uint64_t offset_rows);

    /* Elision for arrays that live in HBM.  pgsd.hoomd does not write a per-particle array that equals frame 0's
       (reference: hoomd.py:654-694, a numpy comparison of host arrays).  For device arrays the test runs on the
       GPU, on staged chunks that have not been written yet:
       pgsd_compare_staged_chunks compares the PACKED rows of chunks [first, first + count) of a ticket with
       ref[i] -- device memory holding the same rows of the other frame as the chunk stores them, N * M *
       sizeof(type) bytes (read with pgsd_read_chunk_device, or kept with pgsd_copy_staged_chunks) -- and sets
       equal[i] = 1 when they are equal, 0 when not or when ref[i] is NULL (a rank without rows: 1).
       ref_bytes (may be NULL): what ref[i] holds.  A reference SHORTER than the chunk REPEATS -- byte j of the chunk
       is compared with byte j % ref_bytes[i] -- so a few thousand rows of a default value stand for any number of
       rows (the "equals the default and frame 0 has no such chunk" half of hoomd.py:654-694); such a reference is
       16-byte aligned and at least 4096 bytes, a multiple of 16 bytes and of whole rows.
       Equality is numpy.array_equal's (hoomd.py:679-682): integer chunks by their bytes, float / double chunks by
       VALUE -- a NaN equals nothing, itself included; +0.0 equals -0.0 -- so the decision is the one the reference's
       host comparison takes for the same values.
       One kernel launch behind the pack -- and behind whatever the caller's source stream
       (pgsd_device_set_source_stream) holds, like the pack itself: the references may have been written there a
       moment ago --, one stream wait; local, no collective: the caller agrees the outcome over the ranks like any
       other write / skip decision and then writes (pgsd_write_staged_chunks) or does not (unwritten chunks are
       dropped by pgsd_end_frame).
       pgsd_copy_staged_chunks copies the packed bytes into caller-owned device memory dst[i] (NULL: skipped),
       asynchronously behind the pack (and behind the caller's source stream: the destinations are the caller's):
       complete after pgsd_device_wait_packed, a comparison or pgsd_end_frame. */
    int pgsd_compare_staged_chunks(struct pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                   const void* const* ref, const uint64_t* ref_bytes, uint8_t* equal);
    int pgsd_copy_staged_chunks(struct pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                void* const* dst);

    /* Seal the frame like pgsd_end_frame, but do not wait for its device chunks: the frame
       counter advances and names / small-chunk buffers / index entries are committed now, while
       the device->host copies and the pwrite()s of the frame keep running behind the caller.
       What a simulation waits for per snapshot is then only the pack kernels
       (pgsd_device_wait_packed), not the file.  The frame's bytes are in the file after
       pgsd_frame_sync(), or after the next pgsd_end_frame / pgsd_flush / pgsd_close /
       pgsd_find_chunk / pgsd_read_chunk on this handle.  Also when the on-disk index has to be
       relocated (every few hundred chunks): where the file will end is known from the ranks' own
       placements, no byte has to be waited for (until round 5 the call fell back to the synchronous
       pgsd_end_frame there).  The file layout is identical either way.  Collective like pgsd_end_frame. */
    int pgsd_end_frame_async(struct pgsd_handle* handle);
    /* Wait until the device chunks of all asynchronously sealed frames of THIS rank are in the file. */
    int pgsd_frame_sync(struct pgsd_handle* handle);

    /* Block until every pack kernel issued for the open frame has finished (the source
       arrays may be overwritten again); copies and file writes keep running. */
    int pgsd_device_wait_packed(struct pgsd_handle* handle);

    struct pgsd_device_config
        {
        int device;             /* HIP device ordinal; -1 = current */
        uint64_t slab_bytes;    /* pinned host staging slab size (default 16 MiB) */
        uint32_t n_slabs;       /* slabs in the ring (default 16) */
        uint32_t n_writers;     /* pwrite threads (default 1: one file = one inode lock) */
        uint32_t profile;       /* 1 = bracket every pack launch with HIP events */
        uint32_t prealloc_mib;  /* 0: staging blocks (256 MiB of HBM each) and ring slabs are allocated when a frame first
                                   needs them and kept until close.  > 0: this call allocates that much staging (rounded up to
                                   whole blocks), the WHOLE ring and the comparison's answer words, and has the runtime load the kernels'
                                   code objects (a millisecond each at first use), so that no snapshot of the run meets any of it: a
                                   hipMalloc / hipHostMalloc in the middle of a run costs 0.03 ... 38 ms depending on the
                                   box's state and stalls streams it has nothing to do with */
        };
    int pgsd_device_configure(struct pgsd_handle* handle, const struct pgsd_device_config* cfg);

    /* Stream ordering: the pack kernels run on a private stream.  Every device write first
       records an event on the caller's *source stream* -- the stream on which the kernels that
       produce the particle arrays were enqueued (hipStream_t as void*; NULL = the null stream,
       which is also PyTorch's default stream) -- and makes the pack stream wait for it, so
       arrays still being written by earlier asynchronous work are packed only when complete.
       The unpack of pgsd_device_wait_read() waits for the same stream before it writes the
       destination arrays. */
    int pgsd_device_set_source_stream(struct pgsd_handle* handle, void* stream);

    struct pgsd_device_stats
        {
        uint64_t pack_launches;     /* kernels launched since open / last reset */
        double pack_ms;             /* sum of event-timed kernel durations (profile=1) */
        uint64_t pack_rows;         /* rows (particles) packed, summed over launches */
        uint64_t pack_bytes_out;    /* chunk bytes produced */
        uint64_t pack_bytes_in;     /* source bytes the kernels had to read (algorithmic) */
        uint64_t d2h_bytes;
        uint64_t written_bytes;
        double d2h_ms;              /* sum of event-timed copy durations (profile=1) */
        double write_ms;            /* sum of pwrite wall time over writer threads */
        };
    int pgsd_device_get_stats(struct pgsd_handle* handle, struct pgsd_device_stats* out, int reset);

    /* Stream compaction for filtered snapshots: out_index[k] = i for the k-th row whose flag byte is non-zero
       (stable; device memory, room for N entries), *out_count (HOST memory) = number selected.  Wavefront
       ballot/popcount scans + one cross-block pass on `stream` (a hipStream_t passed as void*; NULL = the null
       stream); the scratch space is the library's (kept per device, grown on demand); the call returns when the
       count is known, i.e. after synchronising `stream`. */
    int pgsd_select_rows(const uint8_t* flags, uint64_t N, uint32_t* out_index, uint64_t* out_count, void* stream);

    /* Device memory owned by the library, for callers that have no allocator of their own at hand (pgsd.fl /
       pgsd.hoomd keep the references of the GPU-side elision, the rows of device reads and the index lists of
       pgsd.fl.select_rows in it, so that the Python device path needs no tensor library): `bytes` bytes on
       `device` (-1: the current one), 256-byte aligned.  `pattern` (may be NULL: contents undefined): host memory
       of pattern_bytes bytes that is repeated over the whole buffer (a default value's rows; zeros).  NULL on
       failure (pgsd_last_error_string).  pgsd_device_free takes what pgsd_device_alloc returned. */
    void* pgsd_device_alloc(int device, size_t bytes, const void* pattern, size_t pattern_bytes);
    int pgsd_device_free(int device, void* ptr);

    /* --- read side of the device path (restart files): file -> pinned slabs -> HBM -> unpack --- */

    /* Where the rows of a chunk go in device memory (inverse of pgsd_field_desc):
         dst[(order ? order[i] : i) * dst_stride + dst_col0 + c] = convert(chunk[row_offset + i][c])
       converted from the chunk's type to dst_type (same rules as the pack direction; bitcast
       needs equal element sizes).  Columns of the destination rows that no chunk writes are
       left untouched, so position.xyz and the type id can be restored into one Scalar4 array --
       unless fill_rest is set: then every column of the row that no chunk of the same launch
       (pgsd_device_wait_read) writes receives fill_bits, the bit pattern of
       one destination element.  Velocity without a mass chunk thus restores as {vx, vy, vz, 1.0f}:
       ONE whole 16-byte row per particle instead of a 12-byte piece at a 16-byte stride. */
    struct pgsd_field_dst
        {
        void* dst;             /* device pointer */
        const uint32_t* order; /* device scatter index or NULL */
        uint32_t dst_type;     /* enum pgsd_type of a destination element */
        uint32_t dst_stride;   /* elements per destination row */
        uint32_t dst_col0;     /* first destination column */
        uint32_t bitcast;
        uint32_t fill_rest;    /* 1: columns of the row no chunk of the launch writes are set to fill_bits */
        uint32_t reserved;
        uint64_t fill_bits;    /* low sizeof(dst_type) bytes = one destination element */
        };

    /* Device twin of pgsd_read_chunk (reference pgsd.h:604-610) for a row slab: rows
       [row_offset, row_offset + N) of `chunk` (found with pgsd_find_chunk, valid on every rank)
       are read with pread (16 threads, 4 MiB pieces, a pinned ring of their own; PGSD_READERS /
       PGSD_READ_PIECE_MIB override), streamed to HBM and unpacked by a HIP
       kernel.  Asynchronous: complete after pgsd_device_wait_read(), which also issues the
       unpack of everything read since the last wait as one launch (chunks that together restore
       whole rows of one array -- position.xyz + type id -- are written as whole rows). Every
       rank reads its own partition; no collective is involved. */
    int pgsd_read_chunk_device(struct pgsd_handle* handle,
                               const struct pgsd_index_entry* chunk,
                               uint64_t N,
                               uint64_t row_offset,
                               const struct pgsd_field_dst* dst);
    int pgsd_device_wait_read(struct pgsd_handle* handle);

    /* 1 when a gfx950-capable HIP device is visible to this process */
    int pgsd_device_available(void);

    /* A closed handle of the default pipeline geometry PARKS what is dear to make -- two streams, up to four 16 MiB
       pinned slabs, the pinned arena of the small-frame path, up to four 256 MiB staging arenas in HBM (1 GiB), idle
       events -- for the next handle on the same device: at most two sets per process, held until the process ends
       (PGSD_NO_PARKING in the environment: nothing is parked). */

    /* The binary interface of this header.  Entry points that change their signature in place bump it; bindings
       that resolve symbols at run time (ctypes, dlsym, the Cython module of pgsd.fl) compare it with the
       PGSD_ABI_VERSION they were written against before the first call. */
#define PGSD_ABI_VERSION 5u
    uint32_t pgsd_abi_version(void);

#ifdef __cplusplus
    }
#endif

#endif /* PGSD_H */
