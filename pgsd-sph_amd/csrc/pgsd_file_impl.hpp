// pgsd_file_impl.hpp -- state of an open file (Impl) and the functions the three translation units of the file layer
// share: pgsd_container.cpp (the GSD v2 container: skeleton, names, index, flush, open / close), pgsd_placement.cpp
// (where the bytes of a chunk go: exchanges, the frame queue, declared partitions, the device write path) and
// pgsd_read.cpp (lookups and reads, host and device).  Until round 5 they were one 2 900-line file (pgsd_file.cpp).
#ifndef PGSD_FILE_IMPL_HPP
#define PGSD_FILE_IMPL_HPP

#include "pgsd_internal.hpp"

#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>

namespace pgsd_amd
    {
// constants of the format / reference defaults, pgsd.c:54-102
static const uint64_t MAGIC_ID = 0x65DF65DF65DF65DFull;
enum
    {
    INITIAL_INDEX_SIZE = 128,
    INITIAL_NAME_BUFFER_SIZE = 1024,
    INITIAL_FRAME_NAMES_SIZE = 64,
    CURRENT_FILE_VERSION = 2
    };
static const uint64_t DEFAULT_MAXIMUM_WRITE_BUFFER_SIZE = 64ull * 1024 * 1024;
static const uint64_t DEFAULT_INDEX_ENTRIES_TO_BUFFER = 256ull * 1024;

static_assert(sizeof(pgsd_header) == 256, "GSD header is 256 bytes on disk");
static_assert(sizeof(pgsd_index_entry) == 32, "GSD index entry is 32 bytes on disk");

inline uint32_t make_version(unsigned major, unsigned minor)
    {
    return major << 16 | minor; // pgsd.c:1705-1708
    }

// Byte buffer whose capacity doubles the way pgsd_byte_buffer_append does (pgsd.c:490-525).
// The capacity of the name list is observable: it decides when the namelist block is
// relocated to the end of the file and how many bytes are rewritten.
struct ByteBuf
    {
    std::vector<char> d; // d.size() is the "reserved" of the reference
    size_t size = 0;

    void allocate(size_t reserve)
        {
        d.assign(reserve, 0);
        size = 0;
        }

    size_t reserved() const
        {
        return d.size();
        }

    void append(const char* data, size_t n)
        {
        if (n == 0)
            return;
        if (size + n > d.size())
            {
            size_t nr = d.size() * 2;
            while (size + n >= nr)
                nr *= 2;
            d.resize(nr, 0);
            }
        memcpy(d.data() + size, data, n);
        size += n;
        }
    };

// A chunk write whose placement waits for the frame's size exchange (pgsd_set_frame_exchange).
struct Queued
    {
    std::string name;
    uint32_t type = 0;
    uint64_t N = 0;
    uint32_t M = 0;
    uint64_t N_global = 0; // PGSD_PARTITION_AUTO: derived from the exchange, like `offset`
    uint32_t M_global = 0;
    uint64_t offset = 0;
    bool all = false;
    int local_rc = PGSD_SUCCESS;    // this rank's argument / staging verdict
    std::vector<char> host;         // copy of a small host chunk
    const void* borrowed = nullptr; // host rows of the call that is resolving the queue right now
    int ticket = -1;                // device chunk: packed in the staging arena, waiting for its place
    size_t ticket_index = 0;
    };

// chunks packed ahead of their place in the frame (pgsd_stage_chunks_device)
struct EarlyStage
    {
    int ticket = -1;
    uint64_t N = 0;
    int local_rc = PGSD_SUCCESS;
    std::vector<std::string> names;
    std::vector<uint32_t> types, Ms;
    std::vector<bool> claimed;
    };

struct Impl
    {
    std::shared_ptr<CommBox> comm_box; // keeps the communicator alive for as long as the file is open
    pgsd_comm comm;
    int rank = 0, P = 1;
    int fd = -1;
    pgsd_header header;
    std::vector<pgsd_index_entry> file_index; // .size() == entries allocated on disk
    size_t file_index_size = 0;               // entries in use
    std::vector<pgsd_index_entry> frame_index, buffer_index;
    ByteBuf file_names, frame_names;
    size_t file_n_names = 0, frame_n_names = 0;
    std::unordered_map<std::string, uint16_t> name_map;
    std::vector<char> write_buffer;   // this rank's buffered small-chunk bytes
    std::vector<uint64_t> wb_sizes;   // every rank's write_buffer size (replicated)
    uint64_t cur_frame = 0;
    long long file_size = 0;
    pgsd_open_flag flags = PGSD_OPEN_READWRITE;
    uint64_t pending = 0;
    uint64_t maxbuf = DEFAULT_MAXIMUM_WRITE_BUFFER_SIZE;
    uint64_t idxbuf = DEFAULT_INDEX_ENTRIES_TO_BUFFER;
    bool dirty_data = false; // a direct/device chunk was written since the last flush
    bool inflight = false;   // an asynchronous end_frame left device chunks on their way to the file
    // An asynchronous seal hands its metadata bytes (names, small-chunk buffers, index entries) to the pipeline's
    // writer thread instead of pwrite()ing them here: this thread would otherwise queue on the file's inode
    // lock behind every 16 MiB piece the writer is busy with (measured: 4 ms per frame of a back-to-back run of
    // 1 M-particle frames, in a call whose point is not to wait for the file)
    bool meta_async = false;
    // A write of THIS rank's rows failed in pgsd_write_chunk.  The call returned the error at once
    // (as the reference does, pgsd.c:2229-2236), but per-particle chunks involve no collective, so
    // the other ranks learn of it at the next flush: its status exchange reports it on every rank.
    int sticky_rc = PGSD_SUCCESS;
    int sticky_errno = 0;
    WriterPool* pool = nullptr;
    DevicePipeline* dev = nullptr;
    pgsd_device_config devcfg;
    bool devcfg_set = false;
    // frame-batched exchange: chunk writes that do not need their file offset at once are queued and
    // ONE allgather per frame (at pgsd_end_frame) carries their sizes and the ranks' status
    bool batch = false;
    bool local_reads = false; // pgsd_set_local_reads: reads drain this rank's own copies only, no collective flush
    bool defer_rows = false; // batched: host rows of all == true chunks stay valid until the exchange (pgsd_set_deferred_rows)
    bool unsynced = false; // a batched frame was sealed that no barrier between the ranks has covered yet
    std::vector<Queued> queue;
    // declared partition (pgsd_set_partition): every rank's row count is known, chunk writes exchange nothing
    std::vector<uint64_t> partition;
    bool have_partition = false;
    bool poisoned = false; // a call failed on this rank where the other ranks went on: this rank stops writing
    std::map<uint64_t, EarlyStage> early; // tickets of pgsd_stage_chunks_device not fully written yet
    uint64_t next_early = 1;

    bool v1() const
        {
        return header.pgsd_version < make_version(2, 0);
        }

    WriterPool* get_pool()
        {
        if (!pool)
            {
            unsigned n = 1; // one file = one inode lock: more writers only contend
            if (const char* e = getenv("PGSD_WRITERS"))
                n = (unsigned)atoi(e);
            if (devcfg_set && devcfg.n_writers)
                n = devcfg.n_writers;
            pool = writer_pool_create(n);
            }
        return pool;
        }

    uint64_t n_collectives = 0; // allgathers / barriers this handle has issued (pgsd_exchange_stats.collectives)
    // wall time of those allgathers as the calling thread sees it (pgsd_get_exchange_stats): transport
    // latency plus the wait for the slowest rank to arrive
    uint64_t exch_count = 0;
    double exch_us_sum = 0, exch_us_max = 0, exch_us_min = 0;

    int gather(const void* send, void* recv, size_t bytes)
        {
        n_collectives++;
        const auto t0 = std::chrono::steady_clock::now();
        const uint64_t serial = last_error_serial();
        const int rc = comm.allgather(comm.ctx, send, recv, bytes);
        if (rc != 0) // what the back end said (a rank that is gone, ranks out of step, an exchange that timed out) is kept
            set_last_error(last_error_serial() != serial ? std::string("communicator allgather failed: ") + last_error()
                                                         : std::string("communicator allgather failed"));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        exch_count++;
        exch_us_sum += us;
        if (us > exch_us_max)
            exch_us_max = us;
        if (exch_count == 1 || us < exch_us_min)
            exch_us_min = us;
        return rc;
        }

    // The end of the file as it WILL be once everything this rank has written -- or handed to its pipeline to be
    // written -- is in place: the largest offset + size of any write this handle issued, starting from the file's size
    // at open.  The maximum over the ranks is what MPI_File_get_size returns after all of them have finished
    // (pgsd.c:1015), known without waiting for a byte: pgsd_expand_file_index needs no drain (round 5).
    long long placed_end = 0;
    void note_placed(long long offset, uint64_t bytes)
        {
        if (bytes > 0 && offset >= 0 && offset + (long long)bytes > placed_end)
            placed_end = offset + (long long)bytes;
        }

    int allgather_u64(uint64_t v, std::vector<uint64_t>& out)
        {
        out.assign((size_t)P, 0);
        if (P == 1)
            {
            out[0] = v;
            return PGSD_SUCCESS;
            }
        if (gather(&v, out.data(), sizeof(uint64_t)) != 0)
            return PGSD_ERROR_COMM;
        return PGSD_SUCCESS;
        }
    };

inline Impl* impl_of(pgsd_handle* h)
    {
    return h ? (Impl*)h->impl : nullptr;
    }

// refresh the caller-visible mirror (the reference exposes its state directly, pgsd.h:297-353)
inline void publish(pgsd_handle* h, Impl* s)
    {
    h->fd = s->fd;
    h->header = s->header;
    h->file_index.data = s->file_index.data();
    h->file_index.size = s->file_index_size;
    h->file_index.reserved = s->file_index.size();
    h->file_names.data.data = s->file_names.d.data();
    h->file_names.data.size = s->file_names.size;
    h->file_names.data.reserved = s->file_names.reserved();
    h->file_names.n_names = s->file_n_names;
    h->cur_frame = s->cur_frame;
    h->file_size = s->file_size;
    h->open_flags = s->flags;
    h->pending_index_entries = s->pending;
    h->maximum_write_buffer_size = s->maxbuf;
    h->index_entries_to_buffer = s->idxbuf;
    h->rank = s->rank;
    h->nprocs = s->P;
    }

// Where a chunk's bytes go, decided exactly as pgsd_write_chunk decides (pgsd.c:2143-2256).
struct Placement
    {
    bool buffered;         // append to this rank's small-chunk buffer
    bool write;            // this rank writes bytes in the direct path
    long long file_offset; // direct path: where this rank's rows start
    size_t size;           // bytes of this rank
    };

// ---- shared between the units (definitions: the unit named in the comment)
// pgsd_container.cpp
int cmp_entry(const pgsd_index_entry& a, const pgsd_index_entry& b);
void sort_index(std::vector<pgsd_index_entry>& v);
int agree_status(Impl* s, int local_rc, bool check_state = false);
int initialize_file(int fd, const char* application, const char* schema, uint32_t schema_version);
size_t used_entries(const std::vector<pgsd_index_entry>& v);
bool entry_valid(const Impl* s, const pgsd_index_entry& e);
int initialize_handle(Impl* s);
void destroy_impl(Impl* s);
Impl* new_impl(const pgsd_comm* on = nullptr);
int meta_pwrite(Impl* s, const void* buf, size_t n, long long offset);
int flush_name_buffer(Impl* s);
int flush_write_buffer(Impl* s);
int expand_file_index(Impl* s, size_t size_required, int* local_rc);
int do_flush(Impl* s, bool async = false, bool sync_point = true);
bool metadata_pending(const Impl* s);
int drain_own_copies(Impl* s);
void release_early(Impl* s);
int do_end_frame(Impl* s, bool async = false);
// pgsd_placement.cpp
int name_to_id(Impl* s, const char* name, uint16_t* id);
int check_chunk_args(const Impl* s, const char* name, uint64_t N, uint32_t M, uint8_t flags, bool have_data);
int exchange_counts(Impl* s, uint64_t mine, int local_rc, std::vector<uint64_t>& all);
void auto_partition(const Impl* s, const std::vector<uint64_t>& sizes, uint64_t unit, uint32_t M, uint64_t* N_global,
    uint64_t* offset_elems);
int place_chunk(Impl* s, const char* name, uint32_t type, uint64_t N, uint32_t M, uint64_t N_global,
    uint32_t M_global, uint64_t offset, bool all, const std::vector<uint64_t>& sizes, Placement* pl);
void remember_failure(Impl* s, int rc, int err);
int ensure_device(Impl* s);
int deliver_chunk(Impl* s, Queued& q, const Placement& pl, bool skip);
int resolve_queue(Impl* s);
int trusted_place(Impl* s, const char* name, uint32_t type, uint64_t N, uint32_t M, uint64_t* N_global,
    uint32_t M_global, uint64_t* offset, bool all, int local, Placement* pl, bool* deliver);
// pgsd_read.cpp
int flush_for_lookup(Impl* s);
int flush_for_read(Impl* s);
    } // namespace pgsd_amd

#endif
