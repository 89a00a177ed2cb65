// pgsd_file.cpp -- the GSD v2 container writer/reader behind the pgsd.h C ABI.
//
// Produces, for the same sequence of calls on the same particle partition, exactly the
// bytes the reference's MPI-IO implementation produces (/root/reference/pgsd/pgsd/pgsd.c;
// the functions below cite the lines whose on-disk effect they reproduce).  The protocol
// between ranks is different by design:
//
//   reference                                   here
//   ---------                                   ----
//   names / index / buffer_index on rank 0      replicated on every rank (same calls in the
//   only, scalars re-broadcast after each step    same order => same state), rank 0 writes them
//   4 barriers + 3 allreduces + 2 bcasts per    one 16-byte allgather per chunk: every rank's byte
//   chunk (pgsd.c:2143-2257)                      count (-> max, sum, buffer sizes) + its local status;
//                                                 pgsd_write_chunks_device: one for all its chunks
//   ~15 collectives per pgsd_flush              one status allgather (+ one EOF exchange when
//                                                 the on-disk index is relocated)
//   MPI_File_write_at                           pwrite at the identical offset, split over a
//                                                 writer pool; device chunks arrive through
//                                                 the HIP pipeline (pgsd_device.cpp)
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
static const uint64_t INDEX_COPY_ENTRIES = 256ull * 1024;

static_assert(sizeof(pgsd_header) == 256, "GSD header is 256 bytes on disk");
static_assert(sizeof(pgsd_index_entry) == 32, "GSD index entry is 32 bytes on disk");

size_t sizeof_type(uint32_t type)
    {
    static const size_t s[] = {0, 1, 2, 4, 8, 1, 2, 4, 8, 4, 8}; // pgsd.c:2539-2555
    return (type >= 1 && type <= 10) ? s[type] : 0;
    }

static uint32_t make_version(unsigned major, unsigned minor)
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

static int cmp_entry(const pgsd_index_entry& a, const pgsd_index_entry& b)
    {
    // pgsd.c:799-833
    if (a.frame < b.frame)
        return -1;
    if (a.frame > b.frame)
        return 1;
    if (a.id < b.id)
        return -1;
    if (a.id > b.id)
        return 1;
    return 0;
    }

// In-place heap sort in the reference's exact order of swaps (pgsd.c:839-953): the sort is
// not stable, so the position of entries with equal (frame, id) is part of the file bytes.
static void sift_down(std::vector<pgsd_index_entry>& v, size_t start, size_t end)
    {
    size_t root = start;
    while (2 * root + 1 <= end)
        {
        size_t child = 2 * root + 1;
        size_t sw = root;
        if (cmp_entry(v[sw], v[child]) < 0)
            sw = child;
        if (child + 1 <= end && cmp_entry(v[sw], v[child + 1]) < 0)
            sw = child + 1;
        if (sw == root)
            return;
        std::swap(v[root], v[sw]);
        root = sw;
        }
    }

static void sort_index(std::vector<pgsd_index_entry>& v)
    {
    if (v.size() <= 1)
        return;
    for (ssize_t start = (ssize_t)((v.size() - 2) / 2); start >= 0; start--)
        sift_down(v, (size_t)start, v.size() - 1);
    for (size_t end = v.size() - 1; end > 0;)
        {
        std::swap(v[end], v[0]);
        end--;
        sift_down(v, 0, end);
        }
    }

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

    uint64_t n_collectives = 0; // allgathers / barriers this handle has issued (pgsd_get_collective_count)
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

static Impl* impl_of(pgsd_handle* h)
    {
    return h ? (Impl*)h->impl : nullptr;
    }

// refresh the caller-visible mirror (the reference exposes its state directly, pgsd.h:297-353)
static void publish(pgsd_handle* h, Impl* s)
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

// every rank learns the first non-zero status (rank order) and its errno.  check_state (the flush's exchange): the
// ranks also compare what they believe about the file -- its size, the frame counter, the number of names and of index
// entries.  The metadata is replicated, not broadcast (the reference lets rank 0's view win, pgsd.c:2219-2222): it
// stays identical as long as every rank makes the same calls with sizes that agree -- which the exchanges check, and a
// DECLARED partition (pgsd_set_partition) takes on trust for chunks that are not partitioned.  A caller
// that broke that trust is told here, on every rank, instead of leaving ranks with different layouts of one file.
static int agree_status(Impl* s, int local_rc, bool check_state = false)
    {
    if (s->P == 1)
        return local_rc;
    uint64_t mine[5] = {(uint64_t)(uint32_t)local_rc | ((uint64_t)(uint32_t)(local_rc ? errno : 0) << 32), (uint64_t)s->file_size,
                        s->cur_frame, s->file_n_names, s->file_index_size};
    std::vector<uint64_t> all((size_t)s->P * 5);
    if (s->gather(mine, all.data(), sizeof(mine)) != 0)
        return PGSD_ERROR_COMM;
    for (int r = 0; r < s->P; r++)
        if ((int32_t)(uint32_t)all[(size_t)r * 5] != 0)
            {
            if (local_rc == 0)
                errno = (int)(uint32_t)(all[(size_t)r * 5] >> 32);
            return (int32_t)(uint32_t)all[(size_t)r * 5];
            }
    for (int r = 1; r < s->P && check_state; r++)
        for (int k = 1; k < 5; k++)
            if (all[(size_t)r * 5 + k] != all[(size_t)k])
                {
                static const char* what[5] = {"", "file size", "frame counter", "number of names", "number of index entries"};
                set_last_error(std::string("the ranks disagree about the file (") + what[k] + ": "
                               + std::to_string(all[(size_t)k]) + " on rank 0, " + std::to_string(all[(size_t)r * 5 + k])
                               + " on rank " + std::to_string(r)
                               + "): a chunk that is not partitioned was written with different sizes on different ranks");
                return PGSD_ERROR_COMM;
                }
    return PGSD_SUCCESS;
    }

// fresh-file skeleton, pgsd.c:1414-1474 (rank 0 only)
static int initialize_file(int fd, const char* application, const char* schema, uint32_t schema_version)
    {
    if (ftruncate(fd, 0) != 0)
        return PGSD_ERROR_IO;
    std::vector<char> img(sizeof(pgsd_header) + INITIAL_INDEX_SIZE * sizeof(pgsd_index_entry)
                              + INITIAL_NAME_BUFFER_SIZE,
                          0);
    pgsd_header* hd = (pgsd_header*)img.data();
    hd->magic = MAGIC_ID;
    hd->pgsd_version = make_version(CURRENT_FILE_VERSION, 0);
    strncpy(hd->application, application, sizeof(hd->application) - 1);
    strncpy(hd->schema, schema, sizeof(hd->schema) - 1);
    hd->schema_version = schema_version;
    hd->index_location = sizeof(pgsd_header);
    hd->index_allocated_entries = INITIAL_INDEX_SIZE;
    hd->namelist_location = hd->index_location + sizeof(pgsd_index_entry) * hd->index_allocated_entries;
    hd->namelist_allocated_entries = INITIAL_NAME_BUFFER_SIZE / PGSD_NAME_SIZE;
    return pwrite_full(fd, img.data(), img.size(), 0) == 0 ? PGSD_SUCCESS : PGSD_ERROR_IO;
    }

// number of used entries of an index block = first entry with location == 0
// (binary search of pgsd.c:661-704; validity checks are applied by the caller on open)
static size_t used_entries(const std::vector<pgsd_index_entry>& v)
    {
    if (v.empty() || v[0].location == 0)
        return 0;
    size_t L = 0, R = v.size();
    do
        {
        size_t m = (L + R) / 2;
        if (v[m].location != 0)
            L = m;
        else
            R = m;
        } while ((R - L) > 1);
    return R;
    }

static bool entry_valid(const Impl* s, const pgsd_index_entry& e)
    {
    // pgsd.c:414-450
    if (sizeof_type(e.type) == 0)
        return false;
    // as pgsd.c:421-425, in arithmetic that a damaged entry cannot wrap around
    if (e.location < 0 || e.location > s->file_size)
        return false;
    const unsigned __int128 size = (unsigned __int128)e.N * e.M * sizeof_type(e.type);
    if (size > (unsigned __int128)(s->file_size - e.location))
        return false;
    if (e.frame >= s->header.index_allocated_entries)
        return false;
    if (e.id >= (s->file_n_names + s->frame_n_names))
        return false;
    if (e.flags != 0)
        return false;
    return true;
    }

// pgsd.c:1484-1703, executed by every rank (the reference parses names/index on rank 0 only)
static int initialize_handle(Impl* s)
    {
    memset(&s->header, 0, sizeof(s->header));
    pread_some(s->fd, &s->header, sizeof(s->header), 0);
    if (s->header.magic != MAGIC_ID)
        return PGSD_ERROR_NOT_A_PGSD_FILE;
    if (s->header.pgsd_version < make_version(1, 0) && s->header.pgsd_version != make_version(0, 3))
        return PGSD_ERROR_INVALID_PGSD_FILE_VERSION;
    if (s->header.pgsd_version >= make_version(3, 0))
        return PGSD_ERROR_INVALID_PGSD_FILE_VERSION;

    struct stat st;
    if (fstat(s->fd, &st) != 0)
        return PGSD_ERROR_IO;
    s->file_size = (long long)st.st_size;

    // pgsd.c:1558-1562; the products are formed so that a damaged header cannot wrap them around
    if (s->header.namelist_location > (uint64_t)s->file_size
        || s->header.namelist_allocated_entries > ((uint64_t)s->file_size - s->header.namelist_location) / PGSD_NAME_SIZE)
        return PGSD_ERROR_FILE_CORRUPT;

    // name list
    size_t namelist_n_bytes = PGSD_NAME_SIZE * s->header.namelist_allocated_entries;
    if (namelist_n_bytes == 0)
        return PGSD_ERROR_FILE_CORRUPT;
    s->file_names.allocate(namelist_n_bytes);
    pread_some(s->fd, s->file_names.d.data(), namelist_n_bytes, (long long)s->header.namelist_location);
    if (s->file_names.d[namelist_n_bytes - 1] != 0)
        return PGSD_ERROR_FILE_CORRUPT;
    size_t name_start = 0;
    s->file_n_names = 0;
    s->name_map.clear();
    while (name_start < namelist_n_bytes)
        {
        const char* name = s->file_names.d.data() + name_start;
        if (name[0] == 0)
            break;
        // first occurrence wins, like the chained hash map's lookup order (pgsd.c:374-405)
        s->name_map.emplace(std::string(name), (uint16_t)s->file_n_names);
        s->file_n_names++;
        if (s->v1())
            name_start += PGSD_NAME_SIZE;
        else
            name_start += strnlen(name, namelist_n_bytes - name_start) + 1;
        }
    s->file_names.size = name_start;

    // index block, pgsd.c:602-707
    if (s->header.index_location > (uint64_t)s->file_size
        || s->header.index_allocated_entries
               > ((uint64_t)s->file_size - s->header.index_location) / sizeof(pgsd_index_entry))
        return PGSD_ERROR_FILE_CORRUPT;
    if (s->header.index_allocated_entries == 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    s->file_index.assign(s->header.index_allocated_entries, pgsd_index_entry());
    memset(s->file_index.data(), 0, s->file_index.size() * sizeof(pgsd_index_entry));
    pread_some(s->fd, s->file_index.data(), sizeof(pgsd_index_entry) * s->file_index.size(),
               (long long)s->header.index_location);
    if (s->file_index[0].location != 0 && !entry_valid(s, s->file_index[0]))
        return PGSD_ERROR_FILE_CORRUPT;
    if (s->file_index[0].location == 0)
        s->file_index_size = 0;
    else
        {
        size_t L = 0, R = s->file_index.size();
        do
            {
            size_t m = (L + R) / 2;
            if (s->file_index[m].location != 0
                && (!entry_valid(s, s->file_index[m]) || s->file_index[m].frame < s->file_index[L].frame))
                return PGSD_ERROR_FILE_CORRUPT;
            if (s->file_index[m].location != 0)
                L = m;
            else
                R = m;
            } while ((R - L) > 1);
        s->file_index_size = R;
        }

    s->cur_frame = s->file_index_size == 0 ? 0 : s->file_index[s->file_index_size - 1].frame + 1;

    s->frame_index.clear();
    s->buffer_index.clear();
    s->write_buffer.clear();
    s->wb_sizes.assign((size_t)s->P, 0);
    s->frame_n_names = 0;
    if (s->flags != PGSD_OPEN_READONLY)
        s->frame_names.allocate(INITIAL_FRAME_NAMES_SIZE);
    s->pending = 0;
    s->maxbuf = DEFAULT_MAXIMUM_WRITE_BUFFER_SIZE;
    s->idxbuf = DEFAULT_INDEX_ENTRIES_TO_BUFFER;
    return PGSD_SUCCESS;
    }

static void destroy_impl(Impl* s)
    {
    if (!s)
        return;
    if (s->dev)
        device_pipeline_destroy(s->dev);
    if (s->pool)
        writer_pool_destroy(s->pool);
    if (s->fd >= 0)
        close(s->fd);
    delete s;
    }

static Impl* new_impl(const pgsd_comm* on = nullptr)
    {
    Impl* s = new Impl;
    if (on)
        {
        // a communicator of the caller's (pgsd_create_and_open_on): copied, never destroyed by the handle
        pgsd_comm c = *on;
        c.destroy = nullptr;
        s->comm_box = std::make_shared<CommBox>(c);
        }
    else
        s->comm_box = default_comm_box();
    s->comm = s->comm_box->c;
    s->rank = s->comm.rank;
    s->P = s->comm.size;
    memset(&s->header, 0, sizeof(s->header));
    memset(&s->devcfg, 0, sizeof(s->devcfg));
    return s;
    }

// metadata bytes to the file: at once, or -- during an asynchronous seal -- through the pipeline's writer
// thread, behind the frame's data in the same FIFO (a failure then surfaces like a device chunk's: at the next
// drain, on every rank at the next flush)
static int meta_pwrite(Impl* s, const void* buf, size_t n, long long offset)
    {
    if (s->meta_async && s->dev)
        {
        device_pipeline_write_host(s->dev, buf, n, offset);
        return 0;
        }
    return pwrite_full(s->fd, buf, n, offset);
    }

// pgsd_flush_name_buffer, pgsd.c:1216-1319
static int flush_name_buffer(Impl* s)
    {
    if (s->frame_n_names == 0)
        return PGSD_SUCCESS;
    if (s->frame_names.size == 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    size_t old_reserved = s->file_names.reserved();
    size_t old_size = s->file_names.size;
    s->file_names.append(s->frame_names.d.data(), s->frame_names.size);
    s->file_n_names += s->frame_n_names;
    s->frame_n_names = 0;
    s->frame_names.size = 0;
    std::fill(s->frame_names.d.begin(), s->frame_names.d.end(), 0);
    if (s->file_names.reserved() % PGSD_NAME_SIZE != 0)
        return PGSD_ERROR_INVALID_ARGUMENT;

    int rc = PGSD_SUCCESS;
    if (s->file_names.reserved() > old_reserved)
        {
        // the list outgrew its block: append the whole list at the end of the file and
        // point the header at it, pgsd.c:1284-1300
        long long offset = s->file_size;
        s->file_size += (long long)s->file_names.reserved();
        s->header.namelist_location = (uint64_t)offset;
        s->header.namelist_allocated_entries = s->file_names.reserved() / PGSD_NAME_SIZE;
        if (s->rank == 0)
            {
            if (meta_pwrite(s, s->file_names.d.data(), s->file_names.reserved(), offset) != 0
                || meta_pwrite(s, &s->header, sizeof(s->header), 0) != 0)
                rc = PGSD_ERROR_IO;
            }
        }
    else if (s->rank == 0)
        {
        // in place: rewrite [old_size, reserved), pgsd.c:1304-1306
        if (meta_pwrite(s, s->file_names.d.data() + old_size, s->file_names.reserved() - old_size,
                        (long long)s->header.namelist_location + (long long)old_size)
            != 0)
            rc = PGSD_ERROR_IO;
        }
    return rc;
    }

// pgsd_flush_write_buffer, pgsd.c:1108-1201.  The MPI_Allgather of buffer sizes
// (pgsd.c:1126) is not needed: wb_sizes already holds every rank's size.
static int flush_write_buffer(Impl* s)
    {
    uint64_t total = 0;
    for (uint64_t b : s->wb_sizes)
        total += b;
    if (total == 0 && s->buffer_index.empty())
        return PGSD_SUCCESS;
    if (s->wb_sizes[0] > 0 && s->buffer_index.empty())
        return PGSD_ERROR_INVALID_ARGUMENT; // pgsd.c:1135-1143

    // rank r's copy lands at file_size + sum_{j<r} size_j (pgsd.c:1145-1154).  Every rank
    // buffered its own copy of the replicated chunks, so a P-rank file holds P copies and
    // the index points at rank 0's -- kept, because the file must match byte for byte.
    long long offset_root = s->file_size;
    long long offset = s->file_size;
    for (int j = 0; j < s->rank; j++)
        offset += (long long)s->wb_sizes[(size_t)j];
    int rc = PGSD_SUCCESS;
    if (!s->write_buffer.empty())
        if (meta_pwrite(s, s->write_buffer.data(), s->write_buffer.size(), offset) != 0)
            rc = PGSD_ERROR_IO;
    s->write_buffer.clear();
    std::fill(s->wb_sizes.begin(), s->wb_sizes.end(), 0);
    s->file_size += (long long)total;

    for (const pgsd_index_entry& e : s->buffer_index)
        {
        s->frame_index.push_back(e);
        s->frame_index.back().location += offset_root; // pgsd.c:1191-1192
        }
    s->buffer_index.clear();
    return rc;
    }

// pgsd_expand_file_index, pgsd.c:965-1091
static int expand_file_index(Impl* s, size_t size_required, int* local_rc)
    {
    size_t size_old = s->header.index_allocated_entries;
    size_t size_new = size_old * 2;
    while (size_new <= size_required)
        size_new *= 2;

    // The new block goes to the TRUE end of the file as rank 0 sees it
    // (MPI_File_get_size, pgsd.c:1015) once every rank's data is in the file.
    s->n_collectives++;
    int brc = comm_barrier(s->comm);
    if (brc != PGSD_SUCCESS)
        return brc;
    uint64_t eof = 0;
    if (s->rank == 0)
        {
        struct stat st;
        if (fstat(s->fd, &st) != 0)
            *local_rc = PGSD_ERROR_IO;
        eof = (uint64_t)st.st_size;
        }
    std::vector<uint64_t> all;
    int rc = s->allgather_u64(eof, all);
    if (rc != PGSD_SUCCESS)
        return rc;
    long long new_loc = (long long)all[0];
    long long old_loc = (long long)s->header.index_location;
    size_t old_bytes = size_old * sizeof(pgsd_index_entry);
    size_t new_bytes = size_new * sizeof(pgsd_index_entry);

    if (s->rank == 0)
        {
        // copy the old block in pieces, then zero-fill (pgsd.c:1021-1062)
        size_t piece = INDEX_COPY_ENTRIES * sizeof(pgsd_index_entry);
        if (piece > old_bytes)
            piece = old_bytes;
        std::vector<char> buf(piece);
        size_t done = 0;
        while (done < old_bytes)
            {
            size_t n = old_bytes - done < piece ? old_bytes - done : piece;
            pread_some(s->fd, buf.data(), n, old_loc + (long long)done);
            if (pwrite_full(s->fd, buf.data(), n, new_loc + (long long)done) != 0)
                *local_rc = PGSD_ERROR_IO;
            done += n;
            }
        std::fill(buf.begin(), buf.end(), 0);
        while (done < new_bytes)
            {
            size_t n = new_bytes - done < piece ? new_bytes - done : piece;
            if (pwrite_full(s->fd, buf.data(), n, new_loc + (long long)done) != 0)
                *local_rc = PGSD_ERROR_IO;
            done += n;
            }
        }
    s->header.index_location = (uint64_t)new_loc;
    s->file_size = new_loc + (long long)new_bytes;
    s->header.index_allocated_entries = size_new;
    if (s->rank == 0)
        if (pwrite_full(s->fd, &s->header, sizeof(s->header), 0) != 0)
            *local_rc = PGSD_ERROR_IO;

    // the in-memory mirror is the old block plus zeros; its used size is found the way
    // pgsd_index_buffer_map finds it after re-reading (pgsd.c:661-704, 1083)
    pgsd_index_entry zero;
    memset(&zero, 0, sizeof(zero));
    s->file_index.resize(size_new, zero);
    s->file_index_size = used_entries(s->file_index);
    return PGSD_SUCCESS;
    }

static int resolve_queue(Impl* s);
static void remember_failure(Impl* s, int rc, int err);

// pgsd_flush, pgsd.c:1955-2070.  sync_point: the call must leave every rank's bytes of the sealed
// frames in the file and every rank with the same verdict (pgsd_flush, pgsd_close, reads; and
// pgsd_end_frame unless the frame exchange is batched).
static int do_flush(Impl* s, bool async = false, bool sync_point = true)
    {
    if (s->flags == PGSD_OPEN_READONLY)
        return PGSD_ERROR_FILE_MUST_BE_WRITABLE;

    // chunks still waiting for their placement are placed first (one exchange)
    const int qrc = s->queue.empty() ? PGSD_SUCCESS : resolve_queue(s);

    // Replicated state tells every rank alike whether there is anything to do.
    bool work = s->frame_n_names > 0 || !s->buffer_index.empty() || !s->frame_index.empty()
                || s->dirty_data || s->inflight || (sync_point && s->unsynced);
    for (uint64_t b : s->wb_sizes)
        work = work || b > 0;
    if (!work)
        return qrc;

    // Asynchronous sealing commits the metadata now and lets the device chunks finish in
    // the background -- unless the on-disk index must move, which needs the file's true end
    // and therefore every byte of every rank in place (decided alike on all ranks).
    if (async && s->pending <= s->frame_index.size())
        {
        uint64_t will_write = s->frame_index.size() + s->buffer_index.size() - s->pending;
        if (s->file_index_size + will_write > s->file_index.size())
            async = false;
        }

    int local_rc = PGSD_SUCCESS;
    int sticky_errno = 0;
    s->meta_async = false;
    // device chunks of this rank must be in the file before the frame is sealed
    if (s->dev && async)
        {
        s->inflight = true;
        device_pipeline_kick(s->dev);
        s->meta_async = device_pipeline_single_writer(s->dev); // FIFO order of the writes needs ONE writer thread
        }
    else if (s->dev)
        {
        s->inflight = false;
        std::string err;
        int drc = device_pipeline_drain(s->dev, &err);
        if (drc != PGSD_SUCCESS)
            {
            set_last_error(err);
            local_rc = drc;
            if (drc == PGSD_ERROR_IO)
                sticky_errno = errno; // of the pipeline's writer thread
            }
        }
    if (s->sticky_rc != PGSD_SUCCESS)
        {
        if (local_rc == PGSD_SUCCESS)
            {
            local_rc = s->sticky_rc;
            sticky_errno = s->sticky_errno;
            }
        s->sticky_rc = PGSD_SUCCESS;
        }
    int rc = flush_name_buffer(s);
    if (rc != PGSD_SUCCESS && local_rc == PGSD_SUCCESS)
        local_rc = rc;
    rc = flush_write_buffer(s);
    if (rc != PGSD_SUCCESS && local_rc == PGSD_SUCCESS)
        local_rc = rc;

    if (s->pending > s->frame_index.size())
        {
        if (local_rc == PGSD_SUCCESS)
            local_rc = PGSD_ERROR_INVALID_ARGUMENT;
        }
    else
        {
        uint64_t to_write = s->frame_index.size() - s->pending;
        if (to_write > 0)
            {
            if ((s->file_index_size + to_write) > s->file_index.size())
                {
                int erc = expand_file_index(s, s->file_index_size + to_write, &local_rc);
                if (erc != PGSD_SUCCESS)
                    return erc; // communicator failure: nothing sane left to agree on
                }
            sort_index(s->frame_index);
            long long write_pos = (long long)s->header.index_location
                                  + (long long)(sizeof(pgsd_index_entry) * s->file_index_size);
            // all frame_index entries are written, the pending ones of an open frame
            // included (pgsd.c:2032); they are overwritten by the next flush
            if (s->rank == 0)
                if (meta_pwrite(s, s->frame_index.data(),
                                sizeof(pgsd_index_entry) * s->frame_index.size(), write_pos)
                    != 0)
                    local_rc = PGSD_ERROR_IO;
            size_t room = s->file_index.size() - s->file_index_size;
            size_t ncopy = s->frame_index.size() < room ? s->frame_index.size() : room;
            memcpy(s->file_index.data() + s->file_index_size, s->frame_index.data(),
                   sizeof(pgsd_index_entry) * ncopy);
            s->file_index_size += to_write;

            // keep the entries of the open frame: every kept slot receives the first
            // pending entry (the reference copies without "+ i", pgsd.c:2049-2057)
            if (s->pending > 0)
                {
                pgsd_index_entry first = s->frame_index[s->frame_index.size() - s->pending];
                for (uint64_t i = 0; i < s->pending; i++)
                    s->frame_index[i] = first;
                }
            s->frame_index.resize(s->pending);
            }
        }
    s->dirty_data = false;
    s->meta_async = false;
    if (sticky_errno)
        errno = sticky_errno;
    if ((s->batch || s->have_partition) && !sync_point && s->P > 1)
        {
        // batched frame exchange: no second collective per frame.  This rank's verdict is returned now
        // and travels to the others with the next exchange; the barrier that guarantees every rank's
        // rows are in the file is made up at the next synchronisation point.
        if (local_rc != PGSD_SUCCESS)
            remember_failure(s, local_rc, errno);
        s->unsynced = true;
        return local_rc != PGSD_SUCCESS ? local_rc : qrc;
        }
    s->unsynced = false;
    const int arc = agree_status(s, local_rc, true);
    return arc != PGSD_SUCCESS ? arc : qrc;
    }

// What a LOOKUP (pgsd_find_chunk, pgsd_find_matching_chunk_name) needs from the flush the reference runs
// first (pgsd.c:2316, 2586): the replicated index and name list must hold everything written so far.  Whether
// they do is a matter of replicated state, so every rank decides alike.  When they do, nothing is left but
// the barrier a batched frame still owes (`unsynced`) or this rank's own asynchronous copies (`inflight`),
// neither of which a lookup needs: it stays LOCAL then -- a caller may look chunks up on one rank only
// (HOOMDTrajectory._should_write did, from its third frame on: ADVICE r2) without leaving the others
// outside a collective.  Reads still flush in full: they need the other ranks' bytes in the file.
static bool metadata_pending(const Impl* s)
    {
    bool work = !s->queue.empty() || s->frame_n_names > 0 || !s->buffer_index.empty() || !s->frame_index.empty()
                || s->dirty_data;
    for (uint64_t b : s->wb_sizes)
        work = work || b > 0;
    return work;
    }

// local: this rank's rows of the asynchronously sealed frames reach the file (documented in pgsd.h)
static int drain_own_copies(Impl* s)
    {
    if (s->dev && s->inflight)
        {
        s->inflight = false;
        std::string err;
        const int drc = device_pipeline_drain(s->dev, &err);
        if (drc != PGSD_SUCCESS)
            {
            set_last_error(err);
            remember_failure(s, drc, drc == PGSD_ERROR_IO ? errno : 0); // every rank hears of it at the next flush
            return drc;
            }
        }
    return PGSD_SUCCESS;
    }

static int flush_for_lookup(Impl* s)
    {
    if (s->flags == PGSD_OPEN_READONLY)
        return PGSD_SUCCESS;
    // pgsd_set_local_reads covers the lookup that precedes the read: a frame that wrote buffered small
    // chunks only -- everything else elided -- leaves metadata pending behind pgsd_end_frame (pgsd.c:1941-1950), and
    // a rank that then looks one of frame 0's chunks up ALONE (it is the only one that compares that array) must not
    // start the collective flush.  It sees what the last flush committed; chunks of frames still pending are
    // not found until then.  (One rank: the flush is nobody else's business and runs as ever.)
    if (metadata_pending(s) && (!s->local_reads || s->P == 1))
        return do_flush(s);
    return drain_own_copies(s);
    }

// What a READ needs before it touches the file: the reference's flush (collective) -- or, with
// pgsd_set_local_reads, only this rank's own asynchronous copies in place.
static int flush_for_read(Impl* s)
    {
    if (s->flags == PGSD_OPEN_READONLY)
        return PGSD_SUCCESS;
    if (!s->local_reads)
        return do_flush(s);
    return drain_own_copies(s);
    }

// chunks that were staged ahead (pgsd_stage_chunks_device) and never written: their packed bytes go nowhere
static void release_early(Impl* s)
    {
    for (auto& kv : s->early)
        {
        EarlyStage& e = kv.second;
        for (size_t i = 0; i < e.claimed.size() && e.ticket >= 0 && s->dev; i++)
            if (!e.claimed[i])
                (void)device_pipeline_commit(s->dev, e.ticket, i, -1, nullptr, nullptr);
        }
    s->early.clear();
    }

static int do_end_frame(Impl* s, bool async = false)
    {
    // pgsd.c:1916-1953
    if (s->flags == PGSD_OPEN_READONLY)
        return PGSD_ERROR_FILE_MUST_BE_WRITABLE;
    TraceRange tr("pgsd:end_frame frame=%llu", s->cur_frame);
    if (!s->early.empty())
        release_early(s);
    // queued chunks belong to the frame that is being sealed: place them before the counter moves
    const int qrc = s->queue.empty() ? PGSD_SUCCESS : resolve_queue(s);
    s->cur_frame++;
    s->pending = 0;
    int rc = PGSD_SUCCESS;
    if (!s->frame_index.empty() || s->buffer_index.size() > s->idxbuf)
        rc = do_flush(s, async, !(s->batch || s->have_partition));
    return qrc != PGSD_SUCCESS ? qrc : rc;
    }

// name -> id; new names get the next id in first-seen order (pgsd.c:2111-2133, 1340-1404)
static int name_to_id(Impl* s, const char* name, uint16_t* id)
    {
    auto it = s->name_map.find(name);
    if (it != s->name_map.end())
        {
        *id = it->second;
        return PGSD_SUCCESS;
        }
    if (s->flags == PGSD_OPEN_READONLY)
        return PGSD_ERROR_FILE_MUST_BE_WRITABLE;
    if (s->file_n_names + s->frame_n_names == UINT16_MAX)
        return PGSD_ERROR_NAMELIST_FULL;
    *id = (uint16_t)(s->file_n_names + s->frame_n_names);
    if (s->v1())
        {
        char name_v1[PGSD_NAME_SIZE];
        strncpy(name_v1, name, PGSD_NAME_SIZE - 1);
        name_v1[PGSD_NAME_SIZE - 1] = 0;
        s->frame_names.append(name_v1, PGSD_NAME_SIZE);
        s->name_map.emplace(std::string(name_v1), *id);
        }
    else
        {
        s->frame_names.append(name, strlen(name) + 1);
        s->name_map.emplace(std::string(name), *id);
        }
    s->frame_n_names++;
    return PGSD_SUCCESS;
    }

// Where a chunk's bytes go, decided exactly as pgsd_write_chunk decides (pgsd.c:2143-2256).
struct Placement
    {
    bool buffered;         // append to this rank's small-chunk buffer
    bool write;            // this rank writes bytes in the direct path
    long long file_offset; // direct path: where this rank's rows start
    size_t size;           // bytes of this rank
    };

// Local argument checks of pgsd_write_chunk, pgsd.c:2090-2105.  The reference returns from them
// before its first collective, which leaves the other ranks waiting; here the verdict travels
// with the size exchange below, so every rank returns the same error.
static int check_chunk_args(const Impl* s, const char* name, uint64_t N, uint32_t M, uint8_t flags, bool have_data)
    {
    if (N > 0 && !have_data)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (M == 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (s->flags == PGSD_OPEN_READONLY)
        return PGSD_ERROR_FILE_MUST_BE_WRITABLE;
    if (flags != 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!name)
        return PGSD_ERROR_INVALID_ARGUMENT;
    return PGSD_SUCCESS;
    }

// The one exchange of a chunk write: every rank's byte count (or row count) and the status of its
// local preparation.  The reference obtains max and sum of `size` with MPI_Allreduce MAX
// (pgsd.c:2157) and SUM (pgsd.c:2242) and never looks at the caller's global_size
// (pgsd.c:2147-2151 only scales it), so neither does this library: the file advances by what the
// ranks actually contribute.  Returns the first non-zero status in rank order.
static int exchange_counts(Impl* s, uint64_t mine, int local_rc, std::vector<uint64_t>& all)
    {
    all.assign((size_t)s->P, 0);
    if (s->P == 1)
        {
        all[0] = mine;
        return local_rc;
        }
    uint64_t send[2] = {mine, (uint64_t)(uint32_t)local_rc | ((uint64_t)(uint32_t)(local_rc ? errno : 0) << 32)};
    std::vector<uint64_t> recv((size_t)s->P * 2);
    if (s->gather(send, recv.data(), sizeof(send)) != 0)
        return PGSD_ERROR_COMM;
    int rc = PGSD_SUCCESS;
    for (int r = 0; r < s->P; r++)
        {
        all[(size_t)r] = recv[(size_t)r * 2];
        const int rrc = (int)(int32_t)(uint32_t)recv[(size_t)r * 2 + 1];
        if (rrc != 0 && rc == PGSD_SUCCESS)
            {
            rc = rrc;
            if (local_rc == 0)
                errno = (int)(uint32_t)(recv[(size_t)r * 2 + 1] >> 32);
            }
        }
    return rc;
    }

// PGSD_PARTITION_AUTO: the partition the reference's callers obtain with an MPI_Allgather of their own
// (benchmark-write.cc:39-45, fl.pyx:596-598) comes out of the size exchange instead.  sizes[r] / unit
// = rows of rank r; the global count and this rank's first element follow.
static void auto_partition(const Impl* s, const std::vector<uint64_t>& sizes, uint64_t unit, uint32_t M,
                           uint64_t* N_global, uint64_t* offset_elems)
    {
    uint64_t total = 0, before = 0;
    for (int r = 0; r < s->P; r++)
        {
        const uint64_t rows = unit ? sizes[(size_t)r] / unit : 0;
        if (r < s->rank)
            before += rows;
        total += rows;
        }
    *N_global = total;
    *offset_elems = before * M;
    }

// Decide where a chunk's bytes go exactly as pgsd_write_chunk decides (pgsd.c:2143-2256) and record
// its index entry.  `sizes` holds every rank's byte count of the chunk (exchange_counts).
static int place_chunk(Impl* s, const char* name, uint32_t type, uint64_t N, uint32_t M, uint64_t N_global,
                       uint32_t M_global, uint64_t offset, bool all, const std::vector<uint64_t>& sizes,
                       Placement* pl)
    {
    uint16_t id;
    int rc = name_to_id(s, name, &id);
    if (rc != PGSD_SUCCESS)
        return rc;

    pgsd_index_entry entry;
    memset(&entry, 0, sizeof(entry));
    entry.frame = s->cur_frame;
    entry.id = id;
    entry.type = (uint8_t)type;
    entry.N = N_global;
    entry.M = M_global;

    const size_t sz = sizeof_type(type);
    pl->size = (size_t)(N * M * sz);

    uint64_t maxsize = 0, sumsize = 0;
    for (uint64_t v : sizes)
        {
        if (v > maxsize)
            maxsize = v;
        sumsize += v;
        }

    if (maxsize < s->maxbuf && !all)
        {
        // BUFFERED, pgsd.c:2160-2202.  Flush first when the chunk does not fit any more.
        bool need_flush = false;
        for (int r = 0; r < s->P; r++)
            if (sizes[(size_t)r] > (s->maxbuf - s->wb_sizes[(size_t)r]))
                need_flush = true;
        if (need_flush)
            flush_write_buffer(s);
        entry.location = (int64_t)s->wb_sizes[0]; // offset inside rank 0's buffer
        s->buffer_index.push_back(entry);
        for (int r = 0; r < s->P; r++)
            s->wb_sizes[(size_t)r] += sizes[(size_t)r];
        pl->buffered = true;
        pl->write = false;
        pl->file_offset = -1;
        }
    else
        {
        // DIRECT, pgsd.c:2203-2250
        entry.location = s->file_size;
        s->frame_index.push_back(entry);
        pl->buffered = false;
        pl->write = all || s->rank == 0;
        pl->file_offset = s->file_size + (long long)(offset * sz);
        // file_size advances by the sum of the ranks' sizes (MPI_Allreduce SUM, pgsd.c:2240-2249):
        // also when only rank 0 wrote (all == false: the hole is part of the reference's layout)
        // and whatever the caller passed as global_size (replicated data written with all == true
        // and offset 0, fl.pyx's default arguments, advances the file by P copies)
        s->file_size += (long long)sumsize;
        s->dirty_data = true;
        }
    s->pending++;
    return PGSD_SUCCESS;
    }

// a chunk this rank could not deliver after its placement was committed: the index entry exists on
// every rank, so the failure is reported by all of them at the next flush (agree_status)
static void remember_failure(Impl* s, int rc, int err)
    {
    if (s->sticky_rc == PGSD_SUCCESS)
        {
        s->sticky_rc = rc;
        s->sticky_errno = err;
        }
    }

static int ensure_device(Impl* s)
    {
    if (s->dev)
        return PGSD_SUCCESS;
    pgsd_device_config cfg = s->devcfg;
    if (!s->devcfg_set)
        {
        memset(&cfg, 0, sizeof(cfg));
        cfg.device = -1;
        }
    std::string err;
    s->dev = device_pipeline_create(cfg, s->fd, s->P > 1, &err);
    if (!s->dev)
        {
        set_last_error(err);
        return PGSD_ERROR_NO_DEVICE;
        }
    return PGSD_SUCCESS;
    }
// Hand the rows of a placed chunk to where the placement says.
static int deliver_chunk(Impl* s, Queued& q, const Placement& pl, bool skip)
    {
    if (q.ticket >= 0)
        {
        std::string err;
        int rc;
        if (skip || pl.size == 0 || (!pl.buffered && !pl.write))
            rc = device_pipeline_commit(s->dev, q.ticket, q.ticket_index, -1, nullptr, &err);
        else if (pl.buffered)
            {
            const size_t at = s->write_buffer.size();
            s->write_buffer.resize(at + pl.size, 0); // every rank has accounted for this length already
            rc = device_pipeline_commit(s->dev, q.ticket, q.ticket_index, -1, s->write_buffer.data() + at, &err);
            }
        else
            rc = device_pipeline_commit(s->dev, q.ticket, q.ticket_index, pl.file_offset, nullptr, &err);
        if (rc != PGSD_SUCCESS)
            {
            set_last_error(err);
            remember_failure(s, rc, 0);
            }
        return rc;
        }
    if (skip || pl.size == 0)
        return PGSD_SUCCESS;
    const char* data = q.borrowed ? (const char*)q.borrowed : q.host.data();
    if (pl.buffered)
        {
        s->write_buffer.insert(s->write_buffer.end(), data, data + pl.size);
        return PGSD_SUCCESS;
        }
    if (!pl.write)
        return PGSD_SUCCESS;
    // the bytes of the chunk: MPI_File_write_at in the reference (pgsd.c:2229)
    TraceRange tr("pgsd:pwrite_host file_off=%llu bytes=%llu", (unsigned long long)pl.file_offset, pl.size);
    int e = writer_pool_pwrite_sync(s->get_pool(), s->fd, data, pl.size, pl.file_offset, s->P > 1);
    if (e != 0)
        {
        errno = -e;
        remember_failure(s, PGSD_ERROR_IO, -e);
        return PGSD_ERROR_IO;
        }
    return PGSD_SUCCESS;
    }

// The frame exchange: ONE allgather carries, for every rank, its status word (a failure it has not
// shared yet), the number of queued chunks and each chunk's byte count (or, top bit set, the code its
// argument check failed with).  Every rank then replays the reference's placement decisions
// (pgsd.c:2143-2256) for the queued chunks in call order -- max and sum of the sizes decide buffered or
// direct and how far the file advances -- so the bytes land exactly where per-chunk exchanges would
// have put them.  Chunks whose rows are partitioned automatically (PGSD_PARTITION_AUTO) get their
// global row count and this rank's first row from the same vector: no separate row-count allgather.
static int resolve_queue(Impl* s)
    {
    const size_t k = s->queue.size();
    const uint64_t FAILED = 1ull << 63;
    // Every message of the exchange has the SAME length on every rank, whatever a rank has queued:
    // ncclAllGather (like MPI_Allgather) is undefined for unequal send counts, so a rank that queued a
    // different number of chunks -- a caller bug -- must be told apart by the CONTENT of a well-formed
    // message, not by its size.  Round 0: [status, k, the first FRAME_SLOTS sizes, zero padded]; the usual
    // frame (a handful of chunks) is done with it: ONE collective.  Only when all ranks agree on a k beyond
    // FRAME_SLOTS do further rounds of FRAME_WORDS sizes each follow (their number follows from k alone).
    enum
        {
        FRAME_WORDS = 64,
        FRAME_SLOTS = FRAME_WORDS - 2
        };
    auto word_of = [&](size_t i) -> uint64_t
    {
        const Queued& q = s->queue[i];
        return q.local_rc != PGSD_SUCCESS ? (FAILED | (uint64_t)(uint32_t)(-q.local_rc))
                                          : q.N * q.M * sizeof_type(q.type);
    };
    std::vector<uint64_t> send(FRAME_WORDS, 0), recv;
    send[0] = (uint64_t)(uint32_t)s->sticky_rc | ((uint64_t)(uint32_t)s->sticky_errno << 32);
    send[1] = k;
    for (size_t i = 0; i < k && i < FRAME_SLOTS; i++)
        send[2 + i] = word_of(i);
    // sizes[r * k + i]: rank r's word for queued chunk i
    std::vector<uint64_t> words((size_t)s->P * k, 0);
    int first_rc = PGSD_SUCCESS;
    if (s->P > 1)
        {
        recv.assign((size_t)FRAME_WORDS * (size_t)s->P, 0);
        TraceRange tr("pgsd:frame_exchange chunks=%llu ranks=%llu", k, (unsigned long long)s->P);
        bool comm_ok = s->gather(send.data(), recv.data(), FRAME_WORDS * sizeof(uint64_t)) == 0;
        for (int r = 0; r < s->P && comm_ok; r++)
            if (recv[(size_t)r * FRAME_WORDS + 1] != k)
                {
                set_last_error("the ranks queued different numbers of chunks for this frame");
                comm_ok = false;
                }
        std::vector<uint64_t> status((size_t)s->P, 0);
        for (int r = 0; r < s->P && comm_ok; r++)
            {
            status[(size_t)r] = recv[(size_t)r * FRAME_WORDS];
            for (size_t i = 0; i < k && i < FRAME_SLOTS; i++)
                words[(size_t)r * k + i] = recv[(size_t)r * FRAME_WORDS + 2 + i];
            }
        // frames of more than FRAME_SLOTS chunks: every rank knows by now that all ranks hold the same k
        for (size_t base = FRAME_SLOTS; base < k && comm_ok; base += FRAME_WORDS)
            {
            std::fill(send.begin(), send.end(), 0);
            for (size_t i = base; i < k && i < base + FRAME_WORDS; i++)
                send[i - base] = word_of(i);
            comm_ok = s->gather(send.data(), recv.data(), FRAME_WORDS * sizeof(uint64_t)) == 0;
            for (int r = 0; r < s->P && comm_ok; r++)
                for (size_t i = base; i < k && i < base + FRAME_WORDS; i++)
                    words[(size_t)r * k + i] = recv[(size_t)r * FRAME_WORDS + (i - base)];
            }
        if (!comm_ok)
            {
            // Nothing sane can be placed any more.  The queue is dropped (its borrowed row pointers die
            // with this call, packed device chunks are released) and the failure stays with the handle.
            std::vector<Queued> dead;
            dead.swap(s->queue);
            Placement none;
            memset(&none, 0, sizeof(none));
            for (Queued& q : dead)
                if (q.ticket >= 0)
                    (void)deliver_chunk(s, q, none, true);
            remember_failure(s, PGSD_ERROR_COMM, 0);
            return PGSD_ERROR_COMM;
            }
        s->sticky_rc = PGSD_SUCCESS; // shared now
        s->sticky_errno = 0;
        for (int r = 0; r < s->P; r++)
            {
            const int src = (int)(int32_t)(uint32_t)status[(size_t)r];
            if (src != PGSD_SUCCESS && first_rc == PGSD_SUCCESS)
                {
                first_rc = src;
                errno = (int)(uint32_t)(status[(size_t)r] >> 32);
                }
            }
        }
    else
        {
        for (size_t i = 0; i < k; i++)
            words[i] = word_of(i);
        if (s->sticky_rc != PGSD_SUCCESS)
            {
            first_rc = s->sticky_rc;
            errno = s->sticky_errno;
            s->sticky_rc = PGSD_SUCCESS;
            s->sticky_errno = 0;
            }
        }
    std::vector<Queued> queue;
    queue.swap(s->queue);
    std::vector<uint64_t> sizes((size_t)s->P);
    for (size_t i = 0; i < k; i++)
        {
        Queued& q = queue[i];
        int bad = PGSD_SUCCESS;
        for (int r = 0; r < s->P; r++)
            {
            const uint64_t e = words[(size_t)r * k + i];
            if (e & FAILED)
                {
                if (bad == PGSD_SUCCESS)
                    bad = -(int)(uint32_t)(e & 0xffffffffu);
                sizes[(size_t)r] = 0;
                }
            else
                sizes[(size_t)r] = e;
            }
        Placement pl;
        memset(&pl, 0, sizeof(pl));
        int rc = bad;
        if (rc == PGSD_SUCCESS)
            {
            uint64_t N_global = q.N_global, offset = q.offset;
            if (N_global == PGSD_PARTITION_AUTO)
                {
                // rows of rank r = its bytes / bytes per row; this rank starts behind the lower ranks
                const uint64_t rowbytes = (uint64_t)q.M * sizeof_type(q.type);
                N_global = 0;
                offset = 0;
                for (int r = 0; r < s->P; r++)
                    {
                    const uint64_t rows = rowbytes ? sizes[(size_t)r] / rowbytes : 0;
                    if (r < s->rank)
                        offset += rows * q.M;
                    N_global += rows;
                    }
                }
            rc = place_chunk(s, q.name.c_str(), q.type, q.N, q.M, N_global, q.M_global, offset, q.all, sizes, &pl);
            }
        const int drc = deliver_chunk(s, q, pl, rc != PGSD_SUCCESS);
        if (rc == PGSD_SUCCESS)
            rc = drc;
        // A chunk that failed THIS rank's own check was refused when it was written (the call returned the code,
        // as the reference's does, pgsd.c:2090-2105); the resolving call reports what this rank has not been told
        // yet: another rank's refusal of a chunk, or a delivery that failed now.
        if (rc != PGSD_SUCCESS && first_rc == PGSD_SUCCESS && q.local_rc == PGSD_SUCCESS)
            first_rc = rc;
        }
    return first_rc;
    }
// Declared partition (pgsd_set_partition): every rank's byte count of a chunk follows from what the caller
// declared, so the chunk is placed without an exchange.  N_global == PGSD_PARTITION_AUTO: the chunk is
// partitioned by the declared rows (this rank must bring exactly its share); anything else must have the same
// size on every rank (replicated data; the default-argument call shape).  The placement is what place_chunk
// computes from those sizes, i.e. what the exchanges would have produced.
// `local`: this rank's argument / device verdict.  The other ranks cannot be told now, and they WILL place the
// chunk: so does this rank whenever it can (state stays in step, the rows are simply missing), and the failure
// is remembered for the next synchronisation point; when it cannot (no name, M == 0) the handle is poisoned.
static int trusted_place(Impl* s, const char* name, uint32_t type, uint64_t N, uint32_t M, uint64_t* N_global,
                         uint32_t M_global, uint64_t* offset, bool all, int local, Placement* pl, bool* deliver)
    {
    *deliver = false;
    memset(pl, 0, sizeof(*pl));
    if (s->poisoned)
        return s->sticky_rc != PGSD_SUCCESS ? s->sticky_rc : PGSD_ERROR_INVALID_ARGUMENT;
    const uint64_t unit = (uint64_t)M * sizeof_type(type);
    if (!name || unit == 0 || s->flags == PGSD_OPEN_READONLY)
        {
        if (s->flags != PGSD_OPEN_READONLY)
            {
            s->poisoned = true; // the other ranks place a chunk this rank cannot even size
            remember_failure(s, local != PGSD_SUCCESS ? local : PGSD_ERROR_INVALID_ARGUMENT, 0);
            }
        return local != PGSD_SUCCESS ? local : PGSD_ERROR_INVALID_ARGUMENT;
        }
    std::vector<uint64_t> sizes((size_t)s->P);
    if (*N_global == PGSD_PARTITION_AUTO)
        {
        uint64_t total = 0, before = 0;
        for (int r = 0; r < s->P; r++)
            {
            sizes[(size_t)r] = s->partition[(size_t)r] * unit;
            if (r < s->rank)
                before += s->partition[(size_t)r];
            total += s->partition[(size_t)r];
            }
        *N_global = total;
        *offset = before * M;
        if (N != s->partition[(size_t)s->rank] && local == PGSD_SUCCESS)
            {
            set_last_error("pgsd_set_partition declared another row count for this rank than the chunk brings");
            local = PGSD_ERROR_INVALID_ARGUMENT;
            }
        }
    else
        for (int r = 0; r < s->P; r++)
            sizes[(size_t)r] = N * unit;
    // place with the DECLARED size of this rank, so that the replicated state moves as on the other ranks
    const uint64_t n_declared = sizes[(size_t)s->rank] / unit;
    int rc = place_chunk(s, name, type, n_declared, M, *N_global, M_global, *offset, all, sizes, pl);
    if (rc != PGSD_SUCCESS)
        {
        // name list full etc.: replicated state, every rank fails alike
        return rc;
        }
    if (local != PGSD_SUCCESS)
        {
        remember_failure(s, local, errno);
        if (pl->buffered) // the buffer must keep the length every rank has accounted for
            s->write_buffer.insert(s->write_buffer.end(), pl->size, 0);
        return local;
        }
    *deliver = true;
    return PGSD_SUCCESS;
    }
    } // namespace pgsd_amd

using namespace pgsd_amd;

// ============================================================================ C ABI

extern "C" uint32_t pgsd_make_version(unsigned int major, unsigned int minor)
    try
    {
    return make_version(major, minor);
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return 0;
    }

extern "C" size_t pgsd_sizeof_type(enum pgsd_type type)
    try
    {
    return sizeof_type((uint32_t)type);
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return 0;
    }

static bool comm_usable(const pgsd_comm* c)
    {
    return c && c->allgather && c->size >= 1 && c->rank >= 0 && c->rank < c->size;
    }

static int create_and_open(const pgsd_comm* on, struct pgsd_handle* handle, const char* fname, const char* application,
                           const char* schema, uint32_t schema_version, enum pgsd_open_flag flags, int exclusive_create);
static int open_existing(const pgsd_comm* on, struct pgsd_handle* handle, const char* fname, enum pgsd_open_flag flags);

extern "C" int pgsd_create_and_open(struct pgsd_handle* handle, const char* fname, const char* application,
                                    const char* schema, uint32_t schema_version,
                                    enum pgsd_open_flag flags, int exclusive_create)
    try
    {
    return create_and_open(nullptr, handle, fname, application, schema, schema_version, flags, exclusive_create);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_create_and_open_on(const struct pgsd_comm* comm, struct pgsd_handle* handle, const char* fname,
                                       const char* application, const char* schema, uint32_t schema_version,
                                       enum pgsd_open_flag flags, int exclusive_create)
    try
    {
    if (!comm_usable(comm))
        return PGSD_ERROR_INVALID_ARGUMENT;
    return create_and_open(comm, handle, fname, application, schema, schema_version, flags, exclusive_create);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_open_on(const struct pgsd_comm* comm, struct pgsd_handle* handle, const char* fname,
                            enum pgsd_open_flag flags)
    try
    {
    if (!comm_usable(comm))
        return PGSD_ERROR_INVALID_ARGUMENT;
    return open_existing(comm, handle, fname, flags);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_handle_allgather(struct pgsd_handle* handle, const void* send, void* recv, size_t bytes)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || !send || !recv)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (s->P == 1)
        {
        if (send != recv)
            memcpy(recv, send, bytes);
        return PGSD_SUCCESS;
        }
    if (s->gather(send, recv, bytes) != 0)
        return PGSD_ERROR_COMM;
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

static int create_and_open(const pgsd_comm* on, struct pgsd_handle* handle, const char* fname, const char* application,
                           const char* schema, uint32_t schema_version, enum pgsd_open_flag flags, int exclusive_create)
    {
    // pgsd.c:1710-1773
    if (!handle || !fname || !application || !schema)
        return PGSD_ERROR_INVALID_ARGUMENT;
    memset(handle, 0, sizeof(*handle));
    handle->fd = -1;
    if (flags == PGSD_OPEN_READONLY)
        return PGSD_ERROR_FILE_MUST_BE_WRITABLE;
    Impl* s = new_impl(on);
    s->flags = flags;

    // rank 0 creates and lays out the file, then everybody opens it
    int rc = PGSD_SUCCESS;
    if (s->rank == 0)
        {
        s->fd = open(fname, O_RDWR | O_CREAT | (exclusive_create ? O_EXCL : 0), 0644);
        if (s->fd < 0)
            rc = PGSD_ERROR_IO;
        else
            rc = initialize_file(s->fd, application, schema, schema_version);
        }
    rc = agree_status(s, rc);
    if (rc == PGSD_SUCCESS && s->rank != 0)
        {
        s->fd = open(fname, O_RDWR);
        if (s->fd < 0)
            rc = PGSD_ERROR_IO;
        }
    if (rc == PGSD_SUCCESS)
        rc = initialize_handle(s);
    if (s->P > 1)
        {
        // agree after the local open/parse as well (bcast_retval, pgsd.c:1763)
        rc = agree_status(s, rc);
        }
    if (rc != PGSD_SUCCESS)
        {
        int saved = errno;
        destroy_impl(s);
        errno = saved;
        return rc;
        }
    handle->impl = s;
    publish(handle, s);
    return PGSD_SUCCESS;
    }

extern "C" int pgsd_open(struct pgsd_handle* handle, const char* fname, enum pgsd_open_flag flags)
    try
    {
    return open_existing(nullptr, handle, fname, flags);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

static int open_existing(const pgsd_comm* on, struct pgsd_handle* handle, const char* fname, enum pgsd_open_flag flags)
    {
    // pgsd.c:1775-1812
    if (!handle || !fname)
        return PGSD_ERROR_INVALID_ARGUMENT;
    memset(handle, 0, sizeof(*handle));
    handle->fd = -1;
    Impl* s = new_impl(on);
    s->flags = flags;
    int rc = PGSD_SUCCESS;
    s->fd = open(fname, flags == PGSD_OPEN_READONLY ? O_RDONLY : O_RDWR);
    if (s->fd < 0)
        rc = PGSD_ERROR_IO;
    else
        rc = initialize_handle(s);
    rc = agree_status(s, rc);
    if (rc != PGSD_SUCCESS)
        {
        int saved = errno;
        destroy_impl(s);
        errno = saved;
        return rc;
        }
    handle->impl = s;
    publish(handle, s);
    return PGSD_SUCCESS;
    }

extern "C" int pgsd_close(struct pgsd_handle* handle)
    try
    {
    // pgsd.c:1814-1914
    if (!handle)
        return PGSD_ERROR_INVALID_ARGUMENT;
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = PGSD_SUCCESS;
    if (!s->early.empty())
        release_early(s);
    if (s->flags != PGSD_OPEN_READONLY)
        {
        rc = do_flush(s);
        if (rc != PGSD_SUCCESS)
            {
            publish(handle, s);
            return rc;
            }
        }
    int fd = s->fd;
    s->fd = -1;
    destroy_impl(s);
    handle->impl = NULL;
    handle->file_index.data = NULL;
    handle->file_names.data.data = NULL;
    handle->fd = -1;
    if (close(fd) != 0)
        return PGSD_ERROR_IO;
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_end_frame(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = do_end_frame(s);
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_end_frame_async(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = do_end_frame(s, true);
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_frame_sync(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!s->dev || !s->inflight)
        return PGSD_SUCCESS;
    std::string err;
    int rc = device_pipeline_drain(s->dev, &err);
    s->inflight = false;
    if (rc != PGSD_SUCCESS)
        set_last_error(err);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_flush(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = do_flush(s);
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_write_chunk(struct pgsd_handle* handle, const char* name, enum pgsd_type type, uint64_t N,
                                uint32_t M, uint64_t N_global, uint32_t M_global, uint64_t offset,
                                uint64_t global_size, bool all, uint8_t flags, const void* data)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    (void)global_size; // dead in the reference as well (pgsd.c:2147-2151)
    const int local = check_chunk_args(s, name, N, M, flags, data != NULL);
    if (s->have_partition)
        {
        // declared partition: no exchange, placed and written at once
        Placement pl;
        bool deliver = false;
        int rc = trusted_place(s, name, (uint32_t)type, N, M, &N_global, M_global, &offset, all, local, &pl, &deliver);
        if (deliver && pl.size > 0)
            {
            if (pl.buffered)
                s->write_buffer.insert(s->write_buffer.end(), (const char*)data, (const char*)data + pl.size);
            else if (pl.write)
                {
                TraceRange tr("pgsd:pwrite_host file_off=%llu bytes=%llu", (unsigned long long)pl.file_offset, pl.size);
                int e = writer_pool_pwrite_sync(s->get_pool(), s->fd, data, pl.size, pl.file_offset, s->P > 1);
                if (e != 0)
                    {
                    errno = -e;
                    rc = PGSD_ERROR_IO;
                    remember_failure(s, rc, -e);
                    }
                }
            }
        publish(handle, s);
        return rc;
        }
    if (s->batch)
        {
        // frame-batched exchange: a replicated chunk (all == false) waits, with a copy of its rows, for
        // the frame's exchange; per-particle rows (all == true) are borrowed for the call only and need
        // their file offset now: they resolve the queue, themselves included, at once
        Queued q;
        q.name = name ? name : "";
        q.type = (uint32_t)type;
        q.N = N, q.M = M, q.N_global = N_global, q.M_global = M_global, q.offset = offset, q.all = all;
        q.local_rc = local;
        const uint64_t size = local == PGSD_SUCCESS ? N * M * sizeof_type((uint32_t)type) : 0;
        int rc = local;
        // Which of the two it is must not depend on anything a single rank sees differently (its
        // argument check, its byte count): `all` is the caller's flag, the same on every rank.
        if (all)
            {
            q.borrowed = local == PGSD_SUCCESS ? data : nullptr;
            s->queue.push_back(std::move(q));
            if (!s->defer_rows) // rows borrowed for the call only: place them (and everything queued before) now
                {
                const int qrc = resolve_queue(s);
                rc = local != PGSD_SUCCESS ? local : qrc;
                }
            }
        else
            {
            if (size > 0)
                q.host.assign((const char*)data, (const char*)data + size);
            s->queue.push_back(std::move(q));
            }
        publish(handle, s);
        return rc;
        }
    std::vector<uint64_t> sizes;
    int rc = exchange_counts(s, local == PGSD_SUCCESS ? N * M * sizeof_type((uint32_t)type) : 0, local, sizes);
    if (rc == PGSD_SUCCESS && N_global == PGSD_PARTITION_AUTO)
        auto_partition(s, sizes, (uint64_t)M * sizeof_type((uint32_t)type), M, &N_global, &offset);
    Placement pl;
    if (rc == PGSD_SUCCESS)
        rc = place_chunk(s, name, (uint32_t)type, N, M, N_global, M_global, offset, all, sizes, &pl);
    if (rc == PGSD_SUCCESS)
        {
        if (pl.buffered)
            {
            if (pl.size > 0)
                s->write_buffer.insert(s->write_buffer.end(), (const char*)data,
                                       (const char*)data + pl.size);
            }
        else if (pl.write && pl.size > 0)
            {
            // the bytes of the chunk: MPI_File_write_at in the reference (pgsd.c:2229)
            int e = writer_pool_pwrite_sync(s->get_pool(), s->fd, data, pl.size, pl.file_offset, s->P > 1);
            if (e != 0)
                {
                errno = -e;
                rc = PGSD_ERROR_IO;
                remember_failure(s, rc, -e);
                }
            }
        }
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" uint64_t pgsd_get_nframes(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    return s ? s->cur_frame : 0;
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return 0;
    }

extern "C" uint64_t pgsd_get_nnames(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    return s ? s->file_n_names : 0;
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return 0;
    }

extern "C" const struct pgsd_index_entry* pgsd_find_chunk(struct pgsd_handle* handle, uint64_t frame,
                                                          const char* name)
    try
    {
    // pgsd.c:2295-2434; valid on every rank because the index is replicated
    Impl* s = impl_of(handle);
    if (!s || !name)
        return NULL;
    if (frame >= s->cur_frame)
        return NULL;
    if (s->flags != PGSD_OPEN_READONLY)
        {
        int rc = flush_for_lookup(s);
        publish(handle, s);
        if (rc != PGSD_SUCCESS)
            return NULL;
        }
    auto it = s->name_map.find(name);
    if (it == s->name_map.end())
        return NULL;
    uint16_t match_id = it->second;

    if (!s->v1())
        {
        ssize_t L = 0, R = (ssize_t)s->file_index_size - 1;
        pgsd_index_entry T;
        memset(&T, 0, sizeof(T));
        T.frame = frame;
        T.id = match_id;
        while (L <= R)
            {
            size_t m = (size_t)((L + R) / 2);
            int c = cmp_entry(s->file_index[m], T);
            if (c == -1)
                L = (ssize_t)m + 1;
            else if (c == 1)
                R = (ssize_t)m - 1;
            else
                return &s->file_index[m];
            }
        return NULL;
        }
    // v1 files: the index is only ordered by frame (pgsd.c:2380-2430)
    if (s->file_index_size == 0)
        return NULL;
    size_t L = 0, R = s->file_index_size;
    do
        {
        size_t m = (L + R) / 2;
        if (frame < s->file_index[m].frame)
            R = m;
        else
            L = m;
        } while ((R - L) > 1);
    for (int64_t cur = (int64_t)L; cur >= 0 && s->file_index[(size_t)cur].frame == frame; cur--)
        if (s->file_index[(size_t)cur].id == match_id)
            return &s->file_index[(size_t)cur];
    return NULL;
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return nullptr;
    }

extern "C" int pgsd_read_chunk(struct pgsd_handle* handle, void* data, const struct pgsd_index_entry* chunk,
                               uint64_t N, uint32_t M, uint32_t offset, bool all)
    try
    {
    // pgsd.c:2436-2537
    Impl* s = impl_of(handle);
    if (!s || !data || !chunk)
        return PGSD_ERROR_INVALID_ARGUMENT;
    // copy first: a flush may move the index storage the entry points into
    pgsd_index_entry c = *chunk;
    if (s->flags != PGSD_OPEN_READONLY)
        {
        int rc = flush_for_read(s);
        publish(handle, s);
        if (rc != PGSD_SUCCESS)
            return rc;
        }
    size_t sz = sizeof_type(c.type);
    uint64_t stride = 0;
    size_t size;
    uint64_t off_elems = (uint64_t)offset * M;
    if (!all)
        size = c.N * c.M * sz;
    else
        {
        size = N * M * sz;
        stride = off_elems * sz;
        }
    if (size == 0)
        return PGSD_ERROR_FILE_CORRUPT;
    if (c.location == 0)
        return PGSD_ERROR_FILE_CORRUPT;
    if ((uint64_t)(c.location + size + stride) > (uint64_t)s->file_size)
        return PGSD_ERROR_FILE_CORRUPT;
    pread_parallel(s->fd, data, size, c.location + (long long)stride);
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" const char* pgsd_find_matching_chunk_name(struct pgsd_handle* handle, const char* match,
                                                     const char* prev)
    try
    {
    // pgsd.c:2557-2641
    Impl* s = impl_of(handle);
    if (!s || !match)
        return NULL;
    if (s->file_n_names == 0)
        return NULL; // checked before the flush, like pgsd.c:2573-2584
    if (s->flags != PGSD_OPEN_READONLY)
        {
        // `prev` points into the name storage, which a flush may reallocate: carry it over
        size_t prev_off = 0;
        bool have_prev = prev != NULL;
        if (have_prev)
            {
            if (prev < s->file_names.d.data() || prev >= s->file_names.d.data() + s->file_names.reserved())
                return NULL;
            prev_off = (size_t)(prev - s->file_names.d.data());
            }
        int rc = flush_for_lookup(s);
        publish(handle, s);
        if (rc != PGSD_SUCCESS)
            return NULL;
        if (have_prev)
            prev = s->file_names.d.data() + prev_off;
        }
    if (s->file_n_names == 0)
        return NULL;
    const char* base = s->file_names.d.data();
    const char* end = base + s->file_names.reserved();
    if (end[-1] != 0)
        return NULL;
    const char* p;
    if (!prev)
        p = base;
    else
        {
        if (prev < base || prev >= end)
            return NULL;
        p = s->v1() ? prev + PGSD_NAME_SIZE : prev + strlen(prev) + 1;
        }
    size_t ml = strlen(match);
    while (p < end)
        {
        if (p[0] != 0 && 0 == strncmp(match, p, ml))
            return p;
        p += s->v1() ? (size_t)PGSD_NAME_SIZE : strlen(p) + 1;
        }
    return NULL;
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return nullptr;
    }

extern "C" uint64_t pgsd_get_maximum_write_buffer_size(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    return s ? s->maxbuf : 0;
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return 0;
    }

extern "C" int pgsd_set_maximum_write_buffer_size(struct pgsd_handle* handle, uint64_t size)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || size == 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!s->queue.empty()) // queued chunks are placed under the limit they were written under
        {
        int rc = resolve_queue(s);
        if (rc != PGSD_SUCCESS)
            return rc;
        }
    s->maxbuf = size;
    publish(handle, s);
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" uint64_t pgsd_get_index_entries_to_buffer(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    return s ? s->idxbuf : 0;
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return 0;
    }

extern "C" int pgsd_set_index_entries_to_buffer(struct pgsd_handle* handle, uint64_t number)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || number == 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    s->idxbuf = number;
    publish(handle, s);
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_set_frame_exchange(struct pgsd_handle* handle, int batched)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = PGSD_SUCCESS;
    if (!batched && s->batch && s->flags != PGSD_OPEN_READONLY)
        rc = do_flush(s); // leave nothing queued and nothing unsynchronised behind
    s->batch = batched != 0;
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_set_partition(struct pgsd_handle* handle, const uint64_t* rows, uint32_t n_ranks)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || (rows && n_ranks != (uint32_t)s->P))
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = PGSD_SUCCESS;
    if (!s->queue.empty()) // chunks queued under the batched exchange are placed by it, before the rules change
        rc = resolve_queue(s);
    if (rows)
        {
        s->partition.assign(rows, rows + n_ranks);
        s->have_partition = true;
        }
    else
        {
        s->partition.clear();
        s->have_partition = false;
        }
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_set_local_reads(struct pgsd_handle* handle, int on)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    s->local_reads = on != 0;
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_set_deferred_rows(struct pgsd_handle* handle, int on)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = PGSD_SUCCESS;
    if (!on && s->defer_rows && !s->queue.empty()) // rows queued under the promise are placed while it still holds
        rc = resolve_queue(s);
    s->defer_rows = on != 0;
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_frame_exchange(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    const int rc = s->queue.empty() ? PGSD_SUCCESS : resolve_queue(s);
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" uint64_t pgsd_get_collective_count(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    return s ? s->n_collectives : 0;
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return 0;
    }

extern "C" int pgsd_get_exchange_stats(struct pgsd_handle* handle, struct pgsd_exchange_stats* out, int reset)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || !out)
        return PGSD_ERROR_INVALID_ARGUMENT;
    out->count = s->exch_count;
    out->total_us = s->exch_us_sum;
    out->max_us = s->exch_us_max;
    out->min_us = s->exch_us_min;
    if (reset)
        {
        s->exch_count = 0;
        s->exch_us_sum = s->exch_us_max = s->exch_us_min = 0;
        }
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

// ---------------------------------------------------------------------------- device path

extern "C" int pgsd_device_configure(struct pgsd_handle* handle, const struct pgsd_device_config* cfg)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || !cfg)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (s->dev)
        {
        if (!s->early.empty())
            release_early(s);
        if (!s->queue.empty()) // packed chunks of the old pipeline still wait for their placement
            {
            int qrc = resolve_queue(s);
            if (qrc != PGSD_SUCCESS)
                return qrc;
            }
        std::string err;
        int rc = device_pipeline_drain(s->dev, &err);
        if (rc != PGSD_SUCCESS)
            {
            set_last_error(err);
            return rc;
            }
        device_pipeline_destroy(s->dev);
        s->dev = nullptr;
        }
    s->devcfg = *cfg;
    s->devcfg_set = true;
    return ensure_device(s);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

static int check_field(const pgsd_field_desc* f, uint32_t dst_type, uint32_t M)
    {
    if (!f || !f->src)
        return PGSD_ERROR_INVALID_ARGUMENT;
    size_t ssz = sizeof_type(f->src_type), dsz = sizeof_type(dst_type);
    if (ssz == 0 || dsz == 0 || M == 0 || f->src_col0 + M > f->src_stride)
        return PGSD_ERROR_INVALID_ARGUMENT;
    bool s_int = f->src_type <= PGSD_TYPE_INT64, d_int = dst_type <= PGSD_TYPE_INT64;
    if (f->bitcast)
        return dsz <= ssz ? PGSD_SUCCESS : PGSD_ERROR_INVALID_ARGUMENT;
    if (!s_int && d_int)
        return PGSD_ERROR_INVALID_ARGUMENT; // float -> integer is not offered
    if (s_int && !d_int && ssz == 8)
        return PGSD_ERROR_INVALID_ARGUMENT; // 64-bit integer -> float is not offered
    return PGSD_SUCCESS;
    }

extern "C" int pgsd_write_chunk_device(struct pgsd_handle* handle, const char* name, enum pgsd_type type,
                                       uint64_t N, uint32_t M, uint64_t N_global, uint32_t M_global,
                                       uint64_t offset, uint64_t global_size, bool all, uint8_t flags,
                                       const struct pgsd_field_desc* src)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    (void)global_size;
    // everything that can fail on this rank alone comes first and travels with the exchange:
    // a rank without a usable device or with a bad field makes the call fail on every rank
    int local = check_chunk_args(s, name, N, M, flags, N == 0 || (src && src->src));
    if (local == PGSD_SUCCESS && N > 0)
        local = check_field(src, (uint32_t)type, M);
    if (local == PGSD_SUCCESS)
        local = ensure_device(s);
    if (s->have_partition)
        {
        Placement pl;
        bool deliver = false;
        int rc = trusted_place(s, name, (uint32_t)type, N, M, &N_global, M_global, &offset, all, local, &pl, &deliver);
        if (deliver && pl.size > 0 && (pl.buffered || pl.write))
            {
            std::vector<DeviceChunk> chunks(1);
            DeviceChunk& c = chunks[0];
            memset(&c, 0, sizeof(c));
            c.job.dst_type = (uint32_t)type;
            c.job.M = M;
            c.job.src = *src;
            c.N = N;
            std::vector<char> tmp;
            if (pl.buffered)
                {
                tmp.resize(pl.size);
                c.file_offset = -1;
                c.host_dst = tmp.data();
                }
            else
                c.file_offset = pl.file_offset;
            std::string err;
            rc = device_pipeline_submit(s->dev, chunks, N, &err);
            if (rc != PGSD_SUCCESS)
                {
                set_last_error(err);
                remember_failure(s, rc, 0);
                if (pl.buffered)
                    s->write_buffer.insert(s->write_buffer.end(), pl.size, 0);
                }
            else if (pl.buffered)
                s->write_buffer.insert(s->write_buffer.end(), tmp.begin(), tmp.end());
            }
        publish(handle, s);
        return rc;
        }
    if (s->batch)
        {
        // pack now (the kernel needs no file offset), place at the frame's exchange
        Queued q;
        q.name = name ? name : "";
        q.type = (uint32_t)type;
        q.N = N, q.M = M, q.N_global = N_global, q.M_global = M_global, q.offset = offset, q.all = all;
        q.local_rc = local;
        if (local == PGSD_SUCCESS)
            {
            std::vector<DeviceChunk> chunks(1);
            memset(&chunks[0], 0, sizeof(DeviceChunk));
            chunks[0].job.dst_type = (uint32_t)type;
            chunks[0].job.M = M;
            if (N > 0)
                chunks[0].job.src = *src;
            chunks[0].N = N;
            std::string err;
            q.local_rc = device_pipeline_stage(s->dev, chunks, N, &q.ticket, &err);
            if (q.local_rc != PGSD_SUCCESS)
                {
                set_last_error(err);
                q.ticket = -1;
                }
            }
        const int rc = q.local_rc;
        s->queue.push_back(std::move(q));
        publish(handle, s);
        return rc;
        }
    std::vector<uint64_t> sizes;
    int rc = exchange_counts(s, local == PGSD_SUCCESS ? N * M * sizeof_type((uint32_t)type) : 0, local, sizes);
    if (rc == PGSD_SUCCESS && N_global == PGSD_PARTITION_AUTO)
        auto_partition(s, sizes, (uint64_t)M * sizeof_type((uint32_t)type), M, &N_global, &offset);
    Placement pl;
    if (rc == PGSD_SUCCESS)
        rc = place_chunk(s, name, (uint32_t)type, N, M, N_global, M_global, offset, all, sizes, &pl);
    if (rc == PGSD_SUCCESS && pl.size > 0 && (pl.buffered || pl.write))
        {
        std::vector<DeviceChunk> chunks(1);
        DeviceChunk& c = chunks[0];
        memset(&c, 0, sizeof(c));
        c.job.dst_type = (uint32_t)type;
        c.job.M = M;
        c.job.src = *src;
        c.N = N;
        std::vector<char> tmp;
        if (pl.buffered)
            {
            tmp.resize(pl.size);
            c.file_offset = -1;
            c.host_dst = tmp.data();
            }
        else
            c.file_offset = pl.file_offset;
        std::string err;
        rc = device_pipeline_submit(s->dev, chunks, N, &err);
        if (rc != PGSD_SUCCESS)
            {
            set_last_error(err);
            remember_failure(s, rc, 0);
            if (pl.buffered) // the buffer must keep the length every rank has accounted for
                s->write_buffer.insert(s->write_buffer.end(), pl.size, 0);
            }
        else if (pl.buffered)
            s->write_buffer.insert(s->write_buffer.end(), tmp.begin(), tmp.end());
        }
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_write_chunks_device(struct pgsd_handle* handle, uint32_t n_chunks,
                                        const struct pgsd_chunk_req* reqs, uint64_t N, uint64_t N_global,
                                        uint64_t offset_rows)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || !reqs || n_chunks == 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int local = PGSD_SUCCESS;
    for (uint32_t i = 0; i < n_chunks && local == PGSD_SUCCESS; i++)
        {
        local = check_chunk_args(s, reqs[i].name, N, reqs[i].M, 0, N == 0 || reqs[i].src.src);
        if (local == PGSD_SUCCESS && N > 0)
            local = check_field(&reqs[i].src, reqs[i].type, reqs[i].M);
        }
    if (local == PGSD_SUCCESS)
        local = ensure_device(s);
    if (s->have_partition)
        {
        // declared partition: every chunk placed now, ONE fused launch, copies and writes start at once
        std::vector<DeviceChunk> chunks;
        int rc = PGSD_SUCCESS;
        for (uint32_t i = 0; i < n_chunks; i++)
            {
            Placement pl;
            bool deliver = false;
            uint64_t ng = N_global, off = offset_rows * reqs[i].M;
            int prc = trusted_place(s, reqs[i].name, reqs[i].type, N, reqs[i].M, &ng, reqs[i].M, &off, true, local, &pl,
                                    &deliver);
            if (prc != PGSD_SUCCESS && rc == PGSD_SUCCESS)
                rc = prc;
            if (deliver && pl.size > 0)
                {
                DeviceChunk c;
                memset(&c, 0, sizeof(c));
                c.job.dst_type = reqs[i].type;
                c.job.M = reqs[i].M;
                c.job.src = reqs[i].src;
                c.N = N;
                c.file_offset = pl.file_offset;
                chunks.push_back(c);
                }
            }
        if (!chunks.empty())
            {
            std::string err;
            int drc = device_pipeline_submit(s->dev, chunks, N, &err);
            if (drc != PGSD_SUCCESS)
                {
                set_last_error(err);
                remember_failure(s, drc, 0);
                if (rc == PGSD_SUCCESS)
                    rc = drc;
                }
            }
        publish(handle, s);
        return rc;
        }
    if (s->batch)
        {
        // one fused pack launch now, placement of every chunk at the frame's exchange
        int ticket = -1;
        if (local == PGSD_SUCCESS)
            {
            std::vector<DeviceChunk> staged(n_chunks);
            for (uint32_t i = 0; i < n_chunks; i++)
                {
                memset(&staged[i], 0, sizeof(DeviceChunk));
                staged[i].job.dst_type = reqs[i].type;
                staged[i].job.M = reqs[i].M;
                staged[i].job.src = reqs[i].src;
                staged[i].N = N;
                }
            std::string err;
            local = device_pipeline_stage(s->dev, staged, N, &ticket, &err);
            if (local != PGSD_SUCCESS)
                {
                set_last_error(err);
                ticket = -1;
                }
            }
        for (uint32_t i = 0; i < n_chunks; i++)
            {
            Queued q;
            q.name = reqs[i].name ? reqs[i].name : "";
            q.type = reqs[i].type;
            q.N = N, q.M = reqs[i].M, q.N_global = N_global, q.M_global = reqs[i].M;
            q.offset = offset_rows * reqs[i].M;
            q.all = true;
            q.local_rc = local;
            q.ticket = ticket;
            q.ticket_index = i;
            s->queue.push_back(std::move(q));
            }
        publish(handle, s);
        return local;
        }
    // ONE exchange for all chunks of the call: they share the row count, so every rank's byte
    // count of chunk i is rows[r] * M_i * sizeof(type_i)
    std::vector<uint64_t> rows;
    int rc = exchange_counts(s, N, local, rows);
    if (rc == PGSD_SUCCESS && N_global == PGSD_PARTITION_AUTO)
        {
        uint64_t off_elems = 0;
        auto_partition(s, rows, 1, 1, &N_global, &off_elems);
        offset_rows = off_elems;
        }
    std::vector<DeviceChunk> chunks;
    std::vector<uint64_t> sizes((size_t)s->P);
    for (uint32_t i = 0; i < n_chunks && rc == PGSD_SUCCESS; i++)
        {
        const pgsd_chunk_req& q = reqs[i];
        for (int r = 0; r < s->P; r++)
            sizes[(size_t)r] = rows[(size_t)r] * q.M * sizeof_type(q.type);
        Placement pl;
        rc = place_chunk(s, q.name, q.type, N, q.M, N_global, q.M, offset_rows * q.M, true, sizes, &pl);
        if (rc == PGSD_SUCCESS && pl.size > 0)
            {
            DeviceChunk c;
            memset(&c, 0, sizeof(c));
            c.job.dst_type = q.type;
            c.job.M = q.M;
            c.job.src = q.src;
            c.N = N;
            c.file_offset = pl.file_offset;
            chunks.push_back(c);
            }
        }
    if (rc == PGSD_SUCCESS && !chunks.empty())
        {
        std::string err;
        rc = device_pipeline_submit(s->dev, chunks, N, &err);
        if (rc != PGSD_SUCCESS)
            {
            set_last_error(err);
            remember_failure(s, rc, 0);
            }
        }
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_stage_chunks_device(struct pgsd_handle* handle, uint32_t n_chunks, const struct pgsd_chunk_req* reqs,
                                        uint64_t N, uint64_t* ticket_out)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || !reqs || n_chunks == 0 || !ticket_out)
        return PGSD_ERROR_INVALID_ARGUMENT;
    *ticket_out = 0;
    int local = PGSD_SUCCESS;
    for (uint32_t i = 0; i < n_chunks && local == PGSD_SUCCESS; i++)
        {
        local = check_chunk_args(s, reqs[i].name, N, reqs[i].M, 0, N == 0 || reqs[i].src.src);
        if (local == PGSD_SUCCESS && N > 0)
            local = check_field(&reqs[i].src, reqs[i].type, reqs[i].M);
        }
    if (local == PGSD_SUCCESS)
        local = ensure_device(s);
    EarlyStage e;
    e.N = N;
    for (uint32_t i = 0; i < n_chunks; i++)
        {
        e.names.push_back(reqs[i].name ? reqs[i].name : "");
        e.types.push_back(reqs[i].type);
        e.Ms.push_back(reqs[i].M);
        }
    e.claimed.assign(n_chunks, false);
    if (local == PGSD_SUCCESS)
        {
        std::vector<DeviceChunk> staged(n_chunks);
        for (uint32_t i = 0; i < n_chunks; i++)
            {
            memset(&staged[i], 0, sizeof(DeviceChunk));
            staged[i].job.dst_type = reqs[i].type;
            staged[i].job.M = reqs[i].M;
            staged[i].job.src = reqs[i].src;
            staged[i].N = N;
            }
        std::string err;
        local = device_pipeline_stage(s->dev, staged, N, &e.ticket, &err);
        if (local != PGSD_SUCCESS)
            {
            set_last_error(err);
            e.ticket = -1;
            }
        }
    // a failed staging keeps its ticket too: every rank goes on to make the same pgsd_write_staged_chunks calls,
    // which is where the other ranks learn of it
    e.local_rc = local;
    *ticket_out = s->next_early++;
    s->early[*ticket_out] = std::move(e);
    return local;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_write_staged_chunks(struct pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                        uint64_t N_global, uint64_t offset_rows)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || count == 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    auto it = s->early.find(ticket);
    if (it == s->early.end() || (uint64_t)first + count > it->second.claimed.size())
        return PGSD_ERROR_INVALID_ARGUMENT;
    EarlyStage& e = it->second;
    for (uint32_t i = first; i < first + count; i++)
        if (e.claimed[i])
            return PGSD_ERROR_INVALID_ARGUMENT;
    const int local = e.local_rc;
    const uint64_t N = e.N;
    int rc = local;
    if (s->have_partition)
        {
        rc = PGSD_SUCCESS;
        for (uint32_t i = first; i < first + count; i++)
            {
            Placement pl;
            bool deliver = false;
            uint64_t ng = N_global, off = offset_rows * e.Ms[i];
            int prc = trusted_place(s, e.names[i].c_str(), e.types[i], N, e.Ms[i], &ng, e.Ms[i], &off, true, local, &pl,
                                    &deliver);
            if (e.ticket >= 0)
                {
                std::string err;
                const bool skip = !deliver || pl.size == 0;
                int drc = device_pipeline_commit(s->dev, e.ticket, i, skip ? -1 : pl.file_offset, nullptr, &err);
                if (drc != PGSD_SUCCESS && prc == PGSD_SUCCESS)
                    {
                    set_last_error(err);
                    remember_failure(s, drc, 0);
                    prc = drc;
                    }
                }
            if (prc != PGSD_SUCCESS && rc == PGSD_SUCCESS)
                rc = prc;
            }
        }
    else if (s->batch)
        {
        for (uint32_t i = first; i < first + count; i++)
            {
            Queued q;
            q.name = e.names[i];
            q.type = e.types[i];
            q.N = N, q.M = e.Ms[i], q.N_global = N_global, q.M_global = e.Ms[i];
            q.offset = offset_rows * e.Ms[i];
            q.all = true;
            q.local_rc = local;
            q.ticket = e.ticket;
            q.ticket_index = i;
            s->queue.push_back(std::move(q));
            }
        }
    else
        {
        // one exchange for the chunks of the call (they share the row count), then placement and hand-over
        std::vector<uint64_t> rows;
        rc = exchange_counts(s, N, local, rows);
        if (rc == PGSD_SUCCESS && N_global == PGSD_PARTITION_AUTO)
            {
            uint64_t off_elems = 0;
            auto_partition(s, rows, 1, 1, &N_global, &off_elems);
            offset_rows = off_elems;
            }
        std::vector<uint64_t> sizes((size_t)s->P);
        for (uint32_t i = first; i < first + count; i++)
            {
            Placement pl;
            memset(&pl, 0, sizeof(pl));
            int prc = rc;
            if (prc == PGSD_SUCCESS)
                {
                for (int r = 0; r < s->P; r++)
                    sizes[(size_t)r] = rows[(size_t)r] * e.Ms[i] * sizeof_type(e.types[i]);
                prc = place_chunk(s, e.names[i].c_str(), e.types[i], N, e.Ms[i], N_global, e.Ms[i], offset_rows * e.Ms[i],
                                  true, sizes, &pl);
                }
            if (e.ticket >= 0)
                {
                std::string err;
                const bool skip = prc != PGSD_SUCCESS || pl.size == 0;
                int drc = device_pipeline_commit(s->dev, e.ticket, i, skip ? -1 : pl.file_offset, nullptr, &err);
                if (drc != PGSD_SUCCESS && prc == PGSD_SUCCESS)
                    {
                    set_last_error(err);
                    remember_failure(s, drc, 0);
                    prc = drc;
                    }
                }
            if (prc != PGSD_SUCCESS && rc == PGSD_SUCCESS)
                rc = prc;
            }
        }
    for (uint32_t i = first; i < first + count; i++)
        e.claimed[i] = true;
    bool all_claimed = true;
    for (bool c : e.claimed)
        all_claimed = all_claimed && c;
    if (all_claimed)
        s->early.erase(it);
    publish(handle, s);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

// staged chunks [first, first + count) of a ticket that have not been written yet
static int staged_range(Impl* s, uint64_t ticket, uint32_t first, uint32_t count, EarlyStage** out)
    {
    if (!s || count == 0)
        return PGSD_ERROR_INVALID_ARGUMENT;
    auto it = s->early.find(ticket);
    if (it == s->early.end() || (uint64_t)first + count > it->second.claimed.size())
        return PGSD_ERROR_INVALID_ARGUMENT;
    for (uint32_t i = first; i < first + count; i++)
        if (it->second.claimed[i])
            return PGSD_ERROR_INVALID_ARGUMENT;
    *out = &it->second;
    return PGSD_SUCCESS;
    }

extern "C" int pgsd_compare_staged_chunks(struct pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                          const void* const* ref, const uint64_t* ref_bytes, uint8_t* equal)
    try
    {
    Impl* s = impl_of(handle);
    EarlyStage* e = nullptr;
    if (!ref || !equal)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = staged_range(s, ticket, first, count, &e);
    if (rc != PGSD_SUCCESS)
        return rc;
    memset(equal, 0, count);
    if (e->local_rc != PGSD_SUCCESS || e->ticket < 0)
        return e->local_rc != PGSD_SUCCESS ? e->local_rc : PGSD_ERROR_DEVICE; // the staging failed: nothing to compare
    std::string err;
    rc = device_pipeline_compare(s->dev, e->ticket, first, count, ref, ref_bytes, equal, &err);
    if (rc != PGSD_SUCCESS)
        {
        set_last_error(err);
        memset(equal, 0, count);
        }
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_copy_staged_chunks(struct pgsd_handle* handle, uint64_t ticket, uint32_t first, uint32_t count,
                                       void* const* dst)
    try
    {
    Impl* s = impl_of(handle);
    EarlyStage* e = nullptr;
    if (!dst)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = staged_range(s, ticket, first, count, &e);
    if (rc != PGSD_SUCCESS)
        return rc;
    if (e->local_rc != PGSD_SUCCESS || e->ticket < 0)
        return e->local_rc != PGSD_SUCCESS ? e->local_rc : PGSD_ERROR_DEVICE;
    std::string err;
    rc = device_pipeline_copy_staged(s->dev, e->ticket, first, count, dst, &err);
    if (rc != PGSD_SUCCESS)
        set_last_error(err);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_read_chunk_device(struct pgsd_handle* handle, const struct pgsd_index_entry* chunk, uint64_t N,
                                      uint64_t row_offset, const struct pgsd_field_dst* dst)
    try
    {
    // device twin of pgsd_read_chunk's all==true slab read (pgsd.c:2498-2534)
    Impl* s = impl_of(handle);
    if (!s || !chunk || !dst || !dst->dst)
        return PGSD_ERROR_INVALID_ARGUMENT;
    pgsd_index_entry c = *chunk; // a flush may move the index storage
    if (s->flags != PGSD_OPEN_READONLY)
        {
        int rc = flush_for_read(s);
        publish(handle, s);
        if (rc != PGSD_SUCCESS)
            return rc;
        }
    const size_t sz = sizeof_type(c.type);
    if (sz == 0 || c.M == 0)
        return PGSD_ERROR_FILE_CORRUPT;
    if (N == 0)
        return PGSD_SUCCESS;
    if (row_offset + N > c.N)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (c.location == 0)
        return PGSD_ERROR_FILE_CORRUPT;
    const uint64_t rowbytes = (uint64_t)c.M * sz;
    const long long foff = c.location + (long long)(row_offset * rowbytes);
    const size_t bytes = (size_t)(N * rowbytes);
    if ((uint64_t)(foff + (long long)bytes) > (uint64_t)s->file_size)
        return PGSD_ERROR_FILE_CORRUPT;
    int rc = ensure_device(s);
    if (rc != PGSD_SUCCESS)
        return rc;
    pgsd_unpack_job job;
    memset(&job, 0, sizeof(job));
    job.src_type = c.type;
    job.M = c.M;
    job.dst = *dst;
    std::string err;
    rc = device_pipeline_read(s->dev, foff, bytes, job, N, &err);
    if (rc != PGSD_SUCCESS)
        set_last_error(err);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_device_wait_read(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!s->dev)
        return PGSD_SUCCESS;
    std::string err;
    int rc = device_pipeline_wait_read(s->dev, &err);
    if (rc != PGSD_SUCCESS)
        set_last_error(err);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_device_set_source_stream(struct pgsd_handle* handle, void* stream)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int rc = ensure_device(s);
    if (rc != PGSD_SUCCESS)
        return rc;
    device_pipeline_set_source_stream(s->dev, stream);
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_device_wait_packed(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!s->dev)
        return PGSD_SUCCESS;
    std::string err;
    int rc = device_pipeline_wait_packed(s->dev, &err);
    if (rc != PGSD_SUCCESS)
        set_last_error(err);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_device_get_stats(struct pgsd_handle* handle, struct pgsd_device_stats* out, int reset)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || !out)
        return PGSD_ERROR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    if (s->dev)
        device_pipeline_stats(s->dev, out, reset);
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }
