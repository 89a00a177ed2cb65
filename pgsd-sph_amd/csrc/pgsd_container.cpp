// pgsd_container.cpp -- the GSD v2 container behind the pgsd.h C ABI: file skeleton, name list, index (heap sort,
// relocation), small-chunk buffers, flush / end_frame, open / close and the handle's getters and setters.
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
//
// Chunk placement lives in pgsd_placement.cpp, lookups and reads in pgsd_read.cpp; pgsd_file_impl.hpp holds the state
// they share.
#include "pgsd_file_impl.hpp"

namespace pgsd_amd
    {
size_t sizeof_type(uint32_t type)
    {
    static const size_t s[] = {0, 1, 2, 4, 8, 1, 2, 4, 8, 4, 8}; // pgsd.c:2539-2555
    return (type >= 1 && type <= 10) ? s[type] : 0;
    }

int cmp_entry(const pgsd_index_entry& a, const pgsd_index_entry& b)
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

void sort_index(std::vector<pgsd_index_entry>& v)
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

// every rank learns the first non-zero status (rank order) and its errno.  check_state (the flush's exchange): the
// ranks also compare what they believe about the file -- its size, the frame counter, the number of names and of index
// entries.  The metadata is replicated, not broadcast (the reference lets rank 0's view win, pgsd.c:2219-2222): it
// stays identical as long as every rank makes the same calls with sizes that agree -- which the exchanges check, and a
// DECLARED partition (pgsd_set_partition) takes on trust for chunks that are not partitioned.  A caller
// that broke that trust is told here, on every rank, instead of leaving ranks with different layouts of one file.
int agree_status(Impl* s, int local_rc, bool check_state)
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
int initialize_file(int fd, const char* application, const char* schema, uint32_t schema_version)
    {
    if (io_truncate(fd, 0) != 0)
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
size_t used_entries(const std::vector<pgsd_index_entry>& v)
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

bool entry_valid(const Impl* s, const pgsd_index_entry& e)
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
int initialize_handle(Impl* s)
    {
    memset(&s->header, 0, sizeof(s->header));
    pread_some(s->fd, &s->header, sizeof(s->header), 0);
    if (s->header.magic != MAGIC_ID)
        return PGSD_ERROR_NOT_A_PGSD_FILE;
    if (s->header.pgsd_version < make_version(1, 0) && s->header.pgsd_version != make_version(0, 3))
        return PGSD_ERROR_INVALID_PGSD_FILE_VERSION;
    if (s->header.pgsd_version >= make_version(3, 0))
        return PGSD_ERROR_INVALID_PGSD_FILE_VERSION;

    s->file_size = io_file_size(s->fd);
    if (s->file_size < 0)
        return PGSD_ERROR_IO;
    s->placed_end = s->file_size; // nothing of this handle is on its way yet

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

void destroy_impl(Impl* s)
    {
    if (!s)
        return;
    if (s->dev)
        device_pipeline_destroy(s->dev);
    if (s->pool)
        writer_pool_destroy(s->pool);
    if (s->fd >= 0)
        io_close(s->fd);
    delete s;
    }

Impl* new_impl(const pgsd_comm* on)
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

// metadata bytes to the file: at once, or -- during an asynchronous seal -- through the pipeline's ONE writer
// thread, in call order and behind what is queued there at that moment: the data of a small (direct-path) frame; the
// pieces of a frame staged in HBM join the queue as their copies finish, so its index entries may reach the file
// before its last rows do -- the frame is complete after pgsd_frame_sync(), as pgsd.h says (a failure surfaces like
// a device chunk's: at the next drain, on every rank at the next flush)
int meta_pwrite(Impl* s, const void* buf, size_t n, long long offset)
    {
    s->note_placed(offset, n);
    if (s->meta_async && s->dev)
        {
        device_pipeline_write_host(s->dev, buf, n, offset);
        return 0;
        }
    return pwrite_full(s->fd, buf, n, offset);
    }

// pgsd_flush_name_buffer, pgsd.c:1216-1319
int flush_name_buffer(Impl* s)
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
int flush_write_buffer(Impl* s)
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

static bool check_eof_mode()
    {
    static const bool on = getenv("PGSD_CHECK_EOF") != nullptr;
    return on;
    }

// pgsd_expand_file_index, pgsd.c:965-1091
int expand_file_index(Impl* s, size_t size_required, int* local_rc)
    {
    size_t size_old = s->header.index_allocated_entries;
    size_t size_new = size_old * 2;
    while (size_new <= size_required)
        size_new *= 2;

    // The new block goes to the TRUE end of the file (MPI_File_get_size, pgsd.c:1015, once every rank's data is in the
    // file).  Where that end will be is known without waiting for the data: every rank has kept the end of the furthest
    // byte it has written or handed to its pipeline (Impl::placed_end), and the largest of them IS the file's size once
    // everything has landed.  One allgather; no barrier, no drain: a frame sealed asynchronously stays asynchronous when
    // the index moves (until round 5 the seal fell back to the synchronous one here -- a 90 ms hiccup for a simulation
    // with a backlog, examples/dump_writer.hip).
    std::vector<uint64_t> all;
    int rc = s->allgather_u64((uint64_t)s->placed_end, all);
    if (rc != PGSD_SUCCESS)
        return rc;
    uint64_t eof = 0;
    for (uint64_t e : all)
        eof = std::max(eof, e);
    long long new_loc = (long long)eof;
    long long old_loc = (long long)s->header.index_location;
    size_t old_bytes = size_old * sizeof(pgsd_index_entry);
    size_t new_bytes = size_new * sizeof(pgsd_index_entry);

    // PGSD_CHECK_EOF (tests): the old way beside the new one.  do_flush has drained every rank's pipeline (it keeps the
    // synchronous fall-back in this mode), so rank 0's fstat is the reference's MPI_File_get_size, and the old block can
    // be read back and compared with the in-memory mirror that is copied below in its place.
    if (check_eof_mode())
        {
        // (the allgather above was the point every rank reached with its writes done; the others wait in the barrier
        // below while rank 0 looks)
        if (s->rank == 0)
            {
            const long long size = io_file_size(s->fd);
            std::vector<char> on_disk(old_bytes, 0);
            pread_some(s->fd, on_disk.data(), old_bytes, old_loc);
            if (size != new_loc || memcmp(on_disk.data(), s->file_index.data(), old_bytes) != 0)
                {
                set_last_error("PGSD_CHECK_EOF: index relocation: computed end of file " + std::to_string(new_loc)
                               + ", fstat " + std::to_string(size)
                               + (memcmp(on_disk.data(), s->file_index.data(), old_bytes) != 0 ? "; the index mirror differs from the block on disk" : ""));
                fprintf(stderr, "%s\n", last_error());
                *local_rc = PGSD_ERROR_FILE_CORRUPT;
                }
            }
        if (s->P > 1)
            s->n_collectives++;
        int brc = comm_barrier(s->comm);
        if (brc != PGSD_SUCCESS)
            return brc;
        }

    if (s->rank == 0)
        {
        // the old block -- its in-memory mirror: entries of asynchronously sealed frames may still be on their way to
        // the file -- then zeros (pgsd.c:1021-1062 copies through the file)
        std::vector<char> block(new_bytes, 0);
        memcpy(block.data(), s->file_index.data(), old_bytes);
        if (meta_pwrite(s, block.data(), new_bytes, new_loc) != 0)
            *local_rc = PGSD_ERROR_IO;
        }
    s->header.index_location = (uint64_t)new_loc;
    s->file_size = new_loc + (long long)new_bytes;
    s->header.index_allocated_entries = size_new;
    s->note_placed(new_loc, new_bytes); // (every rank: the block is part of the file whoever writes it)
    if (s->rank == 0)
        if (meta_pwrite(s, &s->header, sizeof(s->header), 0) != 0)
            *local_rc = PGSD_ERROR_IO;

    // the in-memory mirror is the old block plus zeros; its used size is found the way
    // pgsd_index_buffer_map finds it after re-reading (pgsd.c:661-704, 1083)
    pgsd_index_entry zero;
    memset(&zero, 0, sizeof(zero));
    s->file_index.resize(size_new, zero);
    s->file_index_size = used_entries(s->file_index);
    return PGSD_SUCCESS;
    }

// pgsd_flush, pgsd.c:1955-2070.  sync_point: the call must leave every rank's bytes of the sealed
// frames in the file and every rank with the same verdict (pgsd_flush, pgsd_close, reads; and
// pgsd_end_frame unless the frame exchange is batched).
int do_flush(Impl* s, bool async, bool sync_point)
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
    // the background -- also when the on-disk index must move: the file's true end is known
    // from the ranks' placements (expand_file_index).  Only the PGSD_CHECK_EOF mode of the
    // tests keeps the old synchronous fall-back, to compare that end with fstat's.
    if (async && check_eof_mode() && s->pending <= s->frame_index.size())
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
bool metadata_pending(const Impl* s)
    {
    bool work = !s->queue.empty() || s->frame_n_names > 0 || !s->buffer_index.empty() || !s->frame_index.empty()
                || s->dirty_data;
    for (uint64_t b : s->wb_sizes)
        work = work || b > 0;
    return work;
    }

// local: this rank's rows of the asynchronously sealed frames reach the file (documented in pgsd.h)
int drain_own_copies(Impl* s)
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

// chunks that were staged ahead (pgsd_stage_chunks_device) and never written: their packed bytes go nowhere
void release_early(Impl* s)
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

int do_end_frame(Impl* s, bool async)
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
        s->fd = io_open(fname, O_RDWR | O_CREAT | (exclusive_create ? O_EXCL : 0), 0644);
        if (s->fd < 0)
            rc = PGSD_ERROR_IO;
        else
            rc = initialize_file(s->fd, application, schema, schema_version);
        }
    rc = agree_status(s, rc);
    if (rc == PGSD_SUCCESS && s->rank != 0)
        {
        s->fd = io_open(fname, O_RDWR, 0);
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
    s->fd = io_open(fname, flags == PGSD_OPEN_READONLY ? O_RDONLY : O_RDWR, 0);
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
        if (rc != PGSD_SUCCESS && rc != PGSD_ERROR_COMM)
            {
            publish(handle, s);
            return rc;
            }
        // PGSD_ERROR_COMM: the communicator is gone for good (a peer that never came, an aborted RCCL communicator,
        // ranks that disagree about the file) -- no later call could flush either.  The handle is ABANDONED: this
        // rank's own copies are drained, its descriptor, threads, pinned slabs and arenas are released, and the
        // error is returned (frames whose index rank 0 could not commit are not in the file's index).
        }
    const int flush_rc = rc;
    int fd = s->fd;
    s->fd = -1;
    destroy_impl(s);
    handle->impl = NULL;
    handle->file_index.data = NULL;
    handle->file_names.data.data = NULL;
    handle->fd = -1;
    if (io_close(fd) != 0)
        return flush_rc != PGSD_SUCCESS ? flush_rc : PGSD_ERROR_IO;
    return flush_rc;
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

extern "C" int pgsd_get_exchange_stats(struct pgsd_handle* handle, struct pgsd_exchange_stats* out, int reset)
    try
    {
    Impl* s = impl_of(handle);
    if (!s || !out)
        return PGSD_ERROR_INVALID_ARGUMENT;
    out->collectives = s->n_collectives; // since open: never reset
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
