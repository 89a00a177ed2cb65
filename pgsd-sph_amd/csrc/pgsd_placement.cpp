// pgsd_placement.cpp -- where the bytes of a chunk go (pgsd_write_chunk, pgsd.c:2072-2259, and its device twins):
// name -> id, the size exchange of a chunk (per chunk, batched per frame, or none with a declared partition), the
// reference's buffered-or-direct decision replayed in call order, the frame queue and its resolution, and the device
// write path (fused pack launches, staged chunks and their comparison).  The container itself is pgsd_container.cpp.
#include "pgsd_file_impl.hpp"

namespace pgsd_amd
    {
// name -> id; new names get the next id in first-seen order (pgsd.c:2111-2133, 1340-1404)
int name_to_id(Impl* s, const char* name, uint16_t* id)
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

// Local argument checks of pgsd_write_chunk, pgsd.c:2090-2105.  The reference returns from them
// before its first collective, which leaves the other ranks waiting; here the verdict travels
// with the size exchange below, so every rank returns the same error.
int check_chunk_args(const Impl* s, const char* name, uint64_t N, uint32_t M, uint8_t flags, bool have_data)
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
int exchange_counts(Impl* s, uint64_t mine, int local_rc, std::vector<uint64_t>& all)
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
void auto_partition(const Impl* s, const std::vector<uint64_t>& sizes, uint64_t unit, uint32_t M,
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
int place_chunk(Impl* s, const char* name, uint32_t type, uint64_t N, uint32_t M, uint64_t N_global,
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
void remember_failure(Impl* s, int rc, int err)
    {
    if (s->sticky_rc == PGSD_SUCCESS)
        {
        s->sticky_rc = rc;
        s->sticky_errno = err;
        }
    }

int ensure_device(Impl* s)
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
int deliver_chunk(Impl* s, Queued& q, const Placement& pl, bool skip)
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
            {
            s->note_placed(pl.file_offset, pl.size);
            rc = device_pipeline_commit(s->dev, q.ticket, q.ticket_index, pl.file_offset, nullptr, &err);
            }
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
    s->note_placed(pl.file_offset, pl.size);
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
int resolve_queue(Impl* s)
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
int trusted_place(Impl* s, const char* name, uint32_t type, uint64_t N, uint32_t M, uint64_t* N_global,
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
                s->note_placed(pl.file_offset, pl.size);
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
            s->note_placed(pl.file_offset, pl.size);
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
                if (!skip)
                    s->note_placed(pl.file_offset, pl.size);
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
                if (!skip)
                    s->note_placed(pl.file_offset, pl.size);
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

extern "C" int pgsd_device_of(struct pgsd_handle* handle)
    try
    {
    Impl* s = impl_of(handle);
    if (!s)
        return PGSD_ERROR_INVALID_ARGUMENT;
    const int rc = ensure_device(s);
    return rc != PGSD_SUCCESS ? rc : device_pipeline_device(s->dev);
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
