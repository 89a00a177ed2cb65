// pgsd_read.cpp -- lookups and reads (pgsd_find_chunk pgsd.c:2295-2434, pgsd_read_chunk :2436-2537,
// pgsd_find_matching_chunk_name :2557-2641) and their device twin (file -> pinned slabs -> HBM -> unpack kernel).
// The index is replicated, so a lookup is valid on every rank; what a read flushes first is decided here
// (pgsd_set_local_reads).
#include "pgsd_file_impl.hpp"

namespace pgsd_amd
    {
int flush_for_lookup(Impl* s)
    {
    if (s->flags == PGSD_OPEN_READONLY)
        return PGSD_SUCCESS;
    // pgsd_set_local_reads covers the lookup that precedes the read: a frame that wrote buffered small
    // chunks only -- everything else elided -- leaves metadata pending behind pgsd_end_frame (pgsd.c:1941-1950), and
    // a rank that then looks one of frame 0's chunks up ALONE (it is the only one that compares that array) must not
    // start the collective flush.  It sees what the last flush committed; chunks of frames still pending are
    // not found until then.  (One rank: the flush is nobody else's business and runs as ever -- a one-rank file stays
    // the reference's byte for byte with local reads on, tests/test_product_golden.py.)  A lookup that MISSES while
    // metadata is pending says so in pgsd_last_error_string() (note_pending_miss below).
    if (metadata_pending(s) && (!s->local_reads || s->P == 1))
        return do_flush(s);
    return drain_own_copies(s);
    }

// What a READ needs before it touches the file: the reference's flush (collective) -- or, with
// pgsd_set_local_reads, only this rank's own asynchronous copies in place.
int flush_for_read(Impl* s)
    {
    if (s->flags == PGSD_OPEN_READONLY)
        return PGSD_SUCCESS;
    if (!s->local_reads)
        return do_flush(s);
    return drain_own_copies(s);
    }
    } // namespace pgsd_amd

using namespace pgsd_amd;

// ============================================================================ C ABI

// A LOCAL lookup (pgsd_set_local_reads, several ranks) that finds nothing while replicated metadata is still pending
// may have missed a chunk the next collective flush commits: the caller is told in pgsd_last_error_string()
// (callers of local_reads got the behaviour with no signal).
static void note_pending_miss(const Impl* s, const char* what, const char* name)
    {
    if (s->flags != PGSD_OPEN_READONLY && s->local_reads && s->P > 1 && metadata_pending(s))
        set_last_error(std::string(what) + " '" + name
                       + "': not found by a LOCAL lookup (pgsd_set_local_reads) while names / index entries of sealed "
                         "frames are pending: they are seen after the next collective pgsd_flush");
    }

static const pgsd_index_entry* find_chunk(Impl* s, uint64_t frame, const char* name)
    {
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
    const pgsd_index_entry* e = find_chunk(s, frame, name);
    if (!e)
        note_pending_miss(s, "pgsd_find_chunk", name);
    return e;
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
