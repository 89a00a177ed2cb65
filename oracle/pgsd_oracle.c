/* pgsd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See pgsd_oracle.h.
 *
 * CPU restatement of /root/reference/pgsd/pgsd/pgsd.c (PGSD 3.2.0) for P simulated ranks.
 * Each function cites the reference lines it follows.  Quirks of the reference that show
 * up in the bytes of the file are reproduced on purpose and marked QUIRK.
 */
#define _GNU_SOURCE
#include "pgsd_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

/* constants: pgsd.c:54-102 */
#define O_MAGIC 0x65DF65DF65DF65DFull
#define O_INITIAL_INDEX_SIZE 128
#define O_INITIAL_NAME_BUFFER_SIZE 1024
#define O_INITIAL_FRAME_INDEX_SIZE 16
#define O_INITIAL_WRITE_BUFFER_SIZE 1024
#define O_DEFAULT_MAXIMUM_WRITE_BUFFER_SIZE (64ull * 1024 * 1024)
#define O_DEFAULT_INDEX_ENTRIES_TO_BUFFER (256ull * 1024)
#define O_CURRENT_FILE_VERSION 2
#define O_NAME_SIZE 64

enum { O_READWRITE = 1, O_READONLY = 2, O_APPEND_FLAG = 3 };

struct o_bytes /* pgsd_byte_buffer, pgsd.h:262-272 */
    {
    char* data;
    size_t size, reserved;
    };

struct o_index /* pgsd_index_buffer, pgsd.h:239-255 */
    {
    struct oracle_index_entry* data;
    size_t size, reserved;
    };

struct pgsd_oracle
    {
    int fd;
    int nprocs;
    struct oracle_header header;
    struct o_index file_index, frame_index, buffer_index; /* root only */
    struct o_bytes* write_buffer;                          /* one per rank */
    struct o_bytes file_names, frame_names;                /* root only */
    size_t file_n_names, frame_n_names;
    uint64_t cur_frame;
    long long file_size;
    int open_flags;
    uint64_t pending_index_entries;
    uint64_t maximum_write_buffer_size;
    uint64_t index_entries_to_buffer;
    };

uint32_t oracle_make_version(unsigned int major, unsigned int minor)
    {
    return major << 16 | minor; /* pgsd.c:1705-1708 */
    }

size_t oracle_sizeof_type(int type)
    {
    static const size_t s[] = {0, 1, 2, 4, 8, 1, 2, 4, 8, 4, 8}; /* pgsd.c:2539-2555 */
    return (type >= 1 && type <= 10) ? s[type] : 0;
    }

/* ---- small helpers: positional IO standing in for MPI_File_{read,write}_at ---- */
static int o_pwrite(int fd, const void* buf, size_t n, long long off)
    {
    const char* p = (const char*)buf;
    while (n > 0)
        {
        ssize_t w = pwrite(fd, p, n, off);
        if (w < 0)
            {
            if (errno == EINTR)
                continue;
            return -1;
            }
        p += w;
        off += w;
        n -= (size_t)w;
        }
    return 0;
    }

/* short reads leave the tail untouched (MPI_File_read_at returns fewer bytes at EOF) */
static void o_pread(int fd, void* buf, size_t n, long long off)
    {
    char* p = (char*)buf;
    while (n > 0)
        {
        ssize_t r = pread(fd, p, n, off);
        if (r <= 0)
            {
            if (r < 0 && errno == EINTR)
                continue;
            return;
            }
        p += r;
        off += r;
        n -= (size_t)r;
        }
    }

static long long o_eof(int fd)
    {
    struct stat st;
    if (fstat(fd, &st) != 0)
        return -1;
    return (long long)st.st_size;
    }

/* pgsd_byte_buffer_allocate, pgsd.c:460-477 */
static int bytes_allocate(struct o_bytes* b, size_t reserve)
    {
    if (b->data || reserve == 0 || b->reserved != 0 || b->size != 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    b->data = (char*)calloc(reserve, 1);
    if (!b->data)
        return ORACLE_ERROR_MEMORY_ALLOCATION_FAILED;
    b->reserved = reserve;
    return ORACLE_SUCCESS;
    }

/* pgsd_byte_buffer_append, pgsd.c:490-525 (doubling rule decides namelist relocation) */
static int bytes_append(struct o_bytes* b, const char* data, size_t size)
    {
    if (b->data == NULL || size == 0 || b->reserved == 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    if (b->size + size > b->reserved)
        {
        size_t new_reserved = b->reserved * 2;
        while (b->size + size >= new_reserved)
            new_reserved *= 2;
        char* nd = (char*)realloc(b->data, new_reserved);
        if (!nd)
            return ORACLE_ERROR_MEMORY_ALLOCATION_FAILED;
        b->data = nd;
        memset(b->data + (b->size + size), 0, new_reserved - (b->size + size));
        b->reserved = new_reserved;
        }
    memcpy(b->data + b->size, data, size);
    b->size += size;
    return ORACLE_SUCCESS;
    }

static void bytes_free(struct o_bytes* b)
    {
    free(b->data);
    memset(b, 0, sizeof(*b));
    }

/* pgsd_index_buffer_allocate, pgsd.c:562-587 */
static int index_allocate(struct o_index* b, size_t reserve)
    {
    if (b->data || reserve == 0 || b->reserved != 0 || b->size != 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    b->data = (struct oracle_index_entry*)calloc(reserve, sizeof(struct oracle_index_entry));
    if (!b->data)
        return ORACLE_ERROR_MEMORY_ALLOCATION_FAILED;
    b->reserved = reserve;
    return ORACLE_SUCCESS;
    }

static void index_free(struct o_index* b)
    {
    free(b->data);
    memset(b, 0, sizeof(*b));
    }

/* pgsd_index_buffer_add, pgsd.c:764-797 */
static int index_add(struct o_index* b, struct oracle_index_entry** entry)
    {
    if (b->reserved == 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    if (b->size == b->reserved)
        {
        size_t nr = b->reserved * 2;
        struct oracle_index_entry* nd
            = (struct oracle_index_entry*)realloc(b->data, sizeof(struct oracle_index_entry) * nr);
        if (!nd)
            return ORACLE_ERROR_MEMORY_ALLOCATION_FAILED;
        b->data = nd;
        memset(b->data + b->reserved, 0, sizeof(struct oracle_index_entry) * (nr - b->reserved));
        b->reserved = nr;
        }
    *entry = b->data + b->size;
    b->size++;
    return ORACLE_SUCCESS;
    }

/* pgsd_cmp_index_entry, pgsd.c:799-833 */
static int cmp_entry(const struct oracle_index_entry* a, const struct oracle_index_entry* b)
    {
    if (a->frame < b->frame)
        return -1;
    if (a->frame > b->frame)
        return 1;
    if (a->id < b->id)
        return -1;
    if (a->id > b->id)
        return 1;
    return 0;
    }

/* heap sort exactly as pgsd.c:839-953 (not stable: order of equal keys is part of the bytes) */
static void heap_swap(struct o_index* b, size_t x, size_t y)
    {
    struct oracle_index_entry t = b->data[x];
    b->data[x] = b->data[y];
    b->data[y] = t;
    }

static void heap_shift_down(struct o_index* b, size_t start, size_t end)
    {
    size_t root = start;
    while (2 * root + 1 <= end)
        {
        size_t child = 2 * root + 1;
        size_t swap = root;
        if (cmp_entry(b->data + swap, b->data + child) < 0)
            swap = child;
        if (child + 1 <= end && cmp_entry(b->data + swap, b->data + child + 1) < 0)
            swap = child + 1;
        if (swap == root)
            return;
        heap_swap(b, root, swap);
        root = swap;
        }
    }

static void index_sort(struct o_index* b)
    {
    if (b->size <= 1)
        return;
    ssize_t start = (ssize_t)((b->size - 1 - 1) / 2);
    while (start >= 0)
        {
        heap_shift_down(b, (size_t)start, b->size - 1);
        start--;
        }
    size_t end = b->size - 1;
    while (end > 0)
        {
        heap_swap(b, end, 0);
        end--;
        heap_shift_down(b, 0, end);
        }
    }

/* name -> id.  The reference keeps a djb2 hash map (pgsd.c:224-405); ids are assigned in
   first-seen order and lookups are exact string matches, so a linear scan over the name
   bytes is the same function. */
static uint16_t name_find_in(const struct o_bytes* names, size_t n_names, int v1, const char* str,
                             uint16_t base)
    {
    size_t pos = 0;
    for (size_t i = 0; i < n_names && pos < names->reserved; i++)
        {
        const char* nm = names->data + pos;
        if (strcmp(nm, str) == 0)
            return (uint16_t)(base + i);
        pos += v1 ? O_NAME_SIZE : strlen(nm) + 1;
        }
    return UINT16_MAX;
    }

static uint16_t name_find(pgsd_oracle* o, const char* str)
    {
    int v1 = o->header.pgsd_version < oracle_make_version(2, 0);
    char key[O_NAME_SIZE];
    if (v1)
        {
        /* v1 names are inserted truncated to 63 bytes (pgsd.c:1371-1378) but looked up with
           the caller's string (pgsd.c:2113): a longer name never matches. */
        (void)key;
        }
    uint16_t id = name_find_in(&o->file_names, o->file_n_names, v1, str, 0);
    if (id != UINT16_MAX)
        return id;
    return name_find_in(&o->frame_names, o->frame_n_names, v1, str, (uint16_t)o->file_n_names);
    }

/* pgsd_is_entry_valid, pgsd.c:414-450 */
static int entry_valid(pgsd_oracle* o, const struct oracle_index_entry* e)
    {
    if (oracle_sizeof_type(e->type) == 0)
        return 0;
    size_t size = e->N * e->M * oracle_sizeof_type(e->type);
    if ((uint64_t)(e->location + size) > (uint64_t)o->file_size)
        return 0;
    if (e->frame >= o->header.index_allocated_entries)
        return 0;
    if (e->id >= (o->file_n_names + o->frame_n_names))
        return 0;
    if (e->flags != 0)
        return 0;
    return 1;
    }

/* pgsd_index_buffer_map (read variant), pgsd.c:602-707 */
static int index_map(pgsd_oracle* o)
    {
    struct o_index* b = &o->file_index;
    if (b->data || b->reserved != 0 || b->size != 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    if (o->header.index_location + sizeof(struct oracle_index_entry) * o->header.index_allocated_entries
        > (uint64_t)o->file_size)
        return ORACLE_ERROR_FILE_CORRUPT;
    int rv = index_allocate(b, o->header.index_allocated_entries);
    if (rv != ORACLE_SUCCESS)
        return rv;
    o_pread(o->fd, b->data, sizeof(struct oracle_index_entry) * o->header.index_allocated_entries,
            (long long)o->header.index_location);

    if (b->data[0].location != 0 && !entry_valid(o, &b->data[0]))
        return ORACLE_ERROR_FILE_CORRUPT;
    if (b->data[0].location == 0)
        {
        b->size = 0;
        }
    else
        {
        size_t L = 0, R = b->reserved;
        do
            {
            size_t m = (L + R) / 2;
            if (b->data[m].location != 0
                && (!entry_valid(o, &b->data[m]) || b->data[m].frame < b->data[L].frame))
                return ORACLE_ERROR_FILE_CORRUPT;
            if (b->data[m].location != 0)
                L = m;
            else
                R = m;
            } while ((R - L) > 1);
        b->size = R;
        }
    return ORACLE_SUCCESS;
    }

/* pgsd_expand_file_index, pgsd.c:965-1091 (root only) */
static int expand_file_index(pgsd_oracle* o, size_t size_required)
    {
    if (o->open_flags == O_READONLY)
        return ORACLE_ERROR_FILE_MUST_BE_WRITABLE;
    size_t size_old = o->header.index_allocated_entries;
    size_t size_new = size_old * 2;
    while (size_new <= size_required)
        size_new *= 2;
    index_free(&o->file_index);

    uint64_t copy_buffer_size = O_DEFAULT_INDEX_ENTRIES_TO_BUFFER * sizeof(struct oracle_index_entry);
    if (copy_buffer_size > size_old * sizeof(struct oracle_index_entry))
        copy_buffer_size = size_old * sizeof(struct oracle_index_entry);
    char* buf = (char*)malloc(copy_buffer_size);
    if (!buf)
        return ORACLE_ERROR_MEMORY_ALLOCATION_FAILED;

    /* QUIRK: the new index goes to the TRUE end of file (MPI_File_get_size, pgsd.c:1015),
       not to handle->file_size. */
    long long new_index_location = o_eof(o->fd);
    long long old_index_location = (long long)o->header.index_location;
    size_t total = 0;
    size_t old_bytes = size_old * sizeof(struct oracle_index_entry);
    while (total < old_bytes)
        {
        size_t n = copy_buffer_size;
        if (old_bytes - total < copy_buffer_size)
            n = old_bytes - total;
        o_pread(o->fd, buf, n, old_index_location + (long long)total);
        o_pwrite(o->fd, buf, n, new_index_location + (long long)total);
        total += n;
        }
    memset(buf, 0, copy_buffer_size);
    size_t new_bytes = size_new * sizeof(struct oracle_index_entry);
    while (total < new_bytes)
        {
        size_t n = copy_buffer_size;
        if (new_bytes - total < copy_buffer_size)
            n = new_bytes - total;
        o_pwrite(o->fd, buf, n, new_index_location + (long long)total);
        total += n;
        }
    free(buf);

    o->header.index_location = (uint64_t)new_index_location;
    o->file_size = new_index_location + (long long)total;
    o->header.index_allocated_entries = size_new;
    o_pwrite(o->fd, &o->header, sizeof(o->header), 0);
    return index_map(o);
    }

/* pgsd_flush_write_buffer, pgsd.c:1108-1201 */
static int flush_write_buffer(pgsd_oracle* o)
    {
    const int P = o->nprocs;
    /* pgsd.c:1126 MPI_Allgather of write_buffer.size == the write_buffer[] array itself */

    /* pgsd.c:1128-1133: per-rank early return.  buffer_index exists on root only, so for
       r > 0 the test is write_buffer[r].size == 0. */
    int n_return = 0;
    for (int r = 0; r < P; r++)
        {
        size_t bidx = (r == 0) ? o->buffer_index.size : 0;
        if (o->write_buffer[r].size == 0 && bidx == 0)
            n_return++;
        }
    if (n_return == P)
        return ORACLE_SUCCESS;
    if (n_return != 0)
        return ORACLE_ERROR_REFERENCE_WOULD_HANG; /* some ranks skip bcast_file_size, pgsd.c:1144 */

    if (o->write_buffer[0].size > 0 && o->buffer_index.size == 0)
        return ORACLE_ERROR_INVALID_ARGUMENT; /* pgsd.c:1135-1143 */

    /* pgsd.c:1145-1154: rank r writes its buffer at file_size + sum_{j<r} size_j */
    long long offset_root = o->file_size;
    long long offset = o->file_size;
    size_t total = 0;
    for (int r = 0; r < P; r++)
        {
        /* QUIRK: every rank appends its own copy of the small chunks, so a P-rank file
           carries P copies; the index points at rank 0's (pgsd.c:2191-2201, 1154). */
        if (o_pwrite(o->fd, o->write_buffer[r].data, o->write_buffer[r].size, offset) != 0)
            return ORACLE_ERROR_IO;
        offset += (long long)o->write_buffer[r].size;
        total += o->write_buffer[r].size;
        o->write_buffer[r].size = 0;
        }
    o->file_size += (long long)total; /* pgsd.c:1162-1171 */

    /* pgsd.c:1175-1194 */
    for (size_t i = 0; i < o->buffer_index.size; i++)
        {
        struct oracle_index_entry* ne;
        int rv = index_add(&o->frame_index, &ne);
        if (rv != ORACLE_SUCCESS)
            return rv;
        *ne = o->buffer_index.data[i];
        ne->location += offset_root;
        }
    o->buffer_index.size = 0;
    return ORACLE_SUCCESS;
    }

/* pgsd_flush_name_buffer, pgsd.c:1216-1319 */
static int flush_name_buffer(pgsd_oracle* o)
    {
    if (o->frame_n_names == 0)
        return ORACLE_SUCCESS;
    if (o->frame_names.size == 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    size_t old_reserved = o->file_names.reserved;
    size_t old_size = o->file_names.size;

    int rv = bytes_append(&o->file_names, o->frame_names.data, o->frame_names.size);
    if (rv != ORACLE_SUCCESS)
        return rv;
    o->file_n_names += o->frame_n_names;
    o->frame_n_names = 0;
    o->frame_names.size = 0;
    memset(o->frame_names.data, 0, o->frame_names.reserved);

    if (o->file_names.reserved % O_NAME_SIZE != 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;

    if (o->file_names.reserved > old_reserved)
        {
        /* relocate: whole list to the end of the file + header rewrite, pgsd.c:1284-1300 */
        long long offset = o->file_size;
        if (o_pwrite(o->fd, o->file_names.data, o->file_names.reserved, offset) != 0)
            return ORACLE_ERROR_IO;
        o->file_size += (long long)o->file_names.reserved;
        o->header.namelist_location = (uint64_t)offset;
        o->header.namelist_allocated_entries = o->file_names.reserved / O_NAME_SIZE;
        if (o_pwrite(o->fd, &o->header, sizeof(o->header), 0) != 0)
            return ORACLE_ERROR_IO;
        }
    else
        {
        /* in place: tail [old_size, reserved), pgsd.c:1304-1306 */
        if (o_pwrite(o->fd, o->file_names.data + old_size, o->file_names.reserved - old_size,
                     (long long)o->header.namelist_location + (long long)old_size)
            != 0)
            return ORACLE_ERROR_IO;
        }
    return ORACLE_SUCCESS;
    }

/* pgsd_append_name, pgsd.c:1340-1404 */
static int append_name(pgsd_oracle* o, uint16_t* id, const char* name)
    {
    if (o->open_flags == O_READONLY)
        return ORACLE_ERROR_FILE_MUST_BE_WRITABLE;
    if (o->file_n_names + o->frame_n_names == UINT16_MAX)
        return ORACLE_ERROR_NAMELIST_FULL;
    *id = (uint16_t)(o->file_n_names + o->frame_n_names);
    if (o->header.pgsd_version < oracle_make_version(2, 0))
        {
        char name_v1[O_NAME_SIZE];
        strncpy(name_v1, name, O_NAME_SIZE - 1);
        name_v1[O_NAME_SIZE - 1] = 0;
        bytes_append(&o->frame_names, name_v1, O_NAME_SIZE);
        }
    else
        {
        bytes_append(&o->frame_names, name, strlen(name) + 1);
        }
    o->frame_n_names++;
    return ORACLE_SUCCESS;
    }

/* pgsd_initialize_file, pgsd.c:1414-1474 */
static int initialize_file(int fd, const char* application, const char* schema, uint32_t schema_version)
    {
    if (ftruncate(fd, 0) != 0)
        return ORACLE_ERROR_IO;
    struct oracle_header h;
    memset(&h, 0, sizeof(h));
    h.magic = O_MAGIC;
    h.pgsd_version = oracle_make_version(O_CURRENT_FILE_VERSION, 0);
    strncpy(h.application, application, sizeof(h.application) - 1);
    h.application[sizeof(h.application) - 1] = 0;
    strncpy(h.schema, schema, sizeof(h.schema) - 1);
    h.schema[sizeof(h.schema) - 1] = 0;
    h.schema_version = schema_version;
    h.index_location = sizeof(h);
    h.index_allocated_entries = O_INITIAL_INDEX_SIZE;
    h.namelist_location = h.index_location + sizeof(struct oracle_index_entry) * h.index_allocated_entries;
    h.namelist_allocated_entries = O_INITIAL_NAME_BUFFER_SIZE / O_NAME_SIZE;
    if (o_pwrite(fd, &h, sizeof(h), 0) != 0)
        return ORACLE_ERROR_IO;
    char zeros[O_INITIAL_INDEX_SIZE * sizeof(struct oracle_index_entry)];
    memset(zeros, 0, sizeof(zeros));
    if (o_pwrite(fd, zeros, sizeof(zeros), (long long)sizeof(h)) != 0)
        return ORACLE_ERROR_IO;
    if (o_pwrite(fd, zeros, O_INITIAL_NAME_BUFFER_SIZE, (long long)(sizeof(h) + sizeof(zeros))) != 0)
        return ORACLE_ERROR_IO;
    return ORACLE_SUCCESS;
    }

/* pgsd_initialize_handle, pgsd.c:1484-1703 */
static int initialize_handle(pgsd_oracle* o)
    {
    memset(&o->header, 0, sizeof(o->header));
    o_pread(o->fd, &o->header, sizeof(o->header), 0);
    if (o->header.magic != O_MAGIC)
        return ORACLE_ERROR_NOT_A_PGSD_FILE;
    if (o->header.pgsd_version < oracle_make_version(1, 0)
        && o->header.pgsd_version != oracle_make_version(0, 3))
        return ORACLE_ERROR_INVALID_PGSD_FILE_VERSION;
    if (o->header.pgsd_version >= oracle_make_version(3, 0))
        return ORACLE_ERROR_INVALID_PGSD_FILE_VERSION;

    o->file_size = o_eof(o->fd);
    if (o->header.namelist_location + (O_NAME_SIZE * o->header.namelist_allocated_entries)
        > (uint64_t)o->file_size)
        return ORACLE_ERROR_FILE_CORRUPT;

    size_t namelist_n_bytes = O_NAME_SIZE * o->header.namelist_allocated_entries;
    int rv = bytes_allocate(&o->file_names, namelist_n_bytes);
    if (rv != ORACLE_SUCCESS)
        return rv;
    o_pread(o->fd, o->file_names.data, namelist_n_bytes, (long long)o->header.namelist_location);
    if (o->file_names.data[o->file_names.reserved - 1] != 0)
        return ORACLE_ERROR_FILE_CORRUPT;

    size_t name_start = 0;
    o->file_n_names = 0;
    while (name_start < o->file_names.reserved)
        {
        char* name = o->file_names.data + name_start;
        if (name[0] == 0)
            break;
        o->file_n_names++;
        if (o->header.pgsd_version < oracle_make_version(2, 0))
            name_start += O_NAME_SIZE;
        else
            name_start += strnlen(name, o->file_names.reserved - name_start) + 1;
        }
    o->file_names.size = name_start;

    rv = index_map(o);
    if (rv != ORACLE_SUCCESS)
        return rv;
    if (o->file_index.size == 0)
        o->cur_frame = 0;
    else
        o->cur_frame = o->file_index.data[o->file_index.size - 1].frame + 1;

    if (o->open_flags != O_READONLY)
        {
        rv = index_allocate(&o->frame_index, O_INITIAL_FRAME_INDEX_SIZE);
        if (rv != ORACLE_SUCCESS)
            return rv;
        rv = index_allocate(&o->buffer_index, O_INITIAL_FRAME_INDEX_SIZE);
        if (rv != ORACLE_SUCCESS)
            return rv;
        for (int r = 0; r < o->nprocs; r++)
            {
            rv = bytes_allocate(&o->write_buffer[r], O_INITIAL_WRITE_BUFFER_SIZE);
            if (rv != ORACLE_SUCCESS)
                return rv;
            }
        o->frame_n_names = 0;
        rv = bytes_allocate(&o->frame_names, O_NAME_SIZE);
        if (rv != ORACLE_SUCCESS)
            return rv;
        }
    o->pending_index_entries = 0;
    o->maximum_write_buffer_size = O_DEFAULT_MAXIMUM_WRITE_BUFFER_SIZE;
    o->index_entries_to_buffer = O_DEFAULT_INDEX_ENTRIES_TO_BUFFER;
    return ORACLE_SUCCESS;
    }

static pgsd_oracle* o_new(int nprocs)
    {
    pgsd_oracle* o = (pgsd_oracle*)calloc(1, sizeof(pgsd_oracle));
    o->fd = -1;
    o->nprocs = nprocs;
    o->write_buffer = (struct o_bytes*)calloc((size_t)nprocs, sizeof(struct o_bytes));
    return o;
    }

static void o_delete(pgsd_oracle* o)
    {
    if (!o)
        return;
    if (o->fd >= 0)
        close(o->fd);
    index_free(&o->file_index);
    index_free(&o->frame_index);
    index_free(&o->buffer_index);
    for (int r = 0; r < o->nprocs; r++)
        bytes_free(&o->write_buffer[r]);
    free(o->write_buffer);
    bytes_free(&o->file_names);
    bytes_free(&o->frame_names);
    free(o);
    }

/* pgsd_create_and_open, pgsd.c:1710-1773 */
pgsd_oracle* oracle_create_and_open(const char* fname, int nprocs, const char* application,
                                    const char* schema, uint32_t schema_version, int flags,
                                    int exclusive_create, int* rc)
    {
    int dummy;
    if (!rc)
        rc = &dummy;
    if (nprocs < 1)
        {
        *rc = ORACLE_ERROR_INVALID_ARGUMENT;
        return NULL;
        }
    if (flags == O_READONLY)
        {
        *rc = ORACLE_ERROR_FILE_MUST_BE_WRITABLE;
        return NULL;
        }
    pgsd_oracle* o = o_new(nprocs);
    o->open_flags = flags;
    o->fd = open(fname, O_RDWR | O_CREAT | (exclusive_create ? O_EXCL : 0), 0644);
    if (o->fd < 0)
        {
        *rc = ORACLE_ERROR_IO;
        o_delete(o);
        return NULL;
        }
    *rc = initialize_file(o->fd, application, schema, schema_version);
    if (*rc == ORACLE_SUCCESS)
        *rc = initialize_handle(o);
    if (*rc != ORACLE_SUCCESS)
        {
        o_delete(o);
        return NULL;
        }
    return o;
    }

/* pgsd_open, pgsd.c:1775-1812 */
pgsd_oracle* oracle_open(const char* fname, int nprocs, int flags, int* rc)
    {
    int dummy;
    if (!rc)
        rc = &dummy;
    if (nprocs < 1)
        {
        *rc = ORACLE_ERROR_INVALID_ARGUMENT;
        return NULL;
        }
    pgsd_oracle* o = o_new(nprocs);
    o->open_flags = flags;
    o->fd = open(fname, flags == O_READONLY ? O_RDONLY : O_RDWR);
    if (o->fd < 0)
        {
        *rc = ORACLE_ERROR_IO;
        o_delete(o);
        return NULL;
        }
    *rc = initialize_handle(o);
    if (*rc != ORACLE_SUCCESS)
        {
        o_delete(o);
        return NULL;
        }
    return o;
    }

/* pgsd_flush, pgsd.c:1955-2070 */
int oracle_flush(pgsd_oracle* o)
    {
    if (!o)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    if (o->open_flags == O_READONLY)
        return ORACLE_ERROR_FILE_MUST_BE_WRITABLE;
    int rv = flush_name_buffer(o);
    if (rv != ORACLE_SUCCESS)
        return rv;
    rv = flush_write_buffer(o);
    if (rv != ORACLE_SUCCESS)
        return rv;

    if (o->pending_index_entries > o->frame_index.size)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    uint64_t to_write = o->frame_index.size - o->pending_index_entries;
    if (to_write > 0)
        {
        if ((o->file_index.size + to_write) > o->file_index.reserved)
            expand_file_index(o, o->file_index.size + to_write); /* return value ignored, pgsd.c:2015 */

        index_sort(&o->frame_index);
        long long write_pos = (long long)o->header.index_location
                              + (long long)(sizeof(struct oracle_index_entry) * o->file_index.size);
        /* QUIRK: all frame_index.size entries are written, pending ones included (pgsd.c:2032) */
        if (o_pwrite(o->fd, o->frame_index.data, sizeof(struct oracle_index_entry) * o->frame_index.size,
                     write_pos)
            != 0)
            return ORACLE_ERROR_IO;
        /* mirror into the in-memory file index (pgsd.c:2039-2042); bounded copy */
        size_t room = o->file_index.reserved - o->file_index.size;
        size_t ncopy = o->frame_index.size < room ? o->frame_index.size : room;
        memcpy(o->file_index.data + o->file_index.size, o->frame_index.data,
               sizeof(struct oracle_index_entry) * ncopy);
        o->file_index.size += to_write;

        /* QUIRK: every kept slot receives the same entry (no "+ i"), pgsd.c:2049-2057 */
        for (uint64_t i = 0; i < o->pending_index_entries; i++)
            o->frame_index.data[i] = o->frame_index.data[o->frame_index.size - o->pending_index_entries];
        o->frame_index.size = o->pending_index_entries;
        }
    return ORACLE_SUCCESS;
    }

/* pgsd_end_frame, pgsd.c:1916-1953 */
int oracle_end_frame(pgsd_oracle* o)
    {
    if (!o)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    if (o->open_flags == O_READONLY)
        return ORACLE_ERROR_FILE_MUST_BE_WRITABLE;
    o->cur_frame++;
    o->pending_index_entries = 0;
    if (o->frame_index.size > 0 || o->buffer_index.size > o->index_entries_to_buffer)
        return oracle_flush(o);
    return ORACLE_SUCCESS;
    }

/* pgsd_close, pgsd.c:1814-1914 */
int oracle_close(pgsd_oracle* o)
    {
    if (!o)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    int rv = ORACLE_SUCCESS;
    if (o->open_flags != O_READONLY)
        {
        rv = oracle_flush(o);
        if (rv != ORACLE_SUCCESS)
            return rv;
        }
    int fd = o->fd;
    o->fd = -1;
    o_delete(o);
    if (close(fd) != 0)
        return ORACLE_ERROR_IO;
    return ORACLE_SUCCESS;
    }

/* pgsd_write_chunk, pgsd.c:2072-2259 */
int oracle_write_chunk(pgsd_oracle* o, const char* name, int type, const uint64_t* N, uint32_t M,
                       uint64_t N_global, uint32_t M_global, const uint64_t* offset,
                       const uint64_t* global_size, bool all, uint8_t flags,
                       const void* const* data)
    {
    const int P = o->nprocs;
    (void)global_size; /* scaled at pgsd.c:2147-2151 and never used afterwards */
    for (int r = 0; r < P; r++)
        if (N[r] > 0 && data[r] == NULL)
            return ORACLE_ERROR_INVALID_ARGUMENT;
    if (M == 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    if (o->open_flags == O_READONLY)
        return ORACLE_ERROR_FILE_MUST_BE_WRITABLE;
    if (flags != 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;

    /* root: name -> id, new entry (pgsd.c:2111-2141) */
    uint16_t id = name_find(o, name);
    if (id == UINT16_MAX)
        {
        int rv = append_name(o, &id, name);
        if (rv != ORACLE_SUCCESS)
            return rv;
        if (id == UINT16_MAX)
            return ORACLE_ERROR_NAMELIST_FULL;
        }
    struct oracle_index_entry entry;
    memset(&entry, 0, sizeof(entry));
    entry.frame = o->cur_frame;
    entry.id = id;
    entry.type = (uint8_t)type;
    entry.N = N_global;
    entry.M = M_global;

    const size_t sz = oracle_sizeof_type(type);
    size_t maxsize = 0, sumsize = 0;
    for (int r = 0; r < P; r++)
        {
        size_t s = N[r] * M * sz;
        if (s > maxsize)
            maxsize = s; /* MPI_Allreduce MAX, pgsd.c:2157 */
        sumsize += s;    /* MPI_Allreduce SUM, pgsd.c:2242 */
        }

    if (maxsize < o->maximum_write_buffer_size && all == false)
        {
        /* BUFFERED path, pgsd.c:2160-2202 */
        int n_flush = 0;
        for (int r = 0; r < P; r++)
            if (N[r] * M * sz > (o->maximum_write_buffer_size - o->write_buffer[r].size))
                n_flush++;
        if (n_flush != 0 && n_flush != P)
            return ORACLE_ERROR_REFERENCE_WOULD_HANG;
        if (n_flush == P)
            flush_write_buffer(o); /* return value ignored, pgsd.c:2167 */

        entry.location = (int64_t)o->write_buffer[0].size;
        struct oracle_index_entry* ie;
        int rv = index_add(&o->buffer_index, &ie);
        if (rv != ORACLE_SUCCESS)
            return rv;
        *ie = entry;
        for (int r = 0; r < P; r++)
            {
            size_t s = N[r] * M * sz;
            if (s > 0)
                {
                rv = bytes_append(&o->write_buffer[r], (const char*)data[r], s);
                if (rv != ORACLE_SUCCESS)
                    return rv;
                }
            }
        }
    else
        {
        /* DIRECT path, pgsd.c:2203-2250 */
        struct oracle_index_entry* ie;
        int rv = index_add(&o->frame_index, &ie);
        if (rv != ORACLE_SUCCESS)
            return rv;
        *ie = entry;
        ie->location = o->file_size;
        for (int r = 0; r < P; r++)
            {
            if (all == true || r == 0)
                {
                long long loc = o->file_size + (long long)(offset[r] * sz);
                if (o_pwrite(o->fd, data[r], N[r] * M * sz, loc) != 0)
                    return ORACLE_ERROR_IO;
                }
            }
        /* QUIRK: file_size advances by the sum over ALL ranks even when only root wrote
           (all == false), leaving a hole (pgsd.c:2240-2249). */
        o->file_size += (long long)sumsize;
        }
    o->pending_index_entries++;
    return ORACLE_SUCCESS;
    }

uint64_t oracle_get_nframes(pgsd_oracle* o) { return o ? o->cur_frame : 0; }
uint64_t oracle_get_nnames(pgsd_oracle* o) { return o ? o->file_n_names : 0; }
long long oracle_get_file_size(pgsd_oracle* o) { return o ? o->file_size : 0; }
const struct oracle_header* oracle_get_header(pgsd_oracle* o) { return o ? &o->header : NULL; }
uint64_t oracle_get_maximum_write_buffer_size(pgsd_oracle* o) { return o ? o->maximum_write_buffer_size : 0; }
uint64_t oracle_get_index_entries_to_buffer(pgsd_oracle* o) { return o ? o->index_entries_to_buffer : 0; }

int oracle_set_maximum_write_buffer_size(pgsd_oracle* o, uint64_t size)
    {
    if (!o || size == 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    o->maximum_write_buffer_size = size;
    return ORACLE_SUCCESS;
    }

int oracle_set_index_entries_to_buffer(pgsd_oracle* o, uint64_t number)
    {
    if (!o || number == 0)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    o->index_entries_to_buffer = number;
    return ORACLE_SUCCESS;
    }

/* pgsd_find_chunk, pgsd.c:2295-2434 */
const struct oracle_index_entry* oracle_find_chunk(pgsd_oracle* o, uint64_t frame, const char* name)
    {
    if (!o || !name)
        return NULL;
    if (frame >= oracle_get_nframes(o))
        return NULL;
    if (o->open_flags != O_READONLY)
        if (oracle_flush(o) != ORACLE_SUCCESS)
            return NULL;
    uint16_t match_id = name_find(o, name);
    if (match_id == UINT16_MAX)
        return NULL;

    if (o->header.pgsd_version >= oracle_make_version(2, 0))
        {
        ssize_t L = 0, R = (ssize_t)o->file_index.size - 1;
        struct oracle_index_entry T;
        T.frame = frame;
        T.id = match_id;
        while (L <= R)
            {
            size_t m = (size_t)((L + R) / 2);
            int c = cmp_entry(o->file_index.data + m, &T);
            if (c == -1)
                L = (ssize_t)m + 1;
            else if (c == 1)
                R = (ssize_t)m - 1;
            else
                return &o->file_index.data[m];
            }
        return NULL;
        }
    else
        {
        if (o->file_index.size == 0)
            return NULL;
        size_t L = 0, R = o->file_index.size;
        do
            {
            size_t m = (L + R) / 2;
            if (frame < o->file_index.data[m].frame)
                R = m;
            else
                L = m;
            } while ((R - L) > 1);
        int64_t cur;
        for (cur = (int64_t)L; cur >= 0 && o->file_index.data[cur].frame == frame; cur--)
            if (match_id == o->file_index.data[cur].id)
                break;
        if (cur < 0 || o->file_index.data[cur].frame != frame || o->file_index.data[cur].id != match_id)
            return NULL;
        return &o->file_index.data[cur];
        }
    }

/* pgsd_read_chunk, pgsd.c:2436-2537 */
int oracle_read_chunk(pgsd_oracle* o, void* data, const struct oracle_index_entry* chunk, uint64_t N,
                      uint32_t M, uint32_t offset, bool all)
    {
    if (!o || !data || !chunk)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    if (o->open_flags != O_READONLY)
        {
        int rv = oracle_flush(o);
        if (rv != ORACLE_SUCCESS)
            return rv;
        }
    size_t sz = oracle_sizeof_type(chunk->type);
    uint64_t stride = 0;
    size_t size;
    offset = offset * M;
    if (!all)
        size = chunk->N * chunk->M * sz;
    else
        {
        size = N * M * sz;
        stride = (uint64_t)offset * sz;
        }
    if (size == 0)
        return ORACLE_ERROR_FILE_CORRUPT;
    if (chunk->location == 0)
        return ORACLE_ERROR_FILE_CORRUPT;
    if ((uint64_t)(chunk->location + size + stride) > (uint64_t)o->file_size)
        return ORACLE_ERROR_FILE_CORRUPT;
    o_pread(o->fd, data, size, chunk->location + (long long)stride);
    return ORACLE_SUCCESS;
    }

/* pgsd_find_matching_chunk_name, pgsd.c:2557-2641 */
const char* oracle_find_matching_chunk_name(pgsd_oracle* o, const char* match, const char* prev)
    {
    if (!o || !match)
        return NULL;
    if (o->file_n_names == 0)
        return NULL;
    if (o->open_flags != O_READONLY)
        if (oracle_flush(o) != ORACLE_SUCCESS)
            return NULL;
    if (o->file_names.data[o->file_names.reserved - 1] != 0)
        return NULL;
    int v1 = o->header.pgsd_version < oracle_make_version(2, 0);
    const char* s;
    if (!prev)
        s = o->file_names.data;
    else
        {
        if (prev < o->file_names.data)
            return NULL;
        if (prev >= o->file_names.data + o->file_names.reserved)
            return NULL;
        s = v1 ? prev + O_NAME_SIZE : prev + strlen(prev) + 1;
        }
    size_t ml = strlen(match);
    while (s < o->file_names.data + o->file_names.reserved)
        {
        if (s[0] != 0 && 0 == strncmp(match, s, ml))
            return s;
        s += v1 ? O_NAME_SIZE : strlen(s) + 1;
        }
    return NULL;
    }

/* ---- caller-side pack (what a CPU writer does before handing rows to pgsd_write_chunk) ---- */
static double load_as_double(const char* p, int t)
    {
    switch (t)
        {
        case 1: return (double)*(const uint8_t*)p;
        case 2: { uint16_t v; memcpy(&v, p, 2); return (double)v; }
        case 3: { uint32_t v; memcpy(&v, p, 4); return (double)v; }
        case 5: return (double)*(const int8_t*)p;
        case 6: { int16_t v; memcpy(&v, p, 2); return (double)v; }
        case 7: { int32_t v; memcpy(&v, p, 4); return (double)v; }
        case 9: { float v; memcpy(&v, p, 4); return (double)v; }
        case 10: { double v; memcpy(&v, p, 8); return v; }
        default: return 0.0;
        }
    }

int oracle_pack_rows(void* dst, int dst_type, const void* src, int src_type, uint64_t N, uint32_t M,
                     uint32_t src_stride, uint32_t col0, const uint32_t* order, int bitcast)
    {
    const size_t ssz = oracle_sizeof_type(src_type), dsz = oracle_sizeof_type(dst_type);
    if (ssz == 0 || dsz == 0 || M == 0 || col0 + M > src_stride)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    const int s_int = src_type <= 8, d_int = dst_type <= 8;
    if (bitcast && dsz > ssz)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    /* supported conversions: identical size+class (copy), bitcast, integer<->integer
       (two's complement wrap / sign- or zero-extension by source signedness),
       f64->f32 (round to nearest even), f32->f64 (exact), 32-bit-or-smaller int -> float. */
    if (!bitcast && !s_int && d_int)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    if (!bitcast && s_int && !d_int && ssz == 8)
        return ORACLE_ERROR_INVALID_ARGUMENT;
    char* d = (char*)dst;
    const char* s = (const char*)src;
    for (uint64_t i = 0; i < N; i++)
        {
        uint64_t row = order ? order[i] : i;
        for (uint32_t c = 0; c < M; c++)
            {
            const char* sp = s + (row * src_stride + col0 + c) * ssz;
            char* dp = d + (i * M + c) * dsz;
            if (bitcast || (src_type == dst_type) || (s_int && d_int && ssz == dsz))
                {
                memcpy(dp, sp, dsz); /* little endian: low bytes */
                }
            else if (s_int && d_int)
                {
                int s_signed = src_type >= 5;
                int64_t v = 0;
                if (s_signed)
                    {
                    switch (ssz)
                        {
                        case 1: v = *(const int8_t*)sp; break;
                        case 2: { int16_t t; memcpy(&t, sp, 2); v = t; break; }
                        case 4: { int32_t t; memcpy(&t, sp, 4); v = t; break; }
                        default: memcpy(&v, sp, 8); break;
                        }
                    }
                else
                    {
                    uint64_t u = 0;
                    memcpy(&u, sp, ssz);
                    v = (int64_t)u;
                    }
                memcpy(dp, &v, dsz);
                }
            else
                {
                double x = load_as_double(sp, src_type);
                if (dst_type == 9)
                    {
                    float fv = (float)x;
                    memcpy(dp, &fv, 4);
                    }
                else
                    memcpy(dp, &x, 8);
                }
            }
        }
    return ORACLE_SUCCESS;
    }
