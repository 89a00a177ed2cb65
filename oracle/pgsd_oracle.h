/* pgsd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's chunk-write hot path
 * (/root/reference/pgsd/pgsd/pgsd.c), used only by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker for the HIP/C++ product in pgsd-sph_amd/.
 *
 * The reference is an MPI program (one process per rank, collectives on MPI_COMM_WORLD).
 * The oracle replays the SAME algorithm for all P ranks inside one process: every
 * per-rank quantity of the reference (local row count, element offset, data pointer,
 * per-rank small-chunk write buffer) is an array indexed by rank, every collective
 * becomes a loop over that array, and every MPI_File_write_at becomes a pwrite at the
 * identical byte offset.  Root-only state of the reference (name list, index buffers)
 * exists once.  Parity pinned: byte-identical to files written by the compiled reference
 * itself (oracle/_ref, goldens under tests/golden/) -- see tests/test_oracle_golden.py.
 */
#ifndef PGSD_ORACLE_H
#define PGSD_ORACLE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes: pgsd.h:85-120 */
enum
    {
    ORACLE_SUCCESS = 0,
    ORACLE_ERROR_IO = -1,
    ORACLE_ERROR_INVALID_ARGUMENT = -2,
    ORACLE_ERROR_NOT_A_PGSD_FILE = -3,
    ORACLE_ERROR_INVALID_PGSD_FILE_VERSION = -4,
    ORACLE_ERROR_FILE_CORRUPT = -5,
    ORACLE_ERROR_MEMORY_ALLOCATION_FAILED = -6,
    ORACLE_ERROR_NAMELIST_FULL = -7,
    ORACLE_ERROR_FILE_MUST_BE_WRITABLE = -8,
    ORACLE_ERROR_FILE_MUST_BE_READABLE = -9,
    /* the reference would dead-lock here (ranks disagree on entering a collective) */
    ORACLE_ERROR_REFERENCE_WOULD_HANG = -100
    };

/* pgsd.h:143-174 */
struct oracle_header
    {
    uint64_t magic;
    uint64_t index_location;
    uint64_t index_allocated_entries;
    uint64_t namelist_location;
    uint64_t namelist_allocated_entries;
    uint32_t schema_version;
    uint32_t pgsd_version;
    char application[64];
    char schema[64];
    char reserved[80];
    };

/* pgsd.h:182-204 */
struct oracle_index_entry
    {
    uint64_t frame;
    uint64_t N;
    int64_t location;
    uint32_t M;
    uint16_t id;
    uint8_t type;
    uint8_t flags;
    };

typedef struct pgsd_oracle pgsd_oracle;

uint32_t oracle_make_version(unsigned int major, unsigned int minor);
size_t oracle_sizeof_type(int type);

/* flags: 1 READWRITE, 2 READONLY, 3 APPEND (pgsd.h:72-82) */
pgsd_oracle* oracle_create_and_open(const char* fname, int nprocs, const char* application,
                                    const char* schema, uint32_t schema_version, int flags,
                                    int exclusive_create, int* rc);
pgsd_oracle* oracle_open(const char* fname, int nprocs, int flags, int* rc);
int oracle_close(pgsd_oracle* o);
int oracle_end_frame(pgsd_oracle* o);
int oracle_flush(pgsd_oracle* o);

/* One collective pgsd_write_chunk call.  N, offset, global_size, data are arrays of
   length nprocs holding what each rank passes (pgsd.h:551-564). */
int oracle_write_chunk(pgsd_oracle* o, const char* name, int type, const uint64_t* N, uint32_t M,
                       uint64_t N_global, uint32_t M_global, const uint64_t* offset,
                       const uint64_t* global_size, bool all, uint8_t flags,
                       const void* const* data);

/* Read side (single reader = rank 0 view; pgsd.c:2295-2537). */
const struct oracle_index_entry* oracle_find_chunk(pgsd_oracle* o, uint64_t frame, const char* name);
int oracle_read_chunk(pgsd_oracle* o, void* data, const struct oracle_index_entry* chunk, uint64_t N,
                      uint32_t M, uint32_t offset, bool all);
const char* oracle_find_matching_chunk_name(pgsd_oracle* o, const char* match, const char* prev);

uint64_t oracle_get_nframes(pgsd_oracle* o);
uint64_t oracle_get_nnames(pgsd_oracle* o);
long long oracle_get_file_size(pgsd_oracle* o);
const struct oracle_header* oracle_get_header(pgsd_oracle* o);
int oracle_set_maximum_write_buffer_size(pgsd_oracle* o, uint64_t size);
int oracle_set_index_entries_to_buffer(pgsd_oracle* o, uint64_t number);
uint64_t oracle_get_maximum_write_buffer_size(pgsd_oracle* o);
uint64_t oracle_get_index_entries_to_buffer(pgsd_oracle* o);

/* CPU restatement of the caller-side pack a CPU writer performs before pgsd_write_chunk:
   out[i, c] = convert(src[order ? order[i] : i][col0 + c]), i < N, c < M.
   src_type/dst_type are pgsd type ids (1..10); src rows are src_stride elements apart.
   bitcast != 0 copies the low dst-size bytes of the source element unchanged
   (HOOMD keeps typeid in position.w via __int_as_scalar). Returns 0 or INVALID_ARGUMENT. */
int oracle_pack_rows(void* dst, int dst_type, const void* src, int src_type, uint64_t N, uint32_t M,
                     uint32_t src_stride, uint32_t col0, const uint32_t* order, int bitcast);

#ifdef __cplusplus
}
#endif
#endif
