// pgsd_pack.hpp -- kernel argument blocks of the pack kernels (pgsd_pack.hip) and the
// host-side launcher shared with the device pipeline (pgsd_device.cpp).
#ifndef PGSD_PACK_HPP
#define PGSD_PACK_HPP

#include "pgsd.h"
#include "pgsd_private.h"

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

namespace pgsd_amd
    {
enum
    {
    PACK_MAX_GROUPS = 8,       // distinct source arrays per launch
    PACK_MAX_OUT = 6,          // chunks fed from one source array
    PACK_LDS_BYTES = 49152,    // LDS budget per workgroup of the workgroup-tiled kernel (3 workgroups per CU)
    PACK_MAX_ROWBYTES = 2048,  // wider source rows take the generic kernel
    PACK_MAX_M = 1024
    };

// element conversions (wave-uniform per output)
enum
    {
    PACK_BITS = 0, // copy the low destination-size bytes (same type, narrowing, zero-extension, bitcast)
    PACK_SEXT = 1, // sign-extend a narrower signed integer
    PACK_F2F = 2,  // f64 -> f32 (round to nearest even) or f32 -> f64
    PACK_U2F = 3,  // unsigned integer (<= 32 bit) -> f32 / f64
    PACK_S2F = 4   // signed integer (<= 32 bit) -> f32 / f64
    };

// compile-time specialisations of the LDS-tiled kernels
enum
    {
    PACK_MODE_GENERIC = 0, // any element sizes / conversions, decided at run time per output
    PACK_MODE_W32 = 1,     // 4-byte source and chunk elements, bits unchanged
    PACK_MODE_F64_F32 = 2  // f64 sources, f32 chunks
    };

struct PackOut
    {
    void* dst;      // chunk buffer, 16-byte aligned
    uint32_t M;     // columns of the chunk
    uint32_t col0;  // first source column
    uint32_t dsz;   // bytes per chunk element
    uint32_t kind;  // PACK_*
    uint32_t magic; // ceil(2^32 / M) for row = e / M (0 when M == 1)
    uint32_t pad;
    };

struct PackGroup
    {
    const void* src;       // source array, 16-byte aligned
    const uint32_t* order; // gather index or nullptr
    uint32_t rowbytes;     // stride * ssz
    uint32_t ssz;          // bytes per source element
    uint32_t stride;       // elements per source row
    uint32_t n_out;
    uint32_t pad0[2];
    uint32_t lds_off;      // byte offset of this group's tile inside the workgroup's LDS (tiles kernel)
    uint32_t pad;
    PackOut out[PACK_MAX_OUT];
    };

struct PackArgs
    {
    uint64_t N;
    uint64_t n_tiles;
    uint32_t tile_rows;
    uint32_t n_groups;
    // groups [batch_start[b], batch_start[b+1]) are staged together, then emitted together
    uint32_t n_batches;
    uint32_t pad;
    uint8_t batch_start[PACK_MAX_GROUPS + 8];
    PackGroup g[PACK_MAX_GROUPS];
    };

// ---- row-per-lane kernel (pack_rows_kernel): the default for 4- and 8-byte elements
enum
    {
    ROWS_MAX_WORDS = 8, // a source row / a chunk row is at most 8 dwords (double4, N x 4 doubles)
    ROWS_MAX_GROUPS = 16, // source arrays per launch (16 x 184 B of descriptors: inside the 4 KiB of kernel arguments)
    ROWS_BITS = 0,      // dwords moved unchanged (equal element sizes; 8-byte elements are dword pairs)
    ROWS_F64_F32 = 1,   // f64 -> f32, round to nearest even
    ROWS_F32_F64 = 2    // f32 -> f64
    };

struct RowsOut
    {
    void* dst;       // chunk buffer
    uint32_t col0;   // first source ELEMENT (not dword)
    uint32_t M;      // chunk columns
    uint32_t kind;   // ROWS_*
    uint32_t nw_out; // dwords per chunk row = M * dsz / 4
    };

struct RowsGroup
    {
    const void* src;
    uint64_t copy_vecs;    // dense same-type copy: 16-byte vectors to move (0 = row mode)
    uint32_t copy_tail;    // ... and bytes behind the last whole vector
    uint32_t row_words;    // dwords per source row
    uint32_t n_out;
    uint32_t pad;
    RowsOut out[PACK_MAX_OUT];
    };

struct RowsArgs
    {
    uint64_t N;
    uint64_t n_blocks;     // gridDim.x: blocks of T*U rows (or 16-byte vectors); gridDim.y = n_groups
    uint32_t n_groups;
    uint32_t pad;
    RowsGroup g[ROWS_MAX_GROUPS];
    };
static_assert(sizeof(RowsArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// columns of destination rows that no chunk of a launch writes, set to one element value
// (pgsd_field_dst.fill_rest on the paths that do not assemble whole rows)
struct FillArgs
    {
    void* dst;
    const uint32_t* order;
    uint64_t N;
    uint64_t bits;
    uint32_t stride, dsz, colmask, pad;
    };

struct PackGenericArgs
    {
    void* dst;
    const void* src;
    const uint32_t* order;
    uint64_t N;
    uint32_t M, stride, col0, ssz, dsz, kind;
    };

enum
    {
    UNPACK_MAX_JOBS = 12,     // chunks per launch
    UNPACK_MAX_GROUPS = 6,    // destination arrays assembled row-wise per launch
    UNPACK_MAX_ROW_COLS = 16, // columns of such a destination row
    UNPACK_LDS_BYTES = 49152, // LDS budget per workgroup
    UNPACK_MAX_SUM_ROWBYTES = UNPACK_LDS_BYTES / 16 // chunk row bytes one launch can stage at the smallest tile
    };

struct UnpackJob
    {
    const void* src;       // dense chunk rows
    void* dst;             // destination array
    const uint32_t* order; // scatter index or nullptr
    uint32_t M, ssz, dsz, kind;
    uint32_t dst_stride, dst_col0, magic, rowbytes; // rowbytes = M * ssz
    uint32_t lds_off;      // where this chunk's tile sits in the workgroup's LDS
    uint32_t in_group;     // 1: written by the row assembly of its destination array
    uint32_t fill_rest;    // host side only: the job asked for the untouched columns to be filled ...
    uint32_t pad;
    uint64_t fill_bits;    // ... with this element
    };

// A destination array every column of which is restored by chunks of the same launch (position.xyz
// + type id into a Scalar4 array): its rows are assembled in registers and written with 16-byte stores.
struct UnpackGroup
    {
    void* dst;
    const uint32_t* order;
    uint32_t stride, dsz;
    uint32_t vec_shift; // log2(16-byte vectors per destination row)
    uint32_t pad;
    uint8_t col_job[UNPACK_MAX_ROW_COLS]; // per destination column: the chunk it comes from ...
    uint8_t col_off[UNPACK_MAX_ROW_COLS]; // ... and the column inside that chunk
    };

struct UnpackArgs
    {
    uint64_t N;
    uint64_t n_tiles;
    uint32_t tile_rows;
    uint32_t n_jobs;
    uint32_t n_groups;
    uint32_t pad;
    UnpackJob j[UNPACK_MAX_JOBS];
    UnpackGroup g[UNPACK_MAX_GROUPS];
    };

// ---- row-per-lane unpack (unpack_rows_kernel): Scalar4 / double4 destination arrays fed by one or two chunks
struct UnrowsGroup
    {
    void* dst;          // destination array: rows of 4 elements (float4-like; double4 when converting f32 -> f64)
    const void* a;      // first chunk (dense rows of a_nw dwords), never null
    const void* b;      // second chunk or nullptr
    uint32_t a_nw, a_col0, b_nw, b_col0; // dwords per chunk row / first destination ELEMENT
    uint64_t copy_vecs; // dense same-type array riding along: 16-byte vectors to copy from `a` to `dst` (0 = row mode)
    uint32_t copy_tail;
    uint32_t fill_on;   // 1: columns no chunk feeds take the fill element, the row is stored whole
    uint32_t fill_lo, fill_hi; // bits of one destination element (hi: the upper half of a double)
    };

struct UnrowsArgs
    {
    uint64_t N;
    uint64_t n_blocks;
    uint32_t n_groups;
    uint32_t pad;
    UnrowsGroup g[ROWS_MAX_GROUPS];
    };

// Enqueue the unpack of `n_jobs` chunks of N rows each on `stream`. Returns a pgsd_error.
int launch_unpack(uint32_t n_jobs, const pgsd_unpack_job* jobs, uint64_t N, hipStream_t stream, std::string* err);

// Enqueue the pack of `n_jobs` fields of N rows each on `stream`. Returns a pgsd_error.
// ev_start / ev_stop (optional) receive the begin time of the first and the end time of the last kernel.
int launch_pack(uint32_t n_jobs, const pgsd_pack_job* jobs, uint64_t N, hipStream_t stream, std::string* err,
                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- comparison of packed chunks with reference rows (pgsd_compare_staged_chunks)
enum
    {
    CMP_MAX_JOBS = 64
    };

enum
    {
    CMP_BYTES = 0, // integers: equal bytes
    CMP_F32 = 1,   // float chunks are compared by VALUE, as numpy.array_equal compares them (hoomd.py:679-682):
    CMP_F64 = 2    // a NaN equals nothing (itself included), +0.0 equals -0.0
    };

struct CompareJob
    {
    const void* a;   // packed chunk (staging: HBM, or the pinned arena of the direct path)
    const void* b;   // reference in device memory
    uint64_t bytes;  // of the chunk
    uint64_t period; // 0: the reference holds `bytes` bytes; else it holds `period` bytes (a multiple of 16 and of the
                     // element size) and REPEATS: byte j of the chunk is compared with byte j % period of the reference
                     // (a few thousand rows of a default value stand for any number of rows)
    uint32_t mode;   // CMP_*
    uint32_t pad;
    };

struct CompareArgs
    {
    uint32_t* dflags; // device memory, one word per job: == gen once a difference was seen (early exit of the other workgroups)
    uint32_t* hflags; // pinned host memory (device-mapped), one word per job: the answer
    uint32_t gen;     // this launch's mark: the flag words are never cleared
    uint32_t n_jobs;
    uint64_t limit;   // compare at most this many bytes of every job (the probe launch); 0: all
    CompareJob j[CMP_MAX_JOBS];
    };

// Enqueue the comparison of `n_jobs` (<= CMP_MAX_JOBS) byte ranges on `stream`; hflags[i] == gen afterwards: they differ.
int launch_compare(uint32_t n_jobs, const CompareJob* jobs, uint32_t gen, uint32_t* dflags, uint32_t* hflags,
                   hipStream_t stream, std::string* err);

// The runtime loads a code object when one of its kernels is first used (about a millisecond each: the pack, the unpack and
// the compare / select kernels are three objects).  A caller who asked for everything up front (prealloc_mib) gets that
// over with in pgsd_device_configure: one attribute query per object.
void warm_pack_kernels();
void warm_unpack_kernels();
void warm_select_kernels();

// algorithmic traffic of one job: bytes that must be read (needed columns only, plus the
// gather index) and chunk bytes written
uint64_t pack_algorithmic_bytes_in(const pgsd_pack_job& j, uint64_t N);
uint64_t pack_bytes_out(const pgsd_pack_job& j, uint64_t N);
    } // namespace pgsd_amd

#endif
