// pgsd_unpack.hip -- gfx950 kernels of the READ path (restart files) and their launcher: the inverse of the pack.
// All chunks of a frame in one launch; whole destination rows assembled in registers where the launch restores every
// column of an array (position.xyz + the type id into a Scalar4 array), 16-byte stores; an LDS-tiled form for narrow
// element types, conversions and scatters.  No reference counterpart; outputs pinned by the oracle and by
// round trips (tests/test_gpu_read.py, test_gpu_config5.py).  Shared device helpers: pgsd_kernels.hpp.
#include "pgsd_kernels.hpp"

namespace pgsd_amd
    {
// ------------------------------------------------------------------ unpack (read path)
// Inverse of the pack.  The dense tiles of ALL chunks of a launch (contiguous bytes) are
// streamed into LDS with 16-byte non-temporal loads, one barrier, then
//   * destination arrays whose rows are completely restored by this launch (position.xyz and
//     the type id into a Scalar4 array, velocity.xyz and mass, double4 builds ...) are
//     assembled row-wise in registers and written with 16-byte non-temporal stores: whole
//     lines, no partial-line writes;
//   * every other chunk is scattered element-wise: lane e converts element e of the tile and
//     stores it to its (row, column); columns no chunk restores are left untouched.
// W32 = every chunk and destination element is 4 bytes wide and moved unchanged.
template<bool W32> __device__ __forceinline__ uint64_t unpack_elem(const char* p, uint32_t ssz, uint32_t dsz, uint32_t kind)
    {
    if constexpr (W32)
        return *(const uint32_t*)p;
    else
        {
        switch (ssz)
            {
            case 1: return dsz == 8 ? convert_elem<1, 8>(lds_load<1>(p), kind) : convert_elem<1, 4>(lds_load<1>(p), kind);
            case 2: return dsz == 8 ? convert_elem<2, 8>(lds_load<2>(p), kind) : convert_elem<2, 4>(lds_load<2>(p), kind);
            case 4: return dsz == 8 ? convert_elem<4, 8>(lds_load<4>(p), kind) : convert_elem<4, 4>(lds_load<4>(p), kind);
            default: return dsz == 4 ? convert_elem<8, 4>(lds_load<8>(p), kind) : convert_elem<8, 8>(lds_load<8>(p), kind);
            }
        }
    }

template<bool W32>
__device__ __forceinline__ void scatter_tile(const UnpackJob& j, const char* lds, uint32_t rows, uint64_t row0)
    {
    const uint32_t M = j.M, ssz = W32 ? 4u : j.ssz, dsz = W32 ? 4u : j.dsz;
    const uint32_t nelem = rows * M;
    for (uint32_t e = threadIdx.x; e < nelem; e += PACK_THREADS)
        {
        uint32_t row = (M == 1) ? e : __umulhi(e, j.magic);
        uint32_t col = e - row * M;
        uint64_t val = unpack_elem<W32>(lds + (size_t)e * ssz, ssz, dsz, j.kind);
        uint64_t drow = j.order ? (uint64_t)j.order[row0 + row] : row0 + row;
        char* p = (char*)j.dst + (drow * j.dst_stride + j.dst_col0 + col) * dsz;
        if (dsz == 8)
            *(uint64_t*)p = val;
        else if (dsz == 4)
            *(uint32_t*)p = (uint32_t)val;
        else if (dsz == 2)
            *(uint16_t*)p = (uint16_t)val;
        else
            *(uint8_t*)p = (uint8_t)val;
        }
    }

// Where one column of an assembled destination row comes from (kept in LDS: lanes that build
// different vectors of a wide row look up different columns).
struct UnpackCol
    {
    uint32_t base; // LDS byte offset of column 0 .. of row 0 of the chunk tile
    uint32_t step; // bytes per chunk row
    uint32_t ssz, kind;
    };
#define UNPACK_TABLE_BYTES (UNPACK_MAX_GROUPS * UNPACK_MAX_ROW_COLS * 16)

// one 16-byte vector of a destination row per lane per step; rows are 16, 32 or 64 bytes
template<bool W32>
__device__ __forceinline__ void assemble_rows(const UnpackGroup& g, const UnpackCol* tab, const char* lds, uint32_t rows,
                                              uint64_t row0)
    {
    const uint32_t shift = g.vec_shift, mask = (1u << shift) - 1u;
    const uint32_t dsz = W32 ? 4u : g.dsz;
    const uint32_t ept = 16u / dsz; // 4 or 2 elements per vector
    const uint32_t nvec = rows << shift;
    const uint32_t rowbytes = g.stride * dsz;
    const uint32_t* order = g.order;
    char* dst = (char*)g.dst;
    if (shift == 0)
        {
        // 16-byte rows (Scalar4 of floats / ints): the four column descriptors are loop invariants
        UnpackCol d[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            d[k] = tab[k < ept ? k : 0];
        for (uint32_t row = threadIdx.x; row < rows; row += PACK_THREADS)
            {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
                {
                if (k >= ept)
                    break;
                uint64_t val = unpack_elem<W32>(lds + d[k].base + row * d[k].step, d[k].ssz, dsz, d[k].kind);
                if (dsz == 8)
                    {
                    w[2 * k] = (uint32_t)val;
                    w[2 * k + 1] = (uint32_t)(val >> 32);
                    }
                else
                    w[k] = (uint32_t)val;
                }
            const uint64_t drow = order ? (uint64_t)order[row0 + row] : row0 + row;
            u32x4 out = {w[0], w[1], w[2], w[3]};
            __builtin_nontemporal_store(out, (u32x4*)(dst + drow * 16));
            }
        return;
        }
    for (uint32_t v = threadIdx.x; v < nvec; v += PACK_THREADS)
        {
        const uint32_t row = v >> shift, q = v & mask;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            {
            if (k >= ept)
                break;
            const UnpackCol d = tab[q * ept + k];
            uint64_t val = unpack_elem<W32>(lds + d.base + row * d.step, d.ssz, dsz, d.kind);
            if (dsz == 8)
                {
                w[2 * k] = (uint32_t)val;
                w[2 * k + 1] = (uint32_t)(val >> 32);
                }
            else
                w[k] = (uint32_t)val;
            }
        const uint64_t drow = order ? (uint64_t)order[row0 + row] : row0 + row;
        u32x4 out = {w[0], w[1], w[2], w[3]};
        __builtin_nontemporal_store(out, (u32x4*)(dst + drow * rowbytes + (size_t)q * 16));
        }
    }

template<bool W32> __global__ __launch_bounds__(PACK_THREADS) void unpack_tiles_kernel(const UnpackArgs args)
    {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t TILE = args.tile_rows;
    // column table of the assembled arrays, once per workgroup (chunk tiles start behind it)
    UnpackCol* tab = (UnpackCol*)lds;
    if (tid < args.n_groups * UNPACK_MAX_ROW_COLS)
        {
        const UnpackGroup& g = args.g[tid / UNPACK_MAX_ROW_COLS];
        const uint32_t col = tid % UNPACK_MAX_ROW_COLS;
        UnpackCol d = {0, 0, 4, 0};
        if (col < g.stride)
            {
            const UnpackJob& j = args.j[g.col_job[col]];
            d.base = j.lds_off + g.col_off[col] * j.ssz;
            d.step = j.rowbytes;
            d.ssz = j.ssz;
            d.kind = j.kind;
            }
        tab[tid] = d;
        }
    // (the first tile's barrier publishes the table)
    for (uint64_t tile = blockIdx.x; tile < args.n_tiles; tile += gridDim.x)
        {
        const uint64_t row0 = tile * TILE;
        const uint32_t rows = (uint32_t)((args.N - row0 < (uint64_t)TILE) ? args.N - row0 : TILE);
        // every chunk tile in flight before the first byte is consumed
        for (uint32_t ji = 0; ji < args.n_jobs; ji++)
            {
            const UnpackJob& j = args.j[ji];
            const char* gsrc = (const char*)j.src + row0 * j.rowbytes;
            char* l = lds + j.lds_off;
            const uint32_t nbytes = rows * j.rowbytes;
            const uint32_t nvec = nbytes >> 4;
            uint32_t v = tid;
            for (; v + 3 * PACK_THREADS < nvec; v += 4 * PACK_THREADS)
                {
                u32x4 a = __builtin_nontemporal_load((const u32x4*)gsrc + v);
                u32x4 b = __builtin_nontemporal_load((const u32x4*)gsrc + v + PACK_THREADS);
                u32x4 c = __builtin_nontemporal_load((const u32x4*)gsrc + v + 2 * PACK_THREADS);
                u32x4 d = __builtin_nontemporal_load((const u32x4*)gsrc + v + 3 * PACK_THREADS);
                ((u32x4*)l)[v] = a;
                ((u32x4*)l)[v + PACK_THREADS] = b;
                ((u32x4*)l)[v + 2 * PACK_THREADS] = c;
                ((u32x4*)l)[v + 3 * PACK_THREADS] = d;
                }
            for (; v < nvec; v += PACK_THREADS)
                ((u32x4*)l)[v] = __builtin_nontemporal_load((const u32x4*)gsrc + v);
            for (uint32_t b = (nvec << 4) + tid; b < nbytes; b += PACK_THREADS)
                l[b] = gsrc[b];
            }
        __syncthreads();
        for (uint32_t gi = 0; gi < args.n_groups; gi++)
            assemble_rows<W32>(args.g[gi], tab + gi * UNPACK_MAX_ROW_COLS, lds, rows, row0);
        for (uint32_t ji = 0; ji < args.n_jobs; ji++)
            if (!args.j[ji].in_group)
                scatter_tile<W32>(args.j[ji], lds + args.j[ji].lds_off, rows, row0);
        __syncthreads();
        }
    }

// ------------------------------------------------------------------ row-per-lane unpack (Scalar4 destinations)
// Inverse of pack_rows_kernel for the restart path's common shape: a float4-like destination array fed by
// one or two dense chunks (position.xyz + the type id into position.w, velocity.xyz + mass).  Lane i loads
// row i of each chunk (12 bytes + 4 bytes: contiguous pieces, consecutive lanes consecutive rows), and
// stores ONE 16-byte row (two for a double4 destination restored from f32 chunks).  No LDS, no barrier;
// blockIdx.y = destination array; dense same-type arrays ride along as 16-byte copies.  Measured in the lab
// (the round-2 lab, profiles/r02_lab_unpack.jsonl): 100 us for position + id + velocity + mass of 10 M
// particles where the LDS-tiled unpack needs 113 us.  A first, fully generic version of this kernel (run-time
// chunk lists and widths) was no faster than the tiled one; the static hot path is what pays.
template<bool F64> __device__ __forceinline__ void unrows_store(uint32_t* drow, const uint32_t* w, uint32_t nw, uint32_t col0)
    {
    // nw source dwords (f32 / 32-bit integers) -> destination elements col0 .. col0+nw
    if constexpr (!F64)
        {
        uint32_t c[ROWS_MAX_WORDS] = {w[0], w[1], w[2], w[3], 0, 0, 0, 0};
        row_store<4>(drow + col0, c, nw);
        }
    else
        {
        uint32_t c[ROWS_MAX_WORDS];
#pragma unroll
        for (uint32_t e = 0; e < 4; e++)
            {
            const uint64_t bits = (uint64_t)__double_as_longlong((double)__uint_as_float(w[e]));
            c[2 * e] = (uint32_t)bits;
            c[2 * e + 1] = (uint32_t)(bits >> 32);
            }
        row_store<8>(drow + 2 * col0, c, 2 * nw);
        }
    }

__device__ __forceinline__ void unrows_load(const uint32_t* p, uint32_t nw, uint32_t (&w)[4])
    {
    RowRegs r;
    r.lo = u32x4 {0, 0, 0, 0};
    switch (nw)
        {
        case 1: row_load<1>(p, r); break;
        case 2: row_load<2>(p, r); break;
        case 3: row_load<3>(p, r); break;
        default: row_load<4>(p, r); break;
        }
    w[0] = r.lo.x, w[1] = r.lo.y, w[2] = r.lo.z, w[3] = r.lo.w;
    }

// One WHOLE destination row: column e takes v[e] (f32 / 32-bit bits from a chunk) unless bit e of fillmask is
// set, then the fill element.  Compile-time loops only (a run-time index into c[] would go to scratch).
template<bool F64>
__device__ __forceinline__ void unrows_store_whole(uint32_t* drow, const uint32_t (&v)[4], uint32_t fillmask, uint32_t fill_lo,
                                                   uint32_t fill_hi)
    {
    if constexpr (!F64)
        {
        uint32_t c[ROWS_MAX_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t e = 0; e < 4; e++)
            c[e] = ((fillmask >> e) & 1u) ? fill_lo : v[e];
        row_store<4>(drow, c, 4);
        }
    else
        {
        uint32_t c[ROWS_MAX_WORDS];
#pragma unroll
        for (uint32_t e = 0; e < 4; e++)
            {
            const uint64_t bits = (uint64_t)__double_as_longlong((double)__uint_as_float(v[e]));
            const bool f = (fillmask >> e) & 1u;
            c[2 * e] = f ? fill_lo : (uint32_t)bits;
            c[2 * e + 1] = f ? fill_hi : (uint32_t)(bits >> 32);
            }
        row_store<8>(drow, c, 8);
        }
    }

// v[col0 + s] = w[s] for s < nw, without run-time indexing
__device__ __forceinline__ void unrows_place(uint32_t (&v)[4], const uint32_t (&w)[4], uint32_t nw, uint32_t col0)
    {
#pragma unroll
    for (uint32_t e = 0; e < 4; e++)
#pragma unroll
        for (uint32_t s = 0; s < 4; s++)
            if (s < nw && col0 + s == e)
                v[e] = w[s];
    }

template<int T, int U, bool F64> __global__ __launch_bounds__(T) void unpack_rows_kernel(const UnrowsArgs args)
    {
    const UnrowsGroup& g = args.g[blockIdx.y];
    if (g.copy_vecs != 0 || g.copy_tail != 0)
        {
        // dense same-type array: the chunk IS the array
        const u32x4* src = (const u32x4*)g.a;
        u32x4* dst = (u32x4*)g.dst;
        const uint64_t nvec = g.copy_vecs;
        const uint64_t per = (nvec + gridDim.x - 1) / gridDim.x;
        const uint64_t first = (uint64_t)blockIdx.x * per;
        const uint64_t last = first + per < nvec ? first + per : nvec;
        for (uint64_t v = first + threadIdx.x; v < last; v += T)
            __builtin_nontemporal_store(__builtin_nontemporal_load(src + v), dst + v);
        if (blockIdx.x == 0 && threadIdx.x < g.copy_tail)
            ((char*)dst)[nvec * 16 + threadIdx.x] = ((const char*)src)[nvec * 16 + threadIdx.x];
        return;
        }
    const uint64_t N = args.N;
    const uint64_t base = (uint64_t)blockIdx.x * (uint64_t)(T * U) + threadIdx.x;
    constexpr uint32_t DW = F64 ? 8 : 4;
    uint32_t* dst = (uint32_t*)g.dst;
    const uint32_t* a = (const uint32_t*)g.a;
    const uint32_t* b = (const uint32_t*)g.b;
    if (g.a_nw == 3 && g.a_col0 == 0 && b == nullptr && g.fill_on)
        {
        // xyz from a chunk, w = a constant (velocity without a mass chunk -> {vx, vy, vz, 1.0f}): whole rows out
        u32x3 xyz[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i < N)
                xyz[k] = __builtin_nontemporal_load((const u32x3_a4*)(a + i * 3));
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i < N)
                {
                const uint32_t c[4] = {xyz[k].x, xyz[k].y, xyz[k].z, 0};
                unrows_store_whole<F64>(dst + i * DW, c, 8u, g.fill_lo, g.fill_hi);
                }
            }
        return;
        }
    if (g.fill_on)
        {
        // any other one- or two-chunk shape with a fill: compose the whole row, one store
        uint32_t mask = 15u;
#pragma unroll
        for (uint32_t s = 0; s < 4; s++)
            {
            if (s < g.a_nw)
                mask &= ~(1u << (g.a_col0 + s));
            if (b != nullptr && s < g.b_nw)
                mask &= ~(1u << (g.b_col0 + s));
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i >= N)
                continue;
            uint32_t wa[4], wb[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
            unrows_load(a + i * g.a_nw, g.a_nw, wa);
            unrows_place(v, wa, g.a_nw, g.a_col0);
            if (b != nullptr)
                {
                unrows_load(b + i * g.b_nw, g.b_nw, wb);
                unrows_place(v, wb, g.b_nw, g.b_col0);
                }
            unrows_store_whole<F64>(dst + i * DW, v, mask, g.fill_lo, g.fill_hi);
            }
        return;
        }
    if (g.a_nw == 3 && g.a_col0 == 0 && b != nullptr && g.b_nw == 1 && g.b_col0 == 3)
        {
        // the hot shape: xyz from one chunk, w from another, whole rows out
        u32x3 xyz[U];
        uint32_t w[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i < N)
                {
                xyz[k] = __builtin_nontemporal_load((const u32x3_a4*)(a + i * 3));
                w[k] = __builtin_nontemporal_load(b + i);
                }
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i < N)
                {
                const uint32_t c[4] = {xyz[k].x, xyz[k].y, xyz[k].z, w[k]};
                unrows_store<F64>(dst + i * DW, c, 4, 0);
                }
            }
        return;
        }
    // any other one- or two-chunk shape: each chunk's elements go to their columns
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * T;
        if (i >= N)
            continue;
        uint32_t wa[4], wb[4] = {0, 0, 0, 0};
        unrows_load(a + i * g.a_nw, g.a_nw, wa);
        if (b != nullptr)
            unrows_load(b + i * g.b_nw, g.b_nw, wb);
        unrows_store<F64>(dst + i * DW, wa, g.a_nw, g.a_col0);
        if (b != nullptr)
            unrows_store<F64>(dst + i * DW, wb, g.b_nw, g.b_col0);
        }
    }

// ------------------------------------------------------------------ fill of untouched columns (generic paths)
// pgsd_field_dst.fill_rest where the launch does not assemble whole rows (scatter index, narrow or wide
// elements, more than two chunks per array): the columns in colmask of every destination row receive the fill
// element before the chunks' kernels run.  Element per lane: a fallback, not a hot path.
__global__ __launch_bounds__(256) void fill_cols_kernel(const FillArgs a)
    {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = t / a.stride;
    const uint32_t col = (uint32_t)(t % a.stride);
    if (row >= a.N || col >= 32 || !((a.colmask >> col) & 1u))
        return;
    const uint64_t r = a.order ? (uint64_t)a.order[row] : row;
    char* p = (char*)a.dst + (r * a.stride + col) * a.dsz;
    switch (a.dsz)
        {
        case 1: *(uint8_t*)p = (uint8_t)a.bits; break;
        case 2: *(uint16_t*)p = (uint16_t)a.bits; break;
        case 4: *(uint32_t*)p = (uint32_t)a.bits; break;
        default: *(uint64_t*)p = a.bits; break;
        }
    }

// ---- row-per-lane unpack: which destination arrays it takes and how it is launched
struct UnrowsPlan
    {
    UnrowsArgs args;
    bool f64 = false;
    };

static void launch_unrows(const UnrowsPlan& p, uint64_t N, hipStream_t stream)
    {
    // measured (profiles/r02_lab_unpack.jsonl, r02_unpack_rows_final.jsonl): thin workgroups; 64 x 2 rows
    // 102.4-103.0 us, 128 x 1 104.6-105.0 us, 256 x 2 105.9-106.0 us (10 M particles, stream events)
    int T = 64, U = 2;
    const PackTuning tune = tuning();
    if (tune.unrows_t)
        T = tune.unrows_t, U = tune.unrows_u;
    UnrowsArgs a = p.args;
    a.n_blocks = (N + (uint64_t)T * U - 1) / ((uint64_t)T * U);
    const dim3 grid((unsigned)a.n_blocks, a.n_groups);
#define UNROWS_LAUNCH(TT, UU)                                                                                  \
    if (T == TT && U == UU)                                                                                    \
        {                                                                                                      \
        if (p.f64)                                                                                             \
            hipLaunchKernelGGL((unpack_rows_kernel<TT, UU, true>), grid, dim3(TT), 0, stream, a);              \
        else                                                                                                   \
            hipLaunchKernelGGL((unpack_rows_kernel<TT, UU, false>), grid, dim3(TT), 0, stream, a);             \
        return;                                                                                                \
        }
    UNROWS_LAUNCH(128, 1)
    UNROWS_LAUNCH(256, 1)
    UNROWS_LAUNCH(256, 2)
    UNROWS_LAUNCH(128, 2)
    a.n_blocks = (N + 127) / 128;
    const dim3 grid1((unsigned)a.n_blocks, a.n_groups);
    if (p.f64)
        hipLaunchKernelGGL((unpack_rows_kernel<64, 2, true>), grid1, dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL((unpack_rows_kernel<64, 2, false>), grid1, dim3(64), 0, stream, a);
#undef UNROWS_LAUNCH
    }

// One batch of <= UNPACK_MAX_JOBS validated chunks -> one launch.
static void launch_unpack_batch(const std::vector<UnpackJob>& jobs, uint64_t N, hipStream_t stream)
    {
    UnpackArgs args;
    memset(&args, 0, sizeof(args));
    args.N = N;
    args.n_jobs = (uint32_t)jobs.size();
    uint32_t sum_rowbytes = 0;
    bool w32 = true;
    for (uint32_t i = 0; i < args.n_jobs; i++)
        {
        args.j[i] = jobs[i];
        sum_rowbytes += jobs[i].rowbytes;
        w32 = w32 && jobs[i].ssz == 4 && jobs[i].dsz == 4 && jobs[i].kind == PACK_BITS;
        }
    // destination arrays whose rows this batch restores completely: dst row of 16, 32 or 64 bytes,
    // 4- or 8-byte elements, every column written by exactly one chunk
    for (uint32_t i = 0; i < args.n_jobs && args.n_groups < UNPACK_MAX_GROUPS; i++)
        {
        UnpackJob& a = args.j[i];
        if (a.in_group)
            continue;
        const uint32_t rowbytes = a.dst_stride * a.dsz;
        if ((a.dsz != 4 && a.dsz != 8) || (rowbytes != 16 && rowbytes != 32 && rowbytes != 64)
            || (((uintptr_t)a.dst) & 15) != 0)
            continue;
        UnpackGroup g;
        memset(&g, 0, sizeof(g));
        uint8_t covered[UNPACK_MAX_ROW_COLS] = {0};
        uint32_t n_cov = 0;
        bool clean = true;
        for (uint32_t k = i; k < args.n_jobs; k++)
            {
            const UnpackJob& b = args.j[k];
            if (b.dst != a.dst || b.in_group)
                continue;
            if (b.order != a.order || b.dst_stride != a.dst_stride || b.dsz != a.dsz)
                {
                clean = false; // same array seen through different shapes: leave it to the element path
                break;
                }
            for (uint32_t c = 0; c < b.M; c++)
                {
                if (covered[b.dst_col0 + c])
                    clean = false;
                covered[b.dst_col0 + c] = 1;
                g.col_job[b.dst_col0 + c] = (uint8_t)k;
                g.col_off[b.dst_col0 + c] = (uint8_t)c;
                n_cov++;
                }
            }
        if (!clean || n_cov != a.dst_stride)
            continue;
        g.dst = a.dst;
        g.order = a.order;
        g.stride = a.dst_stride;
        g.dsz = a.dsz;
        g.vec_shift = rowbytes == 16 ? 0u : (rowbytes == 32 ? 1u : 2u);
        for (uint32_t k = i; k < args.n_jobs; k++)
            if (args.j[k].dst == a.dst)
                args.j[k].in_group = 1;
        args.g[args.n_groups++] = g;
        }
    // measured (profiles/r01_unpack_sweep.log): the unpack wants more resident workgroups than the
    // pack -- 512-row tiles x 8 workgroups per CU beat 1024 x 4 by 8 % at 10 M rows; launches too small
    // to fill the chip twice keep the larger tile
    uint32_t tile = 16, tile_cap = N > (1ull << 21) ? 512 : 1024;
    uint64_t per_cu = 8;
    const PackTuning tune = tuning(); // tuning sweeps (tools/unpack_bench.py)
    if (tune.unpack_tile_cap)
        tile_cap = tune.unpack_tile_cap;
    per_cu = tune.unpack_per_cu;
    while (tile * 2 <= tile_cap && (uint64_t)tile * 2 * sum_rowbytes <= UNPACK_LDS_BYTES)
        tile <<= 1;
    args.tile_rows = tile;
    args.n_tiles = (N + tile - 1) / tile;
    size_t lds_bytes = UNPACK_TABLE_BYTES; // the column table of the kernel sits in front
    for (uint32_t i = 0; i < args.n_jobs; i++)
        {
        args.j[i].lds_off = (uint32_t)lds_bytes;
        lds_bytes += (size_t)tile * args.j[i].rowbytes; // tile is a multiple of 16: stays 16-byte aligned
        }
    uint64_t resident = lds_bytes ? (160u * 1024u) / lds_bytes : 8;
    uint64_t blocks = args.n_tiles;
    uint64_t cap = (uint64_t)num_cus() * std::max<uint64_t>(1, std::min<uint64_t>(per_cu, resident));
    if (blocks > cap)
        blocks = cap;
    if (w32)
        hipLaunchKernelGGL(unpack_tiles_kernel<true>, dim3((unsigned)blocks), dim3(PACK_THREADS), lds_bytes, stream, args);
    else
        hipLaunchKernelGGL(unpack_tiles_kernel<false>, dim3((unsigned)blocks), dim3(PACK_THREADS), lds_bytes, stream, args);
    }

void warm_unpack_kernels()
    {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, (const void*)fill_cols_kernel);
    (void)hipGetLastError();
    }

int launch_unpack(uint32_t n_jobs, const pgsd_unpack_job* jobs, uint64_t N, hipStream_t stream, std::string* err)
    {
    // whatever an earlier call of this thread left in the runtime's last-error slot (a failed hipMalloc, the caller's own
    // calls) is not this launch's: the slot is read again right behind the launches
    (void)hipGetLastError();

    if (n_jobs == 0 || N == 0)
        return PGSD_SUCCESS;
    std::vector<UnpackJob> all;
    all.reserve(n_jobs);
    for (uint32_t i = 0; i < n_jobs; i++)
        {
        const pgsd_unpack_job& q = jobs[i];
        const uint32_t ssz = (uint32_t)sizeof_type(q.src_type), dsz = (uint32_t)sizeof_type(q.dst.dst_type);
        const bool s_int = q.src_type <= PGSD_TYPE_INT64, d_int = q.dst.dst_type <= PGSD_TYPE_INT64;
        bool ok = q.src && q.dst.dst && ssz && dsz && q.M && q.M <= PACK_MAX_M
                  && q.dst.dst_col0 + q.M <= q.dst.dst_stride && (((uintptr_t)q.src) & 15) == 0
                  && (((uintptr_t)q.dst.dst) & (dsz - 1)) == 0 && (uint64_t)q.M * ssz <= PACK_MAX_ROWBYTES;
        if (q.dst.bitcast)
            ok = ok && ssz == dsz;
        else
            ok = ok && !(!s_int && d_int) && !(s_int && !d_int && ssz == 8);
        ok = ok && !(q.dst.fill_rest && q.dst.dst_stride > 32); // the fill addresses columns with a 32-bit mask
        if (!ok)
            {
            if (err)
                *err = "invalid unpack job (types, columns, alignment or pointers)";
            return PGSD_ERROR_INVALID_ARGUMENT;
            }
        UnpackJob j;
        memset(&j, 0, sizeof(j));
        j.src = q.src;
        j.dst = q.dst.dst;
        j.order = q.dst.order;
        j.M = q.M;
        j.ssz = ssz;
        j.dsz = dsz;
        j.kind = conv_kind(q.src_type, q.dst.dst_type, q.dst.bitcast);
        j.dst_stride = q.dst.dst_stride;
        j.dst_col0 = q.dst.dst_col0;
        j.magic = q.M == 1 ? 0u : (uint32_t)(((1ull << 32) + q.M - 1) / q.M);
        j.rowbytes = q.M * ssz;
        j.fill_rest = q.dst.fill_rest ? 1u : 0u;
        j.fill_bits = q.dst.fill_bits;
        all.push_back(j);
        }
    // chunks of one destination array next to each other (their relative order is kept)
    std::stable_sort(all.begin(), all.end(), [](const UnpackJob& a, const UnpackJob& b) { return (uintptr_t)a.dst < (uintptr_t)b.dst; });
    // 1. destination arrays the row-per-lane kernel takes: rows of four 4-byte elements (or four doubles
    //    restored from f32 chunks) fed by one or two chunks of 4-byte elements on disjoint columns, no
    //    scatter index; plus dense same-type arrays (the chunk IS the array).  One launch per conversion
    //    class; everything else goes to the LDS-tiled kernel below.
    if (N < (1ull << 31) && !tuning().unpack_tiles)
        {
        std::vector<bool> taken(all.size(), false);
        for (int f64 = 0; f64 < 2; f64++)
            {
            while (true)
                {
                UnrowsPlan plan;
                memset(&plan.args, 0, sizeof(plan.args));
                plan.args.N = N;
                plan.f64 = f64 != 0;
                for (size_t i = 0; i < all.size();)
                    {
                    size_t e = i; // [i, e) = the chunks of one destination array (sorted by dst, order kept)
                    while (e < all.size() && all[e].dst == all[i].dst)
                        e++;
                    const UnpackJob& j0 = all[i];
                    const size_t n = e - i;
                    bool ok = !taken[i] && n <= 2 && plan.args.n_groups < ROWS_MAX_GROUPS && (((uintptr_t)j0.dst) & 15) == 0;
                    for (size_t k = i; k < e && ok; k++)
                        {
                        const UnpackJob& j = all[k];
                        ok = j.order == nullptr && j.dst_stride == j0.dst_stride && j.dsz == j0.dsz
                             && (((uintptr_t)j.src) & 15) == 0;
                        }
                    // a dense array of the chunk's own type (any element size, any row width): a plain copy
                    const bool dense = ok && n == 1 && j0.kind == PACK_BITS && j0.ssz == j0.dsz && j0.dst_col0 == 0
                                       && j0.M == j0.dst_stride;
                    if (dense)
                        ok = f64 == 0; // rides along in the launch of the first pass
                    else if (ok)
                        {
                        ok = j0.dst_stride == 4 && (f64 ? j0.dsz == 8 : j0.dsz == 4);
                        for (size_t k = i; k < e && ok; k++)
                            ok = all[k].ssz == 4 && all[k].kind == (uint32_t)(f64 ? PACK_F2F : PACK_BITS) && all[k].M <= 4;
                        if (ok && n == 2) // disjoint columns: no "later chunk wins" question inside a row
                            ok = all[i].dst_col0 + all[i].M <= all[i + 1].dst_col0
                                 || all[i + 1].dst_col0 + all[i + 1].M <= all[i].dst_col0;
                        }
                    if (ok)
                        {
                        UnrowsGroup& g = plan.args.g[plan.args.n_groups++];
                        g.dst = j0.dst;
                        g.a = j0.src;
                        for (size_t k = i; k < e && !dense; k++)
                            if (all[k].fill_rest && !g.fill_on)
                                {
                                g.fill_on = 1;
                                g.fill_lo = (uint32_t)all[k].fill_bits;
                                g.fill_hi = (uint32_t)(all[k].fill_bits >> 32);
                                }
                        if (dense)
                            {
                            const uint64_t bytes = N * (uint64_t)j0.M * j0.ssz;
                            g.copy_vecs = bytes >> 4;
                            g.copy_tail = (uint32_t)(bytes & 15);
                            }
                        else
                            {
                            // `a` = the chunk of the lower columns (xyz before w: the kernel's static hot shape)
                            const UnpackJob& lo = (n == 2 && all[i + 1].dst_col0 < j0.dst_col0) ? all[i + 1] : j0;
                            g.a = lo.src;
                            g.a_nw = lo.M;
                            g.a_col0 = lo.dst_col0;
                            if (n == 2)
                                {
                                const UnpackJob& hi = (&lo == &j0) ? all[i + 1] : j0;
                                g.b = hi.src;
                                g.b_nw = hi.M;
                                g.b_col0 = hi.dst_col0;
                                }
                            }
                        for (size_t k = i; k < e; k++)
                            taken[k] = true;
                        }
                    i = e;
                    }
                if (plan.args.n_groups == 0)
                    break;
                launch_unrows(plan, N, stream);
                }
            }
        std::vector<UnpackJob> rest;
        for (size_t i = 0; i < all.size(); i++)
            if (!taken[i])
                rest.push_back(all[i]);
        all.swap(rest);
        }
    // fills the remaining (tiled / generic) chunks asked for: one pass per destination array over the columns
    // none of ITS chunks writes, ahead of the chunks on the stream
    for (size_t i = 0; i < all.size();)
        {
        size_t e = i;
        while (e < all.size() && all[e].dst == all[i].dst)
            e++;
        uint32_t covered = 0;
        const UnpackJob* want = nullptr;
        for (size_t k = i; k < e; k++)
            {
            for (uint32_t c = 0; c < all[k].M && all[k].dst_col0 + c < 32; c++)
                covered |= 1u << (all[k].dst_col0 + c);
            if (all[k].fill_rest && !want)
                want = &all[k];
            }
        if (want && want->dst_stride <= 32)
            {
            FillArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.dst = want->dst;
            fa.order = want->order;
            fa.N = N;
            fa.bits = want->fill_bits;
            fa.stride = want->dst_stride;
            fa.dsz = want->dsz;
            fa.colmask = ~covered & (want->dst_stride >= 32 ? 0xffffffffu : ((1u << want->dst_stride) - 1u));
            if (fa.colmask)
                {
                const uint64_t lanes = N * (uint64_t)fa.stride;
                hipLaunchKernelGGL(fill_cols_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, fa);
                }
            }
        i = e;
        }
    std::vector<UnpackJob> batch;
    uint32_t sum_rowbytes = 0;
    for (size_t i = 0; i < all.size(); i++)
        {
        const UnpackJob& j = all[i];
        // a chunk that rewrites columns an earlier chunk of the batch wrote goes to the next launch:
        // "the later chunk wins" then holds by stream order
        bool overlap = false;
        for (const UnpackJob& b : batch)
            if (b.dst == j.dst && j.dst_col0 < b.dst_col0 + b.M && b.dst_col0 < j.dst_col0 + j.M)
                overlap = true;
        if (!batch.empty()
            && (overlap || batch.size() == UNPACK_MAX_JOBS || sum_rowbytes + j.rowbytes > UNPACK_MAX_SUM_ROWBYTES))
            {
            launch_unpack_batch(batch, N, stream);
            batch.clear();
            sum_rowbytes = 0;
            }
        batch.push_back(j);
        sum_rowbytes += j.rowbytes;
        }
    if (!batch.empty())
        launch_unpack_batch(batch, N, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        {
        if (err)
            *err = std::string("unpack kernel launch failed: ") + hipGetErrorString(e);
        return PGSD_ERROR_DEVICE;
        }
    return PGSD_SUCCESS;
    }
    } // namespace pgsd_amd

using namespace pgsd_amd;

extern "C" int pgsd_unpack_fields(uint32_t n_jobs, const struct pgsd_unpack_job* jobs, uint64_t N, void* stream)
    try
    {
    if (n_jobs > 0 && !jobs)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!pgsd_device_available())
        {
        set_last_error("pgsd_unpack_fields: no HIP device visible (the HIP path has no CPU fallback)");
        return PGSD_ERROR_NO_DEVICE;
        }
    std::string err;
    int rc = launch_unpack(n_jobs, jobs, N, (hipStream_t)stream, &err);
    if (rc != PGSD_SUCCESS)
        set_last_error(err);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }
