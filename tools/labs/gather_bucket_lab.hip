// tools/labs/gather_bucket_lab.hip -- the bounded attempt at a STREAMING tag-order gather (round 5).
//
// Not product code: a stand-alone experiment.  The product packs a tag-ordered snapshot as
//     chunk[t] = src[order[t]]                        (order = HOOMD's reverse-tag array)
// with one random 16-byte row read per particle and array: a 64-byte sector per 16-byte row, 3.3 x read
// amplification, 0.17 of the HBM peak (profiles/r02_pmc_fetch_size_gather_10M.csv).  Tags are a permutation of
// 0 .. N-1, so the alternative is a two-pass bucketed gather whose HBM traffic is streaming only:
//
//   pass 1 (bucket_pass1): stream pos / vel / tag in MEMORY order (tag[i] = destination row of source row i,
//           HOOMD's tag array = the inverse of `order`); a chunk of T*U rows per workgroup is ranked into buckets
//           of 2^shift destination rows with LDS atomics (histogram, the returned value is the row's rank inside
//           its bucket), one global atomicAdd per non-empty bucket reserves the chunk's run in the bucket, the rows
//           are stored at bucket base + reservation + rank.  A bucket of tag range W holds exactly W rows: static
//           offsets, no scan.
//   pass 2 (bucket_pass2): G = 2^shift / W workgroups share one bucket (same blockIdx % 8: one XCD's L2 under
//           round-robin placement, speed only); workgroup q scans the bucket's tags, lists the slots whose tag
//           falls in its W destination rows, brings those records into an LDS tile in destination order and
//           emits N x 3 / N x 3 / N x 1 with row-per-lane coalesced stores.
//
// Measured beside: `direct` (row-per-lane random gather, the shape of the product's kernel) and, when the
// library is given, the product's own pgsd_pack_fields with `order`.  Permutations: uniform random; identity;
// `hilbert` = particles created in lattice order (tag = lattice index) and kept in memory along a 3-D Hilbert
// curve through their positions -- what HOOMD's SFC sorter leaves for a dump writer.
// Every variant is checked against the host gather before it is timed.
//
//   hipcc -O3 --offload-arch=gfx950 tools/labs/gather_bucket_lab.hip -o gpurun_out/gather_bucket_lab -ldl
//   ./gather_bucket_lab [N=10000000] [reps=20] [libpgsd_amd.so]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <numeric>
#include <random>
#include <string>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));

#define CK(x)                                                                                 \
    do                                                                                        \
        {                                                                                     \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess)                                                                 \
            {                                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                          \
            }                                                                                 \
        } while (0)

// ---------------------------------------------------------------- direct gather (baseline shape)
template<int T, int U, bool SPLIT = false>
__global__ __launch_bounds__(T) void direct_gather(const u32x4* __restrict__ pos, const u32x4* __restrict__ vel,
                                                    const uint32_t* __restrict__ order, uint64_t N, uint32_t* opos,
                                                    uint32_t* ovel, uint32_t* otid)
    {
    if constexpr (SPLIT)
        {
        // blockIdx.y = source array: a wave keeps ONE random input stream open (the product's row kernel deals its
        // workgroups to source arrays the same way)
        const uint64_t base = (uint64_t)blockIdx.x * (T * U);
        const u32x4* src = blockIdx.y ? vel : pos;
        uint32_t* out = blockIdx.y ? ovel : opos;
        uint32_t o[U];
        u32x4 p[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            {
            const uint64_t t = base + (uint64_t)u * T + threadIdx.x;
            o[u] = t < N ? __builtin_nontemporal_load(order + t) : 0u;
            }
#pragma unroll
        for (int u = 0; u < U; u++)
            p[u] = src[o[u]];
#pragma unroll
        for (int u = 0; u < U; u++)
            {
            const uint64_t t = base + (uint64_t)u * T + threadIdx.x;
            if (t < N)
                {
                u32x3 a = {p[u].x, p[u].y, p[u].z};
                __builtin_nontemporal_store(a, (u32x3_a4*)(out + 3 * t));
                if (blockIdx.y == 0)
                    __builtin_nontemporal_store(p[u].w, otid + t);
                }
            }
        return;
        }
    const uint64_t base = (uint64_t)blockIdx.x * (T * U);
    uint32_t o[U];
    u32x4 p[U], v[U];
#pragma unroll
    for (int u = 0; u < U; u++)
        {
        const uint64_t t = base + (uint64_t)u * T + threadIdx.x;
        o[u] = t < N ? order[t] : 0u;
        }
#pragma unroll
    for (int u = 0; u < U; u++)
        {
        p[u] = pos[o[u]];
        v[u] = vel[o[u]];
        }
#pragma unroll
    for (int u = 0; u < U; u++)
        {
        const uint64_t t = base + (uint64_t)u * T + threadIdx.x;
        if (t < N)
            {
            u32x3 a = {p[u].x, p[u].y, p[u].z}, b = {v[u].x, v[u].y, v[u].z};
            __builtin_nontemporal_store(a, (u32x3_a4*)(opos + 3 * t));
            __builtin_nontemporal_store(b, (u32x3_a4*)(ovel + 3 * t));
            __builtin_nontemporal_store(p[u].w, otid + t);
            }
        }
    }

// ---------------------------------------------------------------- pass 1: rows -> buckets of destination rows
// err[0] counts rows that could not be placed (tag out of range, bucket full: the input was no permutation)
template<int T, int U>
__global__ __launch_bounds__(T) void bucket_pass1(const u32x4* __restrict__ pos, const u32x4* __restrict__ vel,
                                                   const uint32_t* __restrict__ tag, uint64_t N, uint32_t shift, uint32_t K,
                                                   uint32_t* cursor, u32x4* ipos, u32x4* ivel, uint32_t* itag, uint32_t* err)
    {
    extern __shared__ uint32_t lds[];
    uint32_t* hist = lds;
    uint32_t* gbase = lds + K;
    const uint32_t tid = threadIdx.x;
    const uint64_t C = (uint64_t)T * U;
    const uint64_t n_chunks = (N + C - 1) / C;
    const uint32_t Wb = 1u << shift;
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x)
        {
        for (uint32_t b = tid; b < K; b += T)
            hist[b] = 0;
        __syncthreads();
        const uint64_t base = chunk * C;
        uint32_t t[U], r[U];
        u32x4 p[U], v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            {
            const uint64_t i = base + (uint64_t)u * T + tid;
            t[u] = i < N ? __builtin_nontemporal_load(tag + i) : 0xffffffffu;
            }
#pragma unroll
        for (int u = 0; u < U; u++)
            {
            const uint64_t i = base + (uint64_t)u * T + tid;
            if (i < N)
                {
                p[u] = __builtin_nontemporal_load(pos + i);
                v[u] = __builtin_nontemporal_load(vel + i);
                }
            }
#pragma unroll
        for (int u = 0; u < U; u++)
            {
            r[u] = 0;
            if ((uint64_t)t[u] < N)
                r[u] = atomicAdd(&hist[t[u] >> shift], 1u);
            }
        __syncthreads();
        for (uint32_t b = tid; b < K; b += T)
            {
            const uint32_t c = hist[b];
            if (c)
                gbase[b] = atomicAdd(&cursor[b], c);
            }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++)
            {
            const uint64_t i = base + (uint64_t)u * T + tid;
            if (i >= N)
                continue;
            if ((uint64_t)t[u] >= N)
                {
                atomicAdd(err, 1u);
                continue;
                }
            const uint32_t b = t[u] >> shift;
            const uint64_t row0 = (uint64_t)b << shift;
            const uint64_t cap = (N - row0 < (uint64_t)Wb) ? N - row0 : (uint64_t)Wb;
            const uint64_t slot = (uint64_t)gbase[b] + r[u];
            if (slot < cap)
                {
                ipos[row0 + slot] = p[u];
                ivel[row0 + slot] = v[u];
                itag[row0 + slot] = t[u];
                }
            else
                atomicAdd(err, 1u);
            }
        // (the next iteration's zeroing of hist is separated from this iteration's reads of hist by the barriers
        // above; gbase is rewritten only behind the next iteration's first barrier)
        }
    }

// ---------------------------------------------------------------- pass 2: bucket -> chunks in destination order
template<int T, int LOGW>
__global__ __launch_bounds__(T) void bucket_pass2(const u32x4* __restrict__ ipos, const u32x4* __restrict__ ivel,
                                                   const uint32_t* __restrict__ itag, uint64_t N, uint32_t shift, uint32_t G,
                                                   uint32_t* opos, uint32_t* ovel, uint32_t* otid, uint32_t* err)
    {
    constexpr uint32_t W = 1u << LOGW;
    __shared__ u32x4 lpos[W];
    __shared__ u32x4 lvel[W];
    __shared__ uint32_t list[W];
    __shared__ uint32_t n_list;
    const uint32_t tid = threadIdx.x;
    // blocks i, i + 8, i + 16 ... share an XCD under round-robin placement: the G workgroups of a bucket are
    // G consecutive blocks of one residue class
    const uint32_t x = blockIdx.x & 7u, j8 = blockIdx.x >> 3;
    const uint32_t q = j8 % G;
    const uint64_t b = (uint64_t)(j8 / G) * 8 + x;
    const uint64_t row0 = b << shift;
    if (row0 >= N)
        return;
    const uint32_t Wb = 1u << shift;
    const uint32_t cnt = (uint32_t)((N - row0 < (uint64_t)Wb) ? N - row0 : (uint64_t)Wb);
    const uint32_t lo0 = q * W;
    if (lo0 >= cnt)
        return;
    const uint32_t rows = (cnt - lo0 < W) ? cnt - lo0 : W;
    if (tid == 0)
        n_list = 0;
    __syncthreads();
    // A: scan the bucket's tags, list the slots that belong to this workgroup's destination rows
    for (uint32_t j = tid; j < cnt; j += T)
        {
        const uint32_t lo = itag[row0 + j] - (uint32_t)row0 - lo0; // wraps for tags below the window
        if (lo < rows)
            {
            const uint32_t k = atomicAdd(&n_list, 1u);
            if (k < W)
                list[k] = (j << LOGW) | lo; // j < 2^shift <= 2^16, lo < 2^LOGW <= 2^12
            }
        }
    __syncthreads();
    const uint32_t n = n_list < rows ? n_list : rows;
    if (n_list != rows && tid == 0)
        atomicAdd(err, 1u);
    // B: bring the listed records into the tile, in destination order (4 records in flight per lane)
        {
        uint32_t k = tid;
        for (; k + 3 * T < n; k += 4 * T)
            {
            const uint32_t e0 = list[k], e1 = list[k + T], e2 = list[k + 2 * T], e3 = list[k + 3 * T];
            const u32x4 p0 = ipos[row0 + (e0 >> LOGW)], v0 = ivel[row0 + (e0 >> LOGW)];
            const u32x4 p1 = ipos[row0 + (e1 >> LOGW)], v1 = ivel[row0 + (e1 >> LOGW)];
            const u32x4 p2 = ipos[row0 + (e2 >> LOGW)], v2 = ivel[row0 + (e2 >> LOGW)];
            const u32x4 p3 = ipos[row0 + (e3 >> LOGW)], v3 = ivel[row0 + (e3 >> LOGW)];
            lpos[e0 & (W - 1)] = p0;
            lvel[e0 & (W - 1)] = v0;
            lpos[e1 & (W - 1)] = p1;
            lvel[e1 & (W - 1)] = v1;
            lpos[e2 & (W - 1)] = p2;
            lvel[e2 & (W - 1)] = v2;
            lpos[e3 & (W - 1)] = p3;
            lvel[e3 & (W - 1)] = v3;
            }
        for (; k < n; k += T)
            {
            const uint32_t e = list[k];
            lpos[e & (W - 1)] = ipos[row0 + (e >> LOGW)];
            lvel[e & (W - 1)] = ivel[row0 + (e >> LOGW)];
            }
        }
    __syncthreads();
    // C: one destination row per lane: 768-byte / 256-byte contiguous wave stores
    const uint64_t g0 = row0 + lo0;
    for (uint32_t r = tid; r < rows; r += T)
        {
        const u32x4 p = lpos[r], v = lvel[r];
        u32x3 a = {p.x, p.y, p.z}, c = {v.x, v.y, v.z};
        __builtin_nontemporal_store(a, (u32x3_a4*)(opos + 3 * (g0 + r)));
        __builtin_nontemporal_store(c, (u32x3_a4*)(ovel + 3 * (g0 + r)));
        __builtin_nontemporal_store(p.w, otid + g0 + r);
        }
    }

// ---------------------------------------------------------------- host side
static void hilbert_axes_to_transpose(uint32_t* X, int bits)
    {
    // Skilling, "Programming the Hilbert curve" (AIP Conf. Proc. 707, 2004): in-place transform of the
    // coordinates into the transposed Hilbert index
    const int n = 3;
    uint32_t M = 1u << (bits - 1), P, Q, t;
    for (Q = M; Q > 1; Q >>= 1)
        {
        P = Q - 1;
        for (int i = 0; i < n; i++)
            if (X[i] & Q)
                X[0] ^= P;
            else
                {
                t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
                }
        }
    for (int i = 1; i < n; i++)
        X[i] ^= X[i - 1];
    t = 0;
    for (Q = M; Q > 1; Q >>= 1)
        if (X[n - 1] & Q)
            t ^= Q - 1;
    for (int i = 0; i < n; i++)
        X[i] ^= t;
    }

static uint64_t hilbert_key(uint32_t x, uint32_t y, uint32_t z, int bits)
    {
    uint32_t X[3] = {x, y, z};
    hilbert_axes_to_transpose(X, bits);
    uint64_t key = 0;
    for (int b = bits - 1; b >= 0; b--)
        for (int i = 0; i < 3; i++)
            key = (key << 1) | ((X[i] >> b) & 1u);
    return key;
    }

// tag_of[i] = tag of the particle kept at memory row i
static std::vector<uint32_t> make_perm(const std::string& kind, uint64_t N)
    {
    std::vector<uint32_t> tag_of(N);
    std::iota(tag_of.begin(), tag_of.end(), 0u);
    if (kind == "uniform")
        {
        std::mt19937_64 g(1234);
        std::shuffle(tag_of.begin(), tag_of.end(), g);
        }
    else if (kind == "hilbert")
        {
        const uint32_t n = (uint32_t)std::ceil(std::cbrt((double)N));
        int bits = 1;
        while ((1u << bits) < n)
            bits++;
        std::vector<uint64_t> key(N);
        for (uint64_t t = 0; t < N; t++)
            key[t] = hilbert_key((uint32_t)(t % n), (uint32_t)((t / n) % n), (uint32_t)(t / ((uint64_t)n * n)), bits);
        std::sort(tag_of.begin(), tag_of.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        }
    else if (kind != "identity")
        {
        fprintf(stderr, "unknown permutation %s\n", kind.c_str());
        exit(2);
        }
    return tag_of;
    }

struct PackJob // include/pgsd.h: struct pgsd_pack_job
    {
    void* dst;
    uint32_t dst_type, M;
    const void* src;
    const uint32_t* order;
    uint32_t src_type, src_stride, src_col0, bitcast;
    };
typedef int (*pack_fields_fn)(uint32_t, const PackJob*, uint64_t, void*, float*);

int main(int argc, char** argv)
    {
    const uint64_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const char* libpath = argc > 3 ? argv[3] : nullptr;
    if (N == 0 || N >= (1ull << 32))
        {
        fprintf(stderr, "N out of range\n");
        return 2;
        }
    pack_fields_fn pack_fields = nullptr;
    if (libpath)
        {
        void* h = dlopen(libpath, RTLD_NOW);
        if (h)
            pack_fields = (pack_fields_fn)dlsym(h, "pgsd_pack_fields");
        if (!pack_fields)
            fprintf(stderr, "no pgsd_pack_fields in %s (%s)\n", libpath, dlerror());
        }

    std::vector<float> hpos(4 * N), hvel(4 * N);
        {
        std::mt19937 g(7);
        std::uniform_real_distribution<float> U(-50.f, 50.f);
        for (uint64_t i = 0; i < N; i++)
            {
            for (int c = 0; c < 3; c++)
                {
                hpos[4 * i + c] = U(g);
                hvel[4 * i + c] = U(g) * 0.02f;
                }
            uint32_t tid = (uint32_t)(i % 7);
            memcpy(&hpos[4 * i + 3], &tid, 4);
            hvel[4 * i + 3] = 1.0f;
            }
        }
    u32x4 *dpos, *dvel, *ipos, *ivel;
    uint32_t *dtag, *dorder, *itag, *opos, *ovel, *otid, *cursor, *derr;
    CK(hipMalloc(&dpos, 16 * N));
    CK(hipMalloc(&dvel, 16 * N));
    CK(hipMalloc(&ipos, 16 * N));
    CK(hipMalloc(&ivel, 16 * N));
    CK(hipMalloc(&dtag, 4 * N));
    CK(hipMalloc(&dorder, 4 * N));
    CK(hipMalloc(&itag, 4 * N));
    CK(hipMalloc(&opos, 12 * N));
    CK(hipMalloc(&ovel, 12 * N));
    CK(hipMalloc(&otid, 4 * N));
    CK(hipMalloc(&cursor, 4 * 65536));
    CK(hipMalloc(&derr, 4));
    CK(hipMemcpy(dpos, hpos.data(), 16 * N, hipMemcpyHostToDevice));
    CK(hipMemcpy(dvel, hvel.data(), 16 * N, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));
    std::vector<uint32_t> want_pos(3 * N), want_vel(3 * N), want_tid(N), got(3 * N);

    const char* kinds[] = {"uniform", "hilbert", "identity"};
    for (const char* kind : kinds)
        {
        std::vector<uint32_t> tag_of = make_perm(kind, N), order(N);
        for (uint64_t i = 0; i < N; i++)
            order[tag_of[i]] = (uint32_t)i;
        CK(hipMemcpy(dtag, tag_of.data(), 4 * N, hipMemcpyHostToDevice));
        CK(hipMemcpy(dorder, order.data(), 4 * N, hipMemcpyHostToDevice));
        for (uint64_t t = 0; t < N; t++)
            {
            const uint64_t i = order[t];
            memcpy(&want_pos[3 * t], &hpos[4 * i], 12);
            memcpy(&want_vel[3 * t], &hvel[4 * i], 12);
            memcpy(&want_tid[t], &hpos[4 * i + 3], 4);
            }
        auto clear_out = [&]()
        {
            CK(hipMemsetAsync(opos, 0xff, 12 * N, s));
            CK(hipMemsetAsync(ovel, 0xff, 12 * N, s));
            CK(hipMemsetAsync(otid, 0xff, 4 * N, s));
            CK(hipStreamSynchronize(s));
        };
        auto check = [&](const char* what) -> bool
        {
            bool ok = true;
            CK(hipMemcpy(got.data(), opos, 12 * N, hipMemcpyDeviceToHost));
            ok = ok && memcmp(got.data(), want_pos.data(), 12 * N) == 0;
            CK(hipMemcpy(got.data(), ovel, 12 * N, hipMemcpyDeviceToHost));
            ok = ok && memcmp(got.data(), want_vel.data(), 12 * N) == 0;
            CK(hipMemcpy(got.data(), otid, 4 * N, hipMemcpyDeviceToHost));
            ok = ok && memcmp(got.data(), want_tid.data(), 4 * N) == 0;
            if (!ok)
                fprintf(stderr, "MISMATCH: %s / %s\n", kind, what);
            return ok;
        };

        // ---- direct gather, several launch shapes
        auto direct = [&](auto launch, const char* name)
        {
            clear_out();
            launch();
            CK(hipStreamSynchronize(s));
            const bool ok = check(name);
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; r++)
                launch();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"lab\": \"gather_bucket\", \"N\": %llu, \"perm\": \"%s\", \"variant\": \"direct %s\", \"us\": %.1f, \"ok\": %s}\n",
                   (unsigned long long)N, kind, name, ms * 1e3 / reps, ok ? "true" : "false");
            fflush(stdout);
        };
#define DIRECT(T, U)                                                                                                  \
    direct([&]() { direct_gather<T, U><<<(unsigned)((N + T * U - 1) / (T * U)), T, 0, s>>>(dpos, dvel, dorder, N, opos, ovel, otid); }, \
           #T "x" #U);
#define DIRECT_SPLIT(T, U)                                                                                            \
    direct([&]() { direct_gather<T, U, true><<<dim3((unsigned)((N + T * U - 1) / (T * U)), 2), T, 0, s>>>(dpos, dvel, dorder, N, opos, ovel, otid); }, \
           #T "x" #U " split");
        DIRECT(256, 4)
        DIRECT(256, 2)
        DIRECT(256, 8)
        DIRECT(128, 4)
        DIRECT(64, 4)
        DIRECT(64, 8)
        DIRECT(512, 4)
        DIRECT(1024, 2)
        DIRECT_SPLIT(256, 4)
        DIRECT_SPLIT(256, 8)
        DIRECT_SPLIT(128, 8)
        DIRECT_SPLIT(64, 8)
        DIRECT_SPLIT(512, 8)
        // ---- the product's kernel through the C ABI
        if (pack_fields)
            {
            PackJob jobs[3] = {{opos, 9, 3, dpos, dorder, 9, 4, 0, 0}, {otid, 3, 1, dpos, dorder, 9, 4, 3, 1},
                               {ovel, 9, 3, dvel, dorder, 9, 4, 0, 0}};
            clear_out();
            int rc = pack_fields(3, jobs, N, s, nullptr);
            CK(hipStreamSynchronize(s));
            const bool ok = rc == 0 && check("product");
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; r++)
                pack_fields(3, jobs, N, s, nullptr);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"lab\": \"gather_bucket\", \"N\": %llu, \"perm\": \"%s\", \"variant\": \"product pgsd_pack_fields(order)\", \"us\": %.1f, \"ok\": %s}\n",
                   (unsigned long long)N, kind, ms * 1e3 / reps, ok ? "true" : "false");
            fflush(stdout);
            }
        // ---- two-pass bucketed
        auto two_pass = [&](auto p1, const char* p1name, int C, auto p2, const char* p2name, uint32_t logw, uint32_t shift, int T2)
        {
            const uint32_t Wb = 1u << shift, W = 1u << logw;
            if (W > Wb)
                return;
            const uint32_t G = Wb / W;
            const uint32_t K = (uint32_t)((N + Wb - 1) / Wb);
            const uint32_t K8 = (K + 7) / 8 * 8;
            const uint64_t n_chunks = (N + C - 1) / C;
            const unsigned b1 = (unsigned)std::min<uint64_t>(n_chunks, 512);
            const unsigned b2 = K8 * G;
            const size_t lds1 = (size_t)K * 8;
            auto run = [&]()
            {
                CK(hipMemsetAsync(cursor, 0, 4 * (size_t)K, s));
                CK(hipEventRecord(e0, s));
                p1(b1, lds1, shift, K);
                CK(hipEventRecord(e1, s));
                p2(b2, shift, G);
                CK(hipEventRecord(e2, s));
            };
            clear_out();
            CK(hipMemsetAsync(derr, 0, 4, s));
            run();
            CK(hipStreamSynchronize(s));
            uint32_t herr = 0;
            CK(hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost));
            const bool ok = (hipGetLastError() == hipSuccess) && herr == 0 && check("two-pass");
            float t1 = 0, t2 = 0, tt = 0;
            hipEvent_t a, b;
            CK(hipEventCreate(&a));
            CK(hipEventCreate(&b));
            CK(hipEventRecord(a, s));
            for (int r = 0; r < reps; r++)
                {
                run();
                CK(hipEventSynchronize(e2));
                float x;
                CK(hipEventElapsedTime(&x, e0, e1));
                t1 += x;
                CK(hipEventElapsedTime(&x, e1, e2));
                t2 += x;
                }
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            // back to back (no host wait between repetitions)
            CK(hipEventRecord(a, s));
            for (int r = 0; r < reps; r++)
                run();
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&tt, a, b));
            CK(hipEventDestroy(a));
            CK(hipEventDestroy(b));
            printf("{\"lab\": \"gather_bucket\", \"N\": %llu, \"perm\": \"%s\", \"variant\": \"two-pass %s + %s shift=%u W=%u K=%u G=%u\", "
                   "\"us\": %.1f, \"pass1_us\": %.1f, \"pass2_us\": %.1f, \"err_rows\": %u, \"ok\": %s}\n",
                   (unsigned long long)N, kind, p1name, p2name, shift, W, K, G, tt * 1e3 / reps, t1 * 1e3 / reps, t2 * 1e3 / reps, herr,
                   ok ? "true" : "false");
            fflush(stdout);
        };
        auto p1_1024x8 = [&](unsigned blocks, size_t lds, uint32_t shift, uint32_t K)
        { bucket_pass1<1024, 8><<<std::min(blocks, 256u), 1024, lds, s>>>(dpos, dvel, dtag, N, shift, K, cursor, ipos, ivel, itag, derr); };
        auto p1_512x8 = [&](unsigned blocks, size_t lds, uint32_t shift, uint32_t K)
        { bucket_pass1<512, 8><<<blocks, 512, lds, s>>>(dpos, dvel, dtag, N, shift, K, cursor, ipos, ivel, itag, derr); };
        auto p1_256x8 = [&](unsigned blocks, size_t lds, uint32_t shift, uint32_t K)
        { bucket_pass1<256, 8><<<std::min(2 * blocks, 1024u), 256, lds, s>>>(dpos, dvel, dtag, N, shift, K, cursor, ipos, ivel, itag, derr); };
        auto p2_512_11 = [&](unsigned blocks, uint32_t shift, uint32_t G)
        { bucket_pass2<512, 11><<<blocks, 512, 0, s>>>(ipos, ivel, itag, N, shift, G, opos, ovel, otid, derr); };
        auto p2_512_10 = [&](unsigned blocks, uint32_t shift, uint32_t G)
        { bucket_pass2<512, 10><<<blocks, 512, 0, s>>>(ipos, ivel, itag, N, shift, G, opos, ovel, otid, derr); };
        auto p2_1024_12 = [&](unsigned blocks, uint32_t shift, uint32_t G)
        { bucket_pass2<1024, 12><<<blocks, 1024, 0, s>>>(ipos, ivel, itag, N, shift, G, opos, ovel, otid, derr); };
        if (getenv("LAB_DIRECT_ONLY"))
            continue;
        for (uint32_t shift : {13u, 14u, 15u, 16u})
            {
            two_pass(p1_1024x8, "p1 1024x8", 8192, p2_512_11, "p2 512", 11, shift, 512);
            two_pass(p1_512x8, "p1 512x8", 4096, p2_512_11, "p2 512", 11, shift, 512);
            }
        two_pass(p1_256x8, "p1 256x8", 2048, p2_512_11, "p2 512", 11, 15, 512);
        two_pass(p1_1024x8, "p1 1024x8", 8192, p2_512_10, "p2 512", 10, 14, 512);
        two_pass(p1_1024x8, "p1 1024x8", 8192, p2_1024_12, "p2 1024", 12, 15, 1024);
        two_pass(p1_1024x8, "p1 1024x8", 8192, p2_1024_12, "p2 1024", 12, 16, 1024);
        }
    return 0;
    }
