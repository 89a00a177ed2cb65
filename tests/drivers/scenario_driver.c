/* scenario_driver.c -- replay a chunk-write scenario through the pgsd C ABI.
 *
 * Test infrastructure.  ONE source, FOUR builds:
 *
 *   -DPGSD_DRIVER_REF   links the reference's own pgsd.c (compiled in place from
 *                       /root/reference by oracle/Makefile, output in oracle/_ref/)
 *                       and runs under MPICH `mpiexec -n P`.  This produces the golden
 *                       .gsd files committed under tests/golden/.
 *   (default)           links libpgsd_amd.so (this repo's product) and takes
 *                       rank/size from the shared-memory communicator
 *                       (PGSD_RANK / PGSD_NRANKS / PGSD_SHM_NAME), one process per rank.
 *
 *   -DPGSD_DRIVER_MPI   links libpgsd_amd.so AND MPI: the product under the reference's own
 *                       launcher (mpiexec), its collectives forwarded to MPI_Allgather through
 *                       the communicator vtable.
 *
 *   -DPGSD_DRIVER_DEVICE (a fourth build of the PRODUCT, needs the HIP runtime): as the default build, plus the
 *                       script command `device <0|1|2>` -- from then on the rows of every chunk write are uploaded
 *                       to HBM (1: as a dense N x M array; 2: inside wider rows, x/y/z of a Scalar4 and the like, so
 *                       that the strided pack kernels run) and written through pgsd_write_chunk_device, the device
 *                       twin of pgsd_write_chunk with the same arguments: every reference-written golden is then
 *                       reproduced FROM HBM (tests/test_gpu_golden_device.py).
 *   PGSD_DRIVER_THREADS=P (environment, default and device builds): the P ranks are P THREADS of this process, each
 *                       with a communicator of its own (pgsd_comm_create_shm + pgsd_create_and_open_on) -- the GPU
 *                       boxes admit six processes on the card, the goldens go up to eight ranks.
 *
 * All builds make exactly the same pgsd_* calls (the drop-in boundary of
 * /root/reference/pgsd/pgsd/pgsd.h:362-735), so `cmp` of the two output files is the
 * parity check of the host file layer.
 *
 * The call sequence per chunk follows the two in-tree callers of the reference:
 *   - per-particle arrays (all=1): benchmark-write.cc:91-102 and fl.pyx:594-652
 *       N=N_local, N_global=sum, offset=M*sum_{j<rank}, global_size=N_global*M
 *   - replicated small chunks (all=0): fl.pyx:594-652 with offset=None
 *       N_global=N, offset=0, global_size=N*M
 *
 * Script grammar (one command per line, '#' starts a comment):
 *   prefill <file>                 (rank 0 copies <file> -- relative names are looked up in ../files beside the
 *                                   script's directory -- to the output path, then all ranks meet at a barrier:
 *                                   scenarios that start from an existing file, e.g. the reference's GSD v1.0
 *                                   test fixture, continue with `open`)
 *   create <application> <schema> <major> <minor> <rw|append> <excl 0|1>
 *   open <rw|ro|append>
 *   seed <u64>
 *   chunk <name> <u8|u16|u32|u64|i8|i16|i32|i64|f32|f64> <M> <all 0|1> <dist>
 *         dist = even:<Nglobal> | same:<N> | list:<n0>,<n1>,...   (list is cycled over ranks)
 *   rawchunk <name> <type> <N> <M> <N_global> <M_global> <offset> <global_size> <all>
 *         (every rank passes these very arguments; the data differs per rank)
 *   samechunk <name> <type> <N> <M> <N_global> <M_global> <offset> <global_size> <all>
 *         (as rawchunk with the SAME data on every rank: with all=1 and offset=0 this is what
 *          fl.pyx's write_chunk(name, data) passes by default, fl.pyx:526, 592-598, 640-652 --
 *          the ranks' writes overlap, identical bytes keep the file deterministic)
 *   chunkgs <name> <type> <M> <all> <dist> <global_size>
 *         (as chunk, but the caller's global_size argument is the given value, right or wrong:
 *          the reference ignores it, pgsd.c:2147-2151, 2240-2246)
 *   batch <0|1|2|3>                (product only: pgsd_set_frame_exchange; 2 = batched + pgsd_set_deferred_rows: the
 *                                   driver then keeps every chunk's rows until the next end_frame / flush / close / dump;
 *                                   3 = declared partition: before every `chunk` the driver calls pgsd_set_partition
 *                                   with that chunk's row counts and writes per-particle chunks with
 *                                   PGSD_PARTITION_AUTO -- no exchange; chunks whose sizes the declaration cannot
 *                                   express (chunkgs / rawchunk / samechunk, replicated chunks of unequal size) clear it;
 *                                   ignored by the reference build.
 *                                   `dump` then performs the pending exchange first, so that the trace
 *                                   shows the same file_size the unbatched run shows)
 *   device <0|1|2>                 (PGSD_DRIVER_DEVICE build only, see above; refused by the other product builds,
 *                                   ignored by the reference build)
 *   async <0|1>                    (product builds: `end_frame` seals with pgsd_end_frame_async -- metadata committed at
 *                                   once, device copies and pwrites running on behind the caller -- instead of
 *                                   pgsd_end_frame; the layout must not change by a byte; ignored by the reference build)
 *   end_frame | flush | close | dump
 *   maxbuf <bytes> | idxbuf <entries>
 *   find <frame> <name>            (prints found/N/M/type/location on rank 0)
 *   read <frame> <name> <N> <M> <row_offset> <all 0|1> <bufN> <bufM> <size>
 *         pgsd_find_chunk + pgsd_read_chunk (pgsd.h:581-610) on every rank (the reference's read is
 *         collective: rank 0's entry is broadcast, pgsd.c:2491-2494); all=0 reads the whole chunk, all=1
 *         rows [row_offset, row_offset+N) x M.  <bufN> x <bufM> elements of <size> bytes is what the
 *         buffer must hold (ranks other than 0 cannot look at the entry in the reference).  Rank 0 prints
 *         the return code and an FNV-1a hash of the bytes it read.
 *   names <prefix>                 (prints matching chunk names on rank 0)
 */
#include "pgsd.h"
#if !defined(PGSD_DRIVER_REF)
#include "pgsd_private.h" /* queue plumbing of the product: pgsd_set_deferred_rows, pgsd_frame_exchange */
#endif

#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(PGSD_DRIVER_REF) || defined(PGSD_DRIVER_MPI)
#include <mpi.h>
#else
#include <pthread.h>
#endif

#ifdef PGSD_DRIVER_DEVICE
#include <hip/hip_runtime_api.h>
#endif

#ifdef PGSD_DRIVER_MPI
/* third build: the PRODUCT under the reference's own launcher (mpiexec); the library's
   collectives are forwarded to MPI through the communicator vtable */
static int mpi_allgather_cb(void* ctx, const void* send, void* recv, size_t bytes)
    {
    (void)ctx;
    return MPI_Allgather((void*)send, (int)bytes, MPI_BYTE, recv, (int)bytes, MPI_BYTE, MPI_COMM_WORLD) != MPI_SUCCESS;
    }
static int mpi_barrier_cb(void* ctx)
    {
    (void)ctx;
    return MPI_Barrier(MPI_COMM_WORLD) != MPI_SUCCESS;
    }
#endif

/* per rank: a rank is a process -- or, with PGSD_DRIVER_THREADS, a thread */
static __thread int g_rank = 0, g_size = 1;
#if !defined(PGSD_DRIVER_REF) && !defined(PGSD_DRIVER_MPI)
static __thread struct pgsd_comm g_comm; /* thread ranks: this rank's own communicator */
static __thread int g_own_comm = 0;
#endif

static uint64_t mix64(uint64_t seed, uint64_t gid, uint32_t c, uint32_t M)
    {
    uint64_t u = (gid * (uint64_t)M + (uint64_t)c) * 0x9E3779B97F4A7C15ull
                 + seed * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull;
    u ^= u >> 29;
    u *= 0xBF58476D1CE4E5B9ull;
    u ^= u >> 32;
    return u;
    }

static int parse_type(const char* s, size_t* sz)
    {
    static const char* names[] = {"u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64", "f32", "f64"};
    static const size_t sizes[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8};
    for (int i = 0; i < 10; i++)
        {
        if (strcmp(s, names[i]) == 0)
            {
            *sz = sizes[i];
            return i + 1; /* pgsd_type values 1..10, pgsd.h:38-69 */
            }
        }
    fprintf(stderr, "bad type %s\n", s);
    exit(2);
    }

/* fill rows [gid0, gid0+N) x M of the given pgsd type */
static void* gen_data(int type, uint64_t seed, uint64_t gid0, uint64_t N, uint32_t M, size_t sz)
    {
    if (N == 0)
        return NULL;
    char* buf = (char*)malloc(N * M * sz);
    for (uint64_t i = 0; i < N; i++)
        for (uint32_t c = 0; c < M; c++)
            {
            uint64_t u = mix64(seed, gid0 + i, c, M);
            char* p = buf + (i * M + c) * sz;
            double d = ((double)(u % 2000001ull) - 1000000.0) * 0.001;
            switch (type)
                {
                case 1: { uint8_t v = (uint8_t)u; memcpy(p, &v, 1); break; }
                case 2: { uint16_t v = (uint16_t)u; memcpy(p, &v, 2); break; }
                case 3: { uint32_t v = (uint32_t)u; memcpy(p, &v, 4); break; }
                case 4: { uint64_t v = u; memcpy(p, &v, 8); break; }
                case 5: { uint8_t v = (uint8_t)u; memcpy(p, &v, 1); break; }
                case 6: { uint16_t v = (uint16_t)u; memcpy(p, &v, 2); break; }
                case 7: { uint32_t v = (uint32_t)u; memcpy(p, &v, 4); break; }
                case 8: { uint64_t v = u; memcpy(p, &v, 8); break; }
                case 9: { float v = (float)d; memcpy(p, &v, 4); break; }
                case 10: { memcpy(p, &d, 8); break; }
                }
            }
    return buf;
    }

static void dist_counts(const char* dist, uint64_t* counts)
    {
    if (strncmp(dist, "even:", 5) == 0)
        {
        /* benchmark-write.cc:33-37 */
        uint64_t n = strtoull(dist + 5, NULL, 10);
        for (int r = 0; r < g_size; r++)
            counts[r] = n / g_size + (((uint64_t)r < n % g_size) ? 1 : 0);
        }
    else if (strncmp(dist, "same:", 5) == 0)
        {
        uint64_t n = strtoull(dist + 5, NULL, 10);
        for (int r = 0; r < g_size; r++)
            counts[r] = n;
        }
    else if (strncmp(dist, "list:", 5) == 0)
        {
        uint64_t vals[64];
        int nv = 0;
        char tmp[512];
        strncpy(tmp, dist + 5, sizeof(tmp) - 1);
        tmp[sizeof(tmp) - 1] = 0;
        char* save = NULL; /* (strtok_r: the ranks may be threads) */
        for (char* tok = strtok_r(tmp, ",", &save); tok && nv < 64; tok = strtok_r(NULL, ",", &save))
            vals[nv++] = strtoull(tok, NULL, 10);
        for (int r = 0; r < g_size; r++)
            counts[r] = vals[r % nv];
        }
    else
        {
        fprintf(stderr, "bad dist %s\n", dist);
        exit(2);
        }
    }

static void all_ranks_meet(void)
    {
#if defined(PGSD_DRIVER_REF) || defined(PGSD_DRIVER_MPI)
    MPI_Barrier(MPI_COMM_WORLD);
#else
    if (g_own_comm)
        {
        char one = 0, all[1024];
        if (g_comm.barrier)
            g_comm.barrier(g_comm.ctx);
        else
            g_comm.allgather(g_comm.ctx, &one, all, 1);
        }
    else
        pgsd_comm_barrier();
#endif
    }

/* start the output file as a copy of a fixture */
static int prefill(const char* script, const char* name, const char* out)
    {
    int rc = 0;
    if (g_rank == 0)
        {
        char src[2048];
        const char* slash = strrchr(script, '/');
        if (name[0] == '/')
            snprintf(src, sizeof(src), "%s", name);
        else if (slash)
            snprintf(src, sizeof(src), "%.*s/../files/%s", (int)(slash - script), script, name);
        else
            snprintf(src, sizeof(src), "../files/%s", name);
        FILE* in = fopen(src, "rb");
        FILE* o = in ? fopen(out, "wb") : NULL;
        if (!in || !o)
            {
            perror(in ? out : src);
            rc = 2;
            }
        else
            {
            char buf[65536];
            size_t n;
            while ((n = fread(buf, 1, sizeof(buf), in)) > 0)
                if (fwrite(buf, 1, n, o) != n)
                    rc = 2;
            }
        if (in)
            fclose(in);
        if (o && fclose(o) != 0)
            rc = 2;
        }
    all_ranks_meet();
    return rc;
    }

static enum pgsd_open_flag parse_flag(const char* s)
    {
    if (strcmp(s, "rw") == 0)
        return PGSD_OPEN_READWRITE;
    if (strcmp(s, "ro") == 0)
        return PGSD_OPEN_READONLY;
    if (strcmp(s, "append") == 0)
        return PGSD_OPEN_APPEND;
    fprintf(stderr, "bad flag %s\n", s);
    exit(2);
    }

/* rows of chunk writes that must outlive the call (batch 2: pgsd_set_deferred_rows): freed at the next
   end_frame / flush / close / dump, all of which resolve the queue */
static __thread int g_async = 0;   /* `async 1`: frames are sealed with pgsd_end_frame_async */
static __thread int g_batched = 0; /* pgsd_set_frame_exchange is on (batch 1, 2) */
static __thread int g_trusted = 0; /* batch 3 */
static __thread void** g_kept = NULL;
static __thread int g_nkept = 0;
static __thread int g_defer = 0;
#define KEPT_MAX 65536

static void release_rows(void* data)
    {
    if (g_defer && !g_kept)
        g_kept = (void**)malloc(KEPT_MAX * sizeof(void*));
    if (g_defer && g_nkept < KEPT_MAX)
        g_kept[g_nkept++] = data;
    else
        free(data);
    }

static void free_kept_rows(void)
    {
    for (int i = 0; i < g_nkept; i++)
        free(g_kept[i]);
    g_nkept = 0;
    }

/* ---- device rows (PGSD_DRIVER_DEVICE): what a chunk write passes instead of host rows ---- */
#ifdef PGSD_DRIVER_DEVICE
static __thread int g_device = 0; /* `device` command: 0 host rows, 1 dense device rows, 2 device rows inside wider rows */
static __thread void** g_dev_kept = NULL; /* device arrays a queued or asynchronous chunk may still read: freed with the frame */
static __thread int g_ndev_kept = 0;

static void free_device_rows(void)
    {
    for (int i = 0; i < g_ndev_kept; i++)
        (void)hipFree(g_dev_kept[i]); /* waits for the kernels that read it */
    g_ndev_kept = 0;
    }

/* rows[N][M] of `sz`-byte elements -> device memory; mode 2: row i sits at elements [i * stride + col0, + M) of a
   wider array whose other columns hold a pattern the chunk must not pick up */
static int upload_rows(const void* rows, uint64_t N, uint32_t M, size_t sz, int type, struct pgsd_field_desc* d)
    {
    memset(d, 0, sizeof(*d));
    d->src_type = (uint32_t)type;
    d->src_stride = M;
    if (N == 0 || M == 0 || !rows)
        return 0;
    uint32_t stride = M, col0 = 0;
    if (g_device == 2)
        {
        stride = M == 3 ? 4 : M + 2; /* xyz of a Scalar4; otherwise one foreign column on either side */
        col0 = M == 3 ? 0 : 1;
        }
    const size_t bytes = (size_t)N * stride * sz;
    char* host = (char*)rows;
    if (stride != M)
        {
        host = (char*)malloc(bytes);
        memset(host, 0xA5, bytes);
        for (uint64_t i = 0; i < N; i++)
            memcpy(host + (i * stride + col0) * sz, (const char*)rows + i * M * sz, (size_t)M * sz);
        }
    void* dev = NULL;
    int bad = hipMalloc(&dev, bytes) != hipSuccess || hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess;
    if (host != rows)
        free(host);
    if (bad)
        {
        fprintf(stderr, "rank %d: cannot upload %zu bytes to the device\n", g_rank, bytes);
        exit(4);
        }
    if (!g_dev_kept)
        g_dev_kept = (void**)malloc(KEPT_MAX * sizeof(void*));
    if (g_ndev_kept >= KEPT_MAX)
        exit(4);
    g_dev_kept[g_ndev_kept++] = dev;
    d->src = dev;
    d->src_stride = stride;
    d->src_col0 = col0;
    return 0;
    }
#endif

/* THE chunk write of every chunk command: pgsd_write_chunk (pgsd.h:551-564) -- or its device twin with the same
   arguments and the rows in HBM */
static int write_rows(struct pgsd_handle* h, const char* name, int type, uint64_t N, uint32_t M, uint64_t Ng, uint32_t Mg,
                      uint64_t off, uint64_t gs, int all, const void* data, size_t sz)
    {
#ifdef PGSD_DRIVER_DEVICE
    if (g_device)
        {
        struct pgsd_field_desc d;
        upload_rows(data, N, M, sz, type, &d);
        return pgsd_write_chunk_device(h, name, (enum pgsd_type)type, N, M, Ng, Mg, off, gs, all != 0, 0, &d);
        }
#endif
    (void)sz;
    return pgsd_write_chunk(h, name, (enum pgsd_type)type, N, M, Ng, Mg, off, gs, all != 0, 0, data);
    }

static int run_script(const char* script, const char* path);

#if !defined(PGSD_DRIVER_REF) && !defined(PGSD_DRIVER_MPI)
struct thread_rank
    {
    int rank, size, rc;
    const char *shm, *script, *path;
    };

static void* thread_rank_main(void* p)
    {
    struct thread_rank* t = (struct thread_rank*)p;
    g_rank = t->rank;
    g_size = t->size;
    if (pgsd_comm_create_shm(t->shm, t->rank, t->size, &g_comm) != PGSD_SUCCESS)
        {
        fprintf(stderr, "rank %d: pgsd_comm_create_shm failed: %s\n", t->rank, pgsd_last_error_string());
        t->rc = 3;
        return NULL;
        }
    g_own_comm = 1;
    t->rc = run_script(t->script, t->path);
    pgsd_comm_release(&g_comm);
    return NULL;
    }
#endif

int main(int argc, char** argv)
    {
    if (argc < 3)
        {
        fprintf(stderr, "usage: %s <script> <out.gsd>\n", argv[0]);
        return 2;
        }
#ifdef PGSD_DRIVER_REF
    MPI_Init(NULL, NULL);
    MPI_Comm_rank(MPI_COMM_WORLD, &g_rank);
    MPI_Comm_size(MPI_COMM_WORLD, &g_size);
#elif defined(PGSD_DRIVER_MPI)
    {
    /* PGSD_IO=mpiio: the library's writer threads call into MPI-IO too (serialised by the library) */
    int provided = 0;
    MPI_Init_thread(NULL, NULL, MPI_THREAD_SERIALIZED, &provided);
    }
    MPI_Comm_rank(MPI_COMM_WORLD, &g_rank);
    MPI_Comm_size(MPI_COMM_WORLD, &g_size);
    {
    struct pgsd_comm c;
    memset(&c, 0, sizeof(c));
    c.rank = g_rank;
    c.size = g_size;
    c.allgather = mpi_allgather_cb;
    c.barrier = mpi_barrier_cb;
    if (pgsd_comm_set_default(&c) != PGSD_SUCCESS)
        return 3;
    }
#else
    const char* nthreads = getenv("PGSD_DRIVER_THREADS");
    if (nthreads && atoi(nthreads) > 1)
        {
        /* the ranks are threads: one communicator, one handle and (device build) one pipeline each */
        const int P = atoi(nthreads);
        const char* shm = getenv("PGSD_SHM_NAME");
        struct thread_rank* t = (struct thread_rank*)calloc((size_t)P, sizeof(*t));
        pthread_t* th = (pthread_t*)calloc((size_t)P, sizeof(*th));
        if (!shm || !t || !th)
            return 3;
        for (int r = 0; r < P; r++)
            {
            t[r].rank = r, t[r].size = P, t[r].shm = shm, t[r].script = argv[1], t[r].path = argv[2];
            pthread_create(&th[r], NULL, thread_rank_main, &t[r]);
            }
        int worst = 0;
        for (int r = 0; r < P; r++)
            {
            pthread_join(th[r], NULL);
            if (t[r].rc > worst)
                worst = t[r].rc;
            }
        return worst;
        }
    if (pgsd_comm_init_from_env() != PGSD_SUCCESS)
        {
        fprintf(stderr, "pgsd_comm_init_from_env failed\n");
        return 3;
        }
    g_rank = pgsd_comm_rank();
    g_size = pgsd_comm_size();
#endif
    int rc_script = run_script(argv[1], argv[2]);
#ifdef PGSD_DRIVER_REF
    MPI_Finalize();
#elif defined(PGSD_DRIVER_MPI)
    pgsd_comm_finalize();
    MPI_Finalize();
#else
    pgsd_comm_finalize();
#endif
    return rc_script;
    }

static int run_script(const char* script, const char* path)
    {
    FILE* f = fopen(script, "r");
    if (!f)
        {
        perror(script);
        return 2;
        }

    struct pgsd_handle handle;
    uint64_t seed = 1;
    char line[1024];
    int lineno = 0;
    int rc_all = 0;
    while (fgets(line, sizeof(line), f))
        {
        lineno++;
        char* hash = strchr(line, '#');
        if (hash)
            *hash = 0;
        char* tok[16];
        int nt = 0;
        char* save = NULL;
        for (char* t = strtok_r(line, " \t\r\n", &save); t && nt < 16; t = strtok_r(NULL, " \t\r\n", &save))
            tok[nt++] = t;
        if (nt == 0)
            continue;
        int rc = 0;
        const char* cmd = tok[0];
        if (strcmp(cmd, "prefill") == 0 && nt == 2)
            {
            if (prefill(script, tok[1], path) != 0)
                return 2;
            }
        else if (strcmp(cmd, "create") == 0 && nt == 7)
            {
            g_batched = 0; /* a new handle starts with the per-chunk exchange */
#if !defined(PGSD_DRIVER_REF) && !defined(PGSD_DRIVER_MPI)
            if (g_own_comm)
                rc = pgsd_create_and_open_on(&g_comm, &handle, path, tok[1], tok[2],
                                             pgsd_make_version((unsigned)atoi(tok[3]), (unsigned)atoi(tok[4])),
                                             parse_flag(tok[5]), atoi(tok[6]));
            else
#endif
            rc = pgsd_create_and_open(&handle, path, tok[1], tok[2],
                                      pgsd_make_version((unsigned)atoi(tok[3]), (unsigned)atoi(tok[4])),
                                      parse_flag(tok[5]), atoi(tok[6]));
            }
        else if (strcmp(cmd, "open") == 0 && nt == 2)
            {
            g_batched = 0;
#if !defined(PGSD_DRIVER_REF) && !defined(PGSD_DRIVER_MPI)
            if (g_own_comm)
                rc = pgsd_open_on(&g_comm, &handle, path, parse_flag(tok[1]));
            else
#endif
            rc = pgsd_open(&handle, path, parse_flag(tok[1]));
            }
        else if (strcmp(cmd, "seed") == 0 && nt == 2)
            {
            seed = strtoull(tok[1], NULL, 10);
            }
        else if (strcmp(cmd, "chunk") == 0 && nt == 6)
            {
            size_t sz;
            int type = parse_type(tok[2], &sz);
            uint32_t M = (uint32_t)strtoul(tok[3], NULL, 10);
            int all = atoi(tok[4]);
            uint64_t counts[1024] = {0};
            dist_counts(tok[5], counts);
            uint64_t N = counts[g_rank], Ng = 0, row0 = 0;
            for (int r = 0; r < g_size; r++)
                {
                Ng += counts[r];
                if (r < g_rank)
                    row0 += counts[r];
                }
            void* data;
#ifndef PGSD_DRIVER_REF
            if (g_trusted)
                {
                /* a replicated chunk must have one size on all ranks for the declaration to cover it */
                int same = 1;
                for (int r = 1; r < g_size; r++)
                    same = same && counts[r] == counts[0];
                pgsd_set_partition(&handle, (all || same) ? counts : NULL, (uint32_t)g_size);
                }
            if (g_trusted && all)
                {
                data = gen_data(type, seed, row0, N, M, sz);
                rc = write_rows(&handle, tok[1], type, N, M, PGSD_PARTITION_AUTO, M, 0, 0, 1, data, sz);
                }
            else
#endif
            if (all)
                {
                data = gen_data(type, seed, row0, N, M, sz);
                rc = write_rows(&handle, tok[1], type, N, M, Ng, M, row0 * M, Ng * M, 1, data, sz);
                }
            else
                {
                data = gen_data(type, seed, 0, N, M, sz);
                rc = write_rows(&handle, tok[1], type, N, M, N, M, 0, N * M, 0, data, sz);
                }
            release_rows(data);
            }
        else if (strcmp(cmd, "chunkgs") == 0 && nt == 7)
            {
            size_t sz;
            int type = parse_type(tok[2], &sz);
            uint32_t M = (uint32_t)strtoul(tok[3], NULL, 10);
            int all = atoi(tok[4]);
            uint64_t gs = strtoull(tok[6], NULL, 10);
            uint64_t counts[1024] = {0};
            dist_counts(tok[5], counts);
            uint64_t N = counts[g_rank], Ng = 0, row0 = 0;
            for (int r = 0; r < g_size; r++)
                {
                Ng += counts[r];
                if (r < g_rank)
                    row0 += counts[r];
                }
            void* data = gen_data(type, seed, row0, N, M, sz);
#ifndef PGSD_DRIVER_REF
            if (g_trusted)
                pgsd_set_partition(&handle, NULL, 0);
#endif
            rc = write_rows(&handle, tok[1], type, N, M, Ng, M, row0 * M, gs, all, data, sz);
            release_rows(data);
            }
        else if ((strcmp(cmd, "rawchunk") == 0 || strcmp(cmd, "samechunk") == 0) && nt == 10)
            {
            size_t sz;
            int type = parse_type(tok[2], &sz);
            uint64_t N = strtoull(tok[3], NULL, 10);
            uint32_t M = (uint32_t)strtoul(tok[4], NULL, 10);
            uint64_t Ng = strtoull(tok[5], NULL, 10);
            uint32_t Mg = (uint32_t)strtoul(tok[6], NULL, 10);
            uint64_t off = strtoull(tok[7], NULL, 10);
            uint64_t gs = strtoull(tok[8], NULL, 10);
            int all = atoi(tok[9]);
            void* data = gen_data(type, cmd[0] == 's' ? seed : seed + (uint64_t)g_rank, 0, N, M, sz);
#ifndef PGSD_DRIVER_REF
            if (g_trusted)
                pgsd_set_partition(&handle, NULL, 0);
#endif
            rc = write_rows(&handle, tok[1], type, N, M, Ng, Mg, off, gs, all, data, sz);
            release_rows(data);
            }
        else if (strcmp(cmd, "end_frame") == 0)
            {
#ifndef PGSD_DRIVER_REF
            if (g_async)
                {
                rc = pgsd_end_frame_async(&handle);
#ifdef PGSD_DRIVER_DEVICE
                if (rc == 0)
                    rc = pgsd_device_wait_packed(&handle); /* the rows are freed below: the pack must have read them */
#endif
                }
            else
#endif
            rc = pgsd_end_frame(&handle);
            free_kept_rows();
#ifdef PGSD_DRIVER_DEVICE
            free_device_rows();
#endif
            }
        else if (strcmp(cmd, "flush") == 0)
            {
            rc = pgsd_flush(&handle);
            free_kept_rows();
#ifdef PGSD_DRIVER_DEVICE
            free_device_rows();
#endif
            }
        else if (strcmp(cmd, "close") == 0)
            {
            rc = pgsd_close(&handle);
            free_kept_rows();
#ifdef PGSD_DRIVER_DEVICE
            free_device_rows();
#endif
            }
        else if (strcmp(cmd, "maxbuf") == 0 && nt == 2)
            rc = pgsd_set_maximum_write_buffer_size(&handle, strtoull(tok[1], NULL, 10));
        else if (strcmp(cmd, "idxbuf") == 0 && nt == 2)
            rc = pgsd_set_index_entries_to_buffer(&handle, strtoull(tok[1], NULL, 10));
        else if (strcmp(cmd, "batch") == 0 && nt == 2)
            {
#ifndef PGSD_DRIVER_REF
            g_trusted = atoi(tok[1]) == 3;
            if (g_trusted)
                rc = 0;
            else
                {
                rc = pgsd_set_frame_exchange(&handle, atoi(tok[1]) != 0);
                g_batched = rc == 0 && atoi(tok[1]) != 0;
                }
            if (rc == 0 && atoi(tok[1]) == 2)
                {
                rc = pgsd_set_deferred_rows(&handle, 1);
                g_defer = 1;
                }
            else
                g_defer = 0;
#endif
            }
        else if (strcmp(cmd, "device") == 0 && nt == 2)
            {
#ifdef PGSD_DRIVER_DEVICE
            g_device = atoi(tok[1]);
            if (g_device && hipSetDevice(0) != hipSuccess)
                {
                fprintf(stderr, "rank %d: no HIP device\n", g_rank);
                return 4;
                }
#elif !defined(PGSD_DRIVER_REF)
            fprintf(stderr, "line %d: `device` needs the PGSD_DRIVER_DEVICE build\n", lineno);
            return 2;
#endif
            }
        else if (strcmp(cmd, "async") == 0 && nt == 2)
            g_async = atoi(tok[1]);
        else if (strcmp(cmd, "localreads") == 0 && nt == 2)
            {
#ifndef PGSD_DRIVER_REF
            rc = pgsd_set_local_reads(&handle, atoi(tok[1])); /* reads that take no part in a collective flush */
#endif
            }
        else if (strcmp(cmd, "dump") == 0)
            {
#ifndef PGSD_DRIVER_REF
            if (g_batched)
                {
                rc = pgsd_frame_exchange(&handle);
                free_kept_rows();
                }
#endif
            uint64_t nf = pgsd_get_nframes(&handle);
            uint64_t nn = pgsd_get_nnames(&handle);
            if (g_rank == 0)
                printf("dump line=%d nframes=%" PRIu64 " nnames=%" PRIu64 " file_size=%lld"
                       " index_location=%" PRIu64 " index_allocated=%" PRIu64
                       " namelist_location=%" PRIu64 " namelist_allocated=%" PRIu64 "\n",
                       lineno, nf, nn, (long long)handle.file_size, handle.header.index_location,
                       handle.header.index_allocated_entries, handle.header.namelist_location,
                       handle.header.namelist_allocated_entries);
            }
        else if (strcmp(cmd, "find") == 0 && nt == 3)
            {
            const struct pgsd_index_entry* e
                = pgsd_find_chunk(&handle, strtoull(tok[1], NULL, 10), tok[2]);
            if (g_rank == 0)
                {
                if (e)
                    printf("find line=%d frame=%s name=%s N=%" PRIu64 " M=%u type=%u location=%" PRId64
                           " id=%u\n",
                           lineno, tok[1], tok[2], e->N, e->M, (unsigned)e->type, e->location,
                           (unsigned)e->id);
                else
                    printf("find line=%d frame=%s name=%s NOTFOUND\n", lineno, tok[1], tok[2]);
                }
            }
        else if (strcmp(cmd, "read") == 0 && nt == 10)
            {
            uint64_t frame = strtoull(tok[1], NULL, 10);
            uint64_t N = strtoull(tok[3], NULL, 10);
            uint32_t M = (uint32_t)strtoul(tok[4], NULL, 10);
            uint32_t off = (uint32_t)strtoul(tok[5], NULL, 10);
            int all = atoi(tok[6]);
            size_t bytes = (size_t)strtoull(tok[7], NULL, 10) * strtoull(tok[8], NULL, 10) * strtoull(tok[9], NULL, 10);
            unsigned char* buf = (unsigned char*)calloc(bytes ? bytes : 1, 1);
            const struct pgsd_index_entry* e = pgsd_find_chunk(&handle, frame, tok[2]);
            int rrc = pgsd_read_chunk(&handle, buf, e, N, M, off, all != 0);
            if (g_rank == 0)
                {
                uint64_t h = 0xcbf29ce484222325ull;
                size_t got = 0;
                if (rrc == 0 && e)
                    got = all ? (size_t)(N * M) * pgsd_sizeof_type((enum pgsd_type)e->type)
                              : (size_t)(e->N * e->M) * pgsd_sizeof_type((enum pgsd_type)e->type);
                if (got > bytes)
                    got = bytes;
                for (size_t i = 0; i < got; i++)
                    {
                    h ^= buf[i];
                    h *= 0x100000001b3ull;
                    }
                printf("read line=%d frame=%s name=%s N=%s M=%s offset=%s all=%d rc=%d bytes=%zu fnv=%016" PRIx64 "\n",
                       lineno, tok[1], tok[2], tok[3], tok[4], tok[5], all, rrc, got, h);
                }
            free(buf);
            }
        else if (strcmp(cmd, "names") == 0 && nt <= 2)
            {
            const char* prefix = nt == 2 ? tok[1] : "";
            if (g_rank == 0)
                {
                const char* p = pgsd_find_matching_chunk_name(&handle, prefix, NULL);
                printf("names line=%d prefix=%s:", lineno, prefix);
                while (p)
                    {
                    printf(" %s", p);
                    p = pgsd_find_matching_chunk_name(&handle, prefix, p);
                    }
                printf("\n");
                }
            }
        else
            {
            fprintf(stderr, "line %d: bad command '%s' (%d tokens)\n", lineno, cmd, nt);
            return 2;
            }
        if (rc != 0)
            {
            if (g_rank == 0)
                printf("rc line=%d cmd=%s rc=%d\n", lineno, cmd, rc);
            rc_all = rc;
            }
        }
    fclose(f);
#ifdef PGSD_DRIVER_DEVICE
    free_device_rows();
#endif
    free_kept_rows();
    free(g_kept);
    g_kept = NULL;
    return rc_all ? 1 : 0;
    }
