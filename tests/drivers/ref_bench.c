/* ref_bench.c -- time the REFERENCE's own CPU/MPI-IO write path on this host.
 *
 * Test/bench infrastructure (my code, not reference code): linked by oracle/Makefile against
 * the reference's pgsd.c compiled in place (oracle/_ref/ref_bench, git-ignored).  It does what
 * a CPU caller of the reference does per frame for bench.py's workloads: pack every chunk's columns out of
 * HOOMD-style host arrays (Scalar4 / int4 / scalar arrays; plain C loop), then one pgsd_write_chunk with
 * all=true per chunk + one replicated step chunk + pgsd_end_frame (call pattern of
 * pgsd/scripts/benchmark-write.cc:85-106).  One MPI rank = one core.
 *
 *   mpiexec -n P ref_bench <particles_total> <frames> <out.gsd> [pvi|sph|union]
 *     pvi    position + velocity + typeid, 28 B/particle (bench.py's headline)
 *     sph    the 14 per-particle chunks of the PGSD-SPH schema, 112 B/particle (BASELINE config 4)
 *     union  ... plus the upstream HOOMD attributes, 164 B/particle
 * prints one JSON line on rank 0.
 */
#include "pgsd.h"

#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* source arrays (all 32-bit elements): index, elements per row */
enum { A_POS, A_VEL, A_DPE, A_AUX1, A_AUX2, A_AUX3, A_AUX4, A_IMG, A_BODY, A_CHARGE, A_DIAM, A_INERTIA, A_ORIENT, A_ANGMOM, A_COUNT };
static const int stride_of[A_COUNT] = {4, 4, 4, 4, 4, 4, 4, 4, 1, 1, 1, 3, 4, 4};

struct chunk_def
    {
    const char* name;
    enum pgsd_type type;
    int M, arr, col0;
    };

static const struct chunk_def PVI[] = {{"particles/position", PGSD_TYPE_FLOAT, 3, A_POS, 0},
                                       {"particles/velocity", PGSD_TYPE_FLOAT, 3, A_VEL, 0},
                                       {"particles/typeid", PGSD_TYPE_UINT32, 1, A_POS, 3}};
static const struct chunk_def UNION[] = {{"particles/typeid", PGSD_TYPE_UINT32, 1, A_POS, 3},
                                         {"particles/mass", PGSD_TYPE_FLOAT, 1, A_VEL, 3},
                                         {"particles/body", PGSD_TYPE_INT32, 1, A_BODY, 0},
                                         {"particles/position", PGSD_TYPE_FLOAT, 3, A_POS, 0},
                                         {"particles/velocity", PGSD_TYPE_FLOAT, 3, A_VEL, 0},
                                         {"particles/slength", PGSD_TYPE_FLOAT, 1, A_DPE, 3},
                                         {"particles/density", PGSD_TYPE_FLOAT, 1, A_DPE, 0},
                                         {"particles/pressure", PGSD_TYPE_FLOAT, 1, A_DPE, 1},
                                         {"particles/energy", PGSD_TYPE_FLOAT, 1, A_DPE, 2},
                                         {"particles/auxiliary1", PGSD_TYPE_FLOAT, 3, A_AUX1, 0},
                                         {"particles/auxiliary2", PGSD_TYPE_FLOAT, 3, A_AUX2, 0},
                                         {"particles/auxiliary3", PGSD_TYPE_FLOAT, 3, A_AUX3, 0},
                                         {"particles/auxiliary4", PGSD_TYPE_FLOAT, 3, A_AUX4, 0},
                                         {"particles/image", PGSD_TYPE_INT32, 3, A_IMG, 0},
                                         /* the SPH set ends here (14 chunks); the union continues */
                                         {"particles/charge", PGSD_TYPE_FLOAT, 1, A_CHARGE, 0},
                                         {"particles/diameter", PGSD_TYPE_FLOAT, 1, A_DIAM, 0},
                                         {"particles/moment_inertia", PGSD_TYPE_FLOAT, 3, A_INERTIA, 0},
                                         {"particles/orientation", PGSD_TYPE_FLOAT, 4, A_ORIENT, 0},
                                         {"particles/angmom", PGSD_TYPE_FLOAT, 4, A_ANGMOM, 0}};

int main(int argc, char** argv)
    {
    MPI_Init(NULL, NULL);
    int rank, P;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    if (argc < 4)
        {
        if (rank == 0)
            fprintf(stderr, "usage: %s <particles_total> <frames> <out.gsd> [pvi|sph|union]\n", argv[0]);
        MPI_Finalize();
        return 2;
        }
    const uint64_t Ng = strtoull(argv[1], NULL, 10);
    const int frames = atoi(argv[2]);
    const char* path = argv[3];
    const char* schema = argc > 4 ? argv[4] : "pvi";
    const struct chunk_def* chunks = PVI;
    int n_chunks = 3;
    if (strcmp(schema, "sph") == 0)
        chunks = UNION, n_chunks = 14;
    else if (strcmp(schema, "union") == 0)
        chunks = UNION, n_chunks = 19;
    else if (strcmp(schema, "pvi") != 0)
        {
        if (rank == 0)
            fprintf(stderr, "unknown schema %s\n", schema);
        MPI_Finalize();
        return 2;
        }
    uint64_t n = Ng / P + ((uint64_t)rank < Ng % P ? 1 : 0);
    uint64_t row0 = 0;
    for (int r = 0; r < rank; r++)
        row0 += Ng / P + ((uint64_t)r < Ng % P ? 1 : 0);

    /* the source arrays this schema reads, filled with pseudo-random words; one dense buffer per chunk */
    uint32_t* src[A_COUNT] = {0};
    uint32_t* dst[19] = {0};
    uint64_t payload = 0;
    uint64_t s = 88172645463325252ull + (uint64_t)rank;
    for (int c = 0; c < n_chunks; c++)
        {
        const int a = chunks[c].arr;
        if (!src[a])
            {
            src[a] = (uint32_t*)malloc(n * stride_of[a] * 4 + 16);
            float* f = (float*)src[a];
            for (uint64_t i = 0; i < n * stride_of[a]; i++)
                {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                if (a == A_IMG || a == A_BODY)
                    src[a][i] = (uint32_t)((int32_t)(s % 5) - 2);
                else
                    f[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 100.0 - 50.0);
                }
            }
        dst[c] = (uint32_t*)malloc(n * chunks[c].M * 4 + 16);
        payload += (uint64_t)chunks[c].M * 4;
        }
    for (uint64_t i = 0; i < n; i++)
        src[A_POS][4 * i + 3] = (uint32_t)((row0 + i) % 7); /* HOOMD keeps the type id in position.w */

    struct pgsd_handle h;
    int rc = pgsd_create_and_open(&h, path, "ref_bench", "hoomd", pgsd_make_version(1, 4), PGSD_OPEN_READWRITE, 0);
    if (rc != 0)
        {
        fprintf(stderr, "create failed %d\n", rc);
        MPI_Abort(MPI_COMM_WORLD, 1);
        }
    double t0 = 0;
    for (int f = -1; f < frames; f++)
        {
        if (f == 0)
            {
            MPI_Barrier(MPI_COMM_WORLD);
            t0 = MPI_Wtime();
            }
        for (int c = 0; c < n_chunks; c++)
            {
            const uint32_t* a = src[chunks[c].arr] + chunks[c].col0;
            const int st = stride_of[chunks[c].arr], M = chunks[c].M;
            uint32_t* d = dst[c];
            if (M == 3)
                for (uint64_t i = 0; i < n; i++)
                    {
                    d[3 * i] = a[st * i]; d[3 * i + 1] = a[st * i + 1]; d[3 * i + 2] = a[st * i + 2];
                    }
            else if (M == 1)
                for (uint64_t i = 0; i < n; i++)
                    d[i] = a[st * i];
            else
                for (uint64_t i = 0; i < n; i++)
                    for (int k = 0; k < M; k++)
                        d[(uint64_t)M * i + k] = a[st * i + k];
            }
        uint64_t step = (uint64_t)(f + 1);
        pgsd_write_chunk(&h, "configuration/step", PGSD_TYPE_UINT64, 1, 1, 1, 1, 0, 1, false, 0, &step);
        for (int c = 0; c < n_chunks; c++)
            pgsd_write_chunk(&h, chunks[c].name, chunks[c].type, n, chunks[c].M, Ng, chunks[c].M, row0 * chunks[c].M,
                             Ng * chunks[c].M, true, 0, dst[c]);
        pgsd_end_frame(&h);
        }
    MPI_Barrier(MPI_COMM_WORLD);
    double dt = MPI_Wtime() - t0, dtmax;
    MPI_Allreduce(&dt, &dtmax, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    pgsd_close(&h);
    if (rank == 0)
        {
        printf("{\"ranks\": %d, \"particles\": %llu, \"frames\": %d, \"schema\": \"%s\", \"payload_bytes_per_particle\": %llu, "
               "\"seconds\": %.6f, \"GBps\": %.4f}\n", P, (unsigned long long)Ng, frames, schema,
               (unsigned long long)payload, dtmax, (double)frames * (double)Ng * (double)payload / dtmax / 1e9);
        unlink(path);
        }
    MPI_Finalize();
    return 0;
    }
