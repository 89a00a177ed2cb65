/* ref_bench.c -- time the REFERENCE's own CPU/MPI-IO write path on this host.
 *
 * Test/bench infrastructure (my code, not reference code): linked by oracle/Makefile against
 * the reference's pgsd.c compiled in place (oracle/_ref/ref_bench, git-ignored).  It does what
 * a CPU caller of the reference does per frame for bench.py's workload: pack position.xyz and
 * velocity.xyz out of float4 host arrays (plain C loop), then pgsd_write_chunk x3 with
 * all=true + one replicated step chunk + pgsd_end_frame (call pattern of
 * pgsd/scripts/benchmark-write.cc:85-106).  One MPI rank = one core.
 *
 *   mpiexec -n P ref_bench <particles_total> <frames> <out.gsd>
 * prints one JSON line on rank 0.
 */
#include "pgsd.h"

#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int main(int argc, char** argv)
    {
    MPI_Init(NULL, NULL);
    int rank, P;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    if (argc < 4)
        {
        if (rank == 0)
            fprintf(stderr, "usage: %s <particles_total> <frames> <out.gsd>\n", argv[0]);
        MPI_Finalize();
        return 2;
        }
    const uint64_t Ng = strtoull(argv[1], NULL, 10);
    const int frames = atoi(argv[2]);
    const char* path = argv[3];
    uint64_t n = Ng / P + ((uint64_t)rank < Ng % P ? 1 : 0);
    uint64_t row0 = 0;
    for (int r = 0; r < rank; r++)
        row0 += Ng / P + ((uint64_t)r < Ng % P ? 1 : 0);

    float* pos4 = (float*)malloc(n * 16);
    float* vel4 = (float*)malloc(n * 16);
    float* pos3 = (float*)malloc(n * 12);
    float* vel3 = (float*)malloc(n * 12);
    uint32_t* tid = (uint32_t*)malloc(n * 4);
    uint64_t s = 88172645463325252ull + (uint64_t)rank;
    for (uint64_t i = 0; i < 4 * n; i++)
        {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        pos4[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 100.0 - 50.0);
        vel4[i] = (float)((double)((s * 2685821657736338717ull) >> 11) / 9007199254740992.0 - 0.5);
        }
    for (uint64_t i = 0; i < n; i++)
        {
        uint32_t t = (uint32_t)((row0 + i) % 7);
        memcpy(&pos4[4 * i + 3], &t, 4); /* HOOMD keeps the type id in position.w */
        }

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
        for (uint64_t i = 0; i < n; i++)
            {
            pos3[3 * i] = pos4[4 * i]; pos3[3 * i + 1] = pos4[4 * i + 1]; pos3[3 * i + 2] = pos4[4 * i + 2];
            vel3[3 * i] = vel4[4 * i]; vel3[3 * i + 1] = vel4[4 * i + 1]; vel3[3 * i + 2] = vel4[4 * i + 2];
            memcpy(&tid[i], &pos4[4 * i + 3], 4);
            }
        uint64_t step = (uint64_t)(f + 1);
        pgsd_write_chunk(&h, "configuration/step", PGSD_TYPE_UINT64, 1, 1, 1, 1, 0, 1, false, 0, &step);
        pgsd_write_chunk(&h, "particles/position", PGSD_TYPE_FLOAT, n, 3, Ng, 3, row0 * 3, Ng * 3, true, 0, pos3);
        pgsd_write_chunk(&h, "particles/velocity", PGSD_TYPE_FLOAT, n, 3, Ng, 3, row0 * 3, Ng * 3, true, 0, vel3);
        pgsd_write_chunk(&h, "particles/typeid", PGSD_TYPE_UINT32, n, 1, Ng, 1, row0, Ng, true, 0, tid);
        pgsd_end_frame(&h);
        }
    MPI_Barrier(MPI_COMM_WORLD);
    double dt = MPI_Wtime() - t0, dtmax;
    MPI_Allreduce(&dt, &dtmax, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    pgsd_close(&h);
    if (rank == 0)
        {
        printf("{\"ranks\": %d, \"particles\": %llu, \"frames\": %d, \"seconds\": %.6f, \"GBps\": %.4f}\n", P,
               (unsigned long long)Ng, frames, dtmax, (double)frames * (double)Ng * 28.0 / dtmax / 1e9);
        unlink(path);
        }
    MPI_Finalize();
    return 0;
    }
