// tools/small_frames_c.hip -- per-frame cost of small frames through the C ABI only (no Python): host rows vs device rows
// (dense N x 3 + N x 4 f32 chunks, batched exchange, deferred rows).  hipcc -O2 --offload-arch=gfx950 -Iinclude tools/small_frames_c.hip
//   -o tools/build/small_frames_c -Lpgsd-sph_amd/pgsd -lpgsd_amd -Wl,-rpath,$PWD/pgsd-sph_amd/pgsd ; tools/build/small_frames_c <N> <frames>
#include "pgsd.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
    {
    const uint64_t N = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    const int frames = argc > 2 ? atoi(argv[2]) : 2000;
    std::vector<float> hp(N * 3, 1.f), ho(N * 4, 2.f);
    float *dp, *dq;
    hipMalloc((void**)&dp, N * 12); hipMalloc((void**)&dq, N * 16);
    hipMemcpy(dp, hp.data(), N * 12, hipMemcpyHostToDevice); hipMemcpy(dq, ho.data(), N * 16, hipMemcpyHostToDevice);
    for (int device = 0; device < 2; device++)
        for (int rep = 0; rep < 2; rep++)
            {
            pgsd_handle h;
            if (pgsd_create_and_open(&h, "/dev/shm/pgsd_small_c.gsd", "a", "hoomd", pgsd_make_version(1, 4), PGSD_OPEN_READWRITE, 0)) return 1;
            pgsd_set_frame_exchange(&h, 1);
            pgsd_set_deferred_rows(&h, 1);
            double t0 = 0;
            for (int f = -50; f < frames; f++)
                {
                if (f == 0) t0 = now();
                uint64_t step = (uint64_t)(f + 50);
                pgsd_write_chunk(&h, "configuration/step", PGSD_TYPE_UINT64, 1, 1, 1, 1, 0, 1, false, 0, &step);
                if (device)
                    {
                    pgsd_chunk_req r[2] = {{"particles/position", PGSD_TYPE_FLOAT, 3, {dp, nullptr, PGSD_TYPE_FLOAT, 3, 0, 0}},
                                           {"particles/orientation", PGSD_TYPE_FLOAT, 4, {dq, nullptr, PGSD_TYPE_FLOAT, 4, 0, 0}}};
                    if (pgsd_write_chunks_device(&h, 2, r, N, N, 0)) return 2;
                    }
                else
                    {
                    pgsd_write_chunk(&h, "particles/position", PGSD_TYPE_FLOAT, N, 3, N, 3, 0, N * 3, true, 0, hp.data());
                    pgsd_write_chunk(&h, "particles/orientation", PGSD_TYPE_FLOAT, N, 4, N, 4, 0, N * 4, true, 0, ho.data());
                    }
                if (pgsd_end_frame(&h)) return 3;
                }
            double dt = now() - t0;
            pgsd_close(&h);
            unlink("/dev/shm/pgsd_small_c.gsd");
            printf("%s rows, N=%llu: %.1f us/frame\n", device ? "device" : "host", (unsigned long long)N, dt / frames * 1e6);
            }
    return 0;
    }
