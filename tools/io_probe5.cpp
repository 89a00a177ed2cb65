// io_probe5: what a disk-backed target of the GPU box takes -- O_DIRECT vs buffered(+fdatasync),
// 16 MiB aligned pieces, T writer threads on disjoint ranges of ONE file.
// build: g++ -O2 -pthread tools/io_probe5.cpp -o /tmp/io_probe5 ; run: /tmp/io_probe5 <dir>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

static double now()
    {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

int main(int argc, char** argv)
    {
    std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t piece = (size_t)16 << 20, total = (size_t)2800 << 20;
    for (int direct = 1; direct >= 0; direct--)
        for (int T : {1, 2, 4, 8})
            {
            std::string path = dir + "/io_probe5.bin";
            int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | (direct ? O_DIRECT : 0), 0644);
            if (fd < 0)
                {
                printf("open(%s, direct=%d): %s\n", path.c_str(), direct, strerror(errno));
                continue;
                }
            std::vector<void*> bufs(T);
            for (auto& b : bufs)
                {
                if (posix_memalign(&b, 4096, piece))
                    return 1;
                memset(b, 0x5a, piece);
                }
            const size_t n_pieces = total / piece;
            double t0 = now();
            std::vector<std::thread> th;
            bool ok = true;
            for (int t = 0; t < T; t++)
                th.emplace_back(
                    [&, t]
                    {
                        for (size_t i = t; i < n_pieces; i += T)
                            if (pwrite(fd, bufs[t], piece, (off_t)(i * piece)) != (ssize_t)piece)
                                ok = false;
                    });
            for (auto& x : th)
                x.join();
            double t1 = now();
            fdatasync(fd);
            double t2 = now();
            printf("%s T=%d  write %.2f GB/s  (+fdatasync %.2f s => %.2f GB/s durable)%s\n", direct ? "O_DIRECT" : "buffered",
                   T, total / (t1 - t0) / 1e9, t2 - t1, total / (t2 - t0) / 1e9, ok ? "" : "  WRITE FAILED");
            fflush(stdout);
            close(fd);
            unlink(path.c_str());
            for (auto b : bufs)
                free(b);
            }
    return 0;
    }
