// io_probe6: can bytes reach ONE tmpfs file faster than a pwrite thread inserts them
// (5.5-7 GB/s)?  Pipeline over 16 MiB extents of one long-lived MAP_SHARED mapping:
//   allocator thread: fallocate(fd, 0, off, len)          (allocates pages, extends the file)
//   R register threads: hipHostRegister(map + off, len)    (pins the page-cache pages for the GPU)
//   copy: hipMemcpyAsync(map + off, dev, len, D2H)         (DMA straight into the file's pages)
//   unregister threads: hipHostUnregister
// build: hipcc -O2 --offload-arch=gfx950 tools/io_probe6.cpp -o /tmp/io_probe6 -pthread
// run:   /tmp/io_probe6 <dir> [register threads] [extent MiB] [total MiB]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now()
    {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

template<class T> struct Queue
    {
    std::deque<T> q;
    std::mutex m;
    std::condition_variable cv;
    bool closed = false;
    void push(T v)
        {
            {
            std::lock_guard<std::mutex> g(m);
            q.push_back(v);
            }
        cv.notify_one();
        }
    bool pop(T& v)
        {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return closed || !q.empty(); });
        if (q.empty())
            return false;
        v = q.front();
        q.pop_front();
        return true;
        }
    void close()
        {
            {
            std::lock_guard<std::mutex> g(m);
            closed = true;
            }
        cv.notify_all();
        }
    };

int main(int argc, char** argv)
    {
    std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const int R = argc > 2 ? atoi(argv[2]) : 4;
    const size_t ext = (size_t)(argc > 3 ? atoi(argv[3]) : 16) << 20;
    const size_t total = (size_t)(argc > 4 ? atoi(argv[4]) : 2800) << 20;
    const size_t n_ext = total / ext;
    const size_t head = 5376; // data starts unaligned in a real file; extents here stay page-aligned behind it
    std::string path = dir + "/io_probe6.bin";

    char* dev = nullptr;
    if (hipMalloc((void**)&dev, ext * 4) != hipSuccess)
        return 1;
    hipMemset(dev, 0x5a, ext * 4);
    hipStream_t stream;
    hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);

    for (int rep = 0; rep < 3; rep++)
        {
        int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0)
            return 2;
        const size_t map_len = (size_t)1 << 40;
        char* map = (char*)mmap(NULL, map_len, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
        if (map == MAP_FAILED)
            {
            perror("mmap");
            return 3;
            }
        (void)head;
        Queue<size_t> q_alloc, q_reg, q_copy, q_unreg;
        std::atomic<size_t> done{0};
        std::atomic<bool> failed{false};
        double t_alloc = 0, t_reg = 0, t_unreg = 0;
        std::mutex tm;
        double t0 = now();
        std::thread allocator(
            [&]
            {
                size_t k;
                while (q_alloc.pop(k))
                    {
                    double a = now();
                    if (fallocate(fd, 0, (off_t)(k * ext), (off_t)ext) != 0)
                        failed = true;
                    t_alloc += now() - a;
                    q_reg.push(k);
                    }
                q_reg.close();
            });
        std::vector<std::thread> regs;
        std::atomic<int> regs_left{R};
        for (int r = 0; r < R; r++)
            regs.emplace_back(
                [&]
                {
                    size_t k;
                    while (q_reg.pop(k))
                        {
                        double a = now();
                        if (hipHostRegister(map + k * ext, ext, hipHostRegisterDefault) != hipSuccess)
                            failed = true;
                        double d = now() - a;
                            {
                            std::lock_guard<std::mutex> g(tm);
                            t_reg += d;
                            }
                        q_copy.push(k);
                        }
                    if (--regs_left == 0)
                        q_copy.close();
                });
        std::thread copier(
            [&]
            {
                size_t k;
                std::deque<std::pair<size_t, hipEvent_t>> inflight;
                auto retire = [&](bool all)
                {
                    while (!inflight.empty() && (all || inflight.size() >= 8))
                        {
                        hipEventSynchronize(inflight.front().second);
                        hipEventDestroy(inflight.front().second);
                        q_unreg.push(inflight.front().first);
                        inflight.pop_front();
                        }
                };
                while (q_copy.pop(k))
                    {
                    hipEvent_t ev;
                    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                    if (hipMemcpyAsync(map + k * ext, dev + (k % 4) * ext, ext, hipMemcpyDeviceToHost, stream) != hipSuccess)
                        failed = true;
                    hipEventRecord(ev, stream);
                    inflight.push_back({k, ev});
                    retire(false);
                    }
                retire(true);
                q_unreg.close();
            });
        std::vector<std::thread> unregs;
        for (int r = 0; r < 2; r++)
            unregs.emplace_back(
                [&]
                {
                    size_t k;
                    while (q_unreg.pop(k))
                        {
                        double a = now();
                        if (hipHostUnregister(map + k * ext) != hipSuccess)
                            failed = true;
                        double d = now() - a;
                            {
                            std::lock_guard<std::mutex> g(tm);
                            t_unreg += d;
                            }
                        done++;
                        }
                });
        for (size_t k = 0; k < n_ext; k++)
            q_alloc.push(k);
        q_alloc.close();
        allocator.join();
        for (auto& t : regs)
            t.join();
        copier.join();
        for (auto& t : unregs)
            t.join();
        double dt = now() - t0;
        // verify a few bytes through the file descriptor (page cache coherence)
        unsigned char probe[4] = {0, 0, 0, 0};
        pread(fd, probe, 4, (off_t)(total - 4));
        printf("R=%d ext=%zu MiB: %.2f GB/s  (thread-seconds: fallocate %.3f, register %.3f, unregister %.3f; wall %.3f)%s last bytes %02x%02x\n",
               R, ext >> 20, total / dt / 1e9, t_alloc, t_reg, t_unreg, dt, failed ? " FAILED" : "", probe[0], probe[3]);
        fflush(stdout);
        munmap(map, map_len);
        close(fd);
        unlink(path.c_str());
        }
    return 0;
    }
