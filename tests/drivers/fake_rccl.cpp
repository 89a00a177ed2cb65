// fake_rccl.cpp -- TEST DOUBLE of the seven librccl entry points the communicator uses (ncclGetUniqueId,
// ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString, ncclCommGetAsyncError, ncclCommAbort).
//
// Test infrastructure only; never loaded unless PGSD_RCCL_LIBRARY points at it.  Why it exists: the GPU boxes of
// this project have ONE GPU and RCCL refuses two ranks on one device, so pgsd_comm_rccl.cpp's multi-rank glue --
// who makes the unique id and how it travels, per-rank counts vs totals, the rank-ordered layout of the receive
// buffer, buffer growth, the barrier as a one-byte allgather, teardown order, and what happens when a rank does
// NOT come -- could never run at P > 1.  This library has RCCL's semantics for those calls: sendcount elements from
// every rank, gathered in rank order, and -- like the real thing -- ASYNCHRONOUS on the given HIP stream:
// ncclAllGather enqueues and returns, a one-lane kernel on the stream then waits for the peers (RCCL's kernels spin
// on their peers' flags in the same way), so a rank that never arrives leaves the stream blocked until
// ncclCommAbort tells the kernel to leave.  The transport is trivial: ranks that share the box meet in a POSIX
// shared-memory segment named after the unique id, device buffers are staged through pinned host memory by a
// helper thread per communicator.  What it does NOT show is RCCL's own transport over xGMI; that needs an 8-GPU
// node.
//
// PGSD_FAKE_RCCL_SYNC=1 -- for ranks that are THREADS of one process (eight ranks on a box that admits six GPU
// processes): ncclAllGather then does the whole exchange on the calling thread, synchronously, with no kernel.  Several
// ranks of one process cannot each park a waiting kernel on the one device: their streams share the process's few
// hardware queues, and a kernel waiting for a peer whose kernel sits behind it in the same queue waits for ever --
// the reason real RCCL wants one rank per device.  The asynchronous form is for ranks that are processes.
//
// Every wait is bounded (PGSD_FAKE_RCCL_MAX_WAIT_S, default 60 s: the waiting kernel leaves by itself then and the
// communicator reports ncclSystemError), so no wave outlives a test.  PGSD_FAKE_RCCL_STALL_RANK=k with
// PGSD_FAKE_RCCL_STALL_AT=n makes rank k "hang" inside its n-th allgather (counted from 0): it enqueues like the
// others but never meets them.
//
//   hipcc -O2 -fPIC -shared --offload-arch=gfx950 fake_rccl.cpp -o libpgsd_fake_rccl.so -lrt -pthread
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <mutex>
#include <new>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace
    {
const size_t SLOT = 1u << 20; // bytes one rank may contribute to one allgather

struct Shared
    {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> attached;
    uint32_t size;
    };

// words the waiting kernel and the host share (pinned, device-mapped)
struct Flags
    {
    uint32_t in_seq;  // kernel -> helper: the send bytes of exchange `seq` are in h_in
    uint32_t out_seq; // helper -> kernel: the gathered bytes of exchange `seq` are in h_out
    uint32_t abort;   // host -> kernel: leave
    uint32_t gave_up; // kernel -> host: left at its own deadline
    };

struct Job
    {
    uint32_t seq;
    size_t bytes;
    bool stall;
    };

struct FakeComm
    {
    int rank, size;
    Shared* sh;
    char* slots;
    size_t map_bytes;
    unsigned long long calls;
    // asynchronous machinery
    Flags* flags;  // pinned
    char* h_in;    // pinned, SLOT
    char* h_out;   // pinned, size * SLOT
    uint32_t seq;
    hipStream_t last_stream; // where the latest exchange was enqueued (what an abort waits for)
    double max_wait;
    int stall_at;  // this rank hangs inside its stall_at-th allgather (-1: never)
    bool sync;     // PGSD_FAKE_RCCL_SYNC: the exchange runs on the calling thread
    std::atomic<int> async_error;
    std::atomic<bool> quit;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> jobs;
    std::thread helper;
    };

double now()
    {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

uint32_t load32(const uint32_t* p)
    {
    return __atomic_load_n(p, __ATOMIC_ACQUIRE);
    }

// sense-reversing barrier over the segment; gives up at the deadline (a peer died or never came) or when told to
bool meet(FakeComm* c)
    {
    Shared* s = c->sh;
    const uint32_t gen = s->generation.load(std::memory_order_acquire);
    if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->size)
        {
        s->arrived.store(0, std::memory_order_relaxed);
        s->generation.store(gen + 1, std::memory_order_release);
        return true;
        }
    const double t0 = now();
    while (s->generation.load(std::memory_order_acquire) == gen)
        {
        sched_yield();
        if (now() - t0 > c->max_wait || (c->flags && load32(&c->flags->abort)) || c->quit.load())
            return false;
        }
    return true;
    }

// One lane on the caller's stream: publishes "my bytes are in h_in", then waits for the helper thread's "the
// gathered bytes are in h_out" -- or for the abort word, or for its own deadline (100 MHz constant clock).
__global__ void fake_rccl_wait_kernel(Flags* f, uint32_t seq, unsigned long long max_ticks)
    {
    __hip_atomic_store(&f->in_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    for (;;)
        {
        if ((int)(__hip_atomic_load(&f->out_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0)
            return;
        if (__hip_atomic_load(&f->abort, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM))
            return;
        if (wall_clock64() - t0 > max_ticks)
            {
            __hip_atomic_store(&f->gave_up, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
            }
        __builtin_amdgcn_s_sleep(32);
        }
    }

void helper_main(FakeComm* c)
    {
    for (;;)
        {
        Job job;
            {
            std::unique_lock<std::mutex> lk(c->m);
            c->cv.wait(lk, [&] { return c->quit.load() || !c->jobs.empty(); });
            if (c->jobs.empty())
                return;
            job = c->jobs.front();
            c->jobs.pop_front();
            }
        // the kernel says when the stream has reached this exchange and the send bytes are on the host
        const double t0 = now();
        while ((int)(load32(&c->flags->in_seq) - job.seq) < 0)
            {
            if (c->quit.load() || load32(&c->flags->abort) || now() - t0 > c->max_wait)
                break;
            sched_yield();
            }
        bool ok = (int)(load32(&c->flags->in_seq) - job.seq) >= 0 && !job.stall;
        if (ok)
            {
            memcpy(c->slots + (size_t)c->rank * SLOT, c->h_in, job.bytes);
            ok = meet(c);
            }
        if (ok)
            {
            for (int r = 0; r < c->size; r++) // rank order, sendcount elements each
                memcpy(c->h_out + (size_t)r * job.bytes, c->slots + (size_t)r * SLOT, job.bytes);
            ok = meet(c); // nobody overwrites a slot before everybody has read it
            }
        if (!ok)
            {
            // a hung rank: nothing is published, the kernel waits until it is aborted or gives up by itself
            if (!job.stall)
                c->async_error.store((int)ncclSystemError);
            continue;
            }
        __atomic_store_n(&c->flags->out_seq, job.seq, __ATOMIC_RELEASE);
        }
    }

void stop_helper(FakeComm* c)
    {
        {
        std::lock_guard<std::mutex> lk(c->m);
        c->quit.store(true);
        }
    c->cv.notify_all();
    if (c->helper.joinable())
        c->helper.join();
    }

void segment_name(const ncclUniqueId& id, char* out, size_t n)
    {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(id.internal); i++)
        h = (h ^ (unsigned char)id.internal[i]) * 1099511628211ull;
    snprintf(out, n, "/pgsd_fake_rccl_%016llx", h);
    }

void write_log(FakeComm* c, const char* how)
    {
    if (const char* log = getenv("PGSD_FAKE_RCCL_LOG")) // lets a test see who answered and how often
        {
        char path[1024];
        snprintf(path, sizeof(path), "%s.%d", log, c->rank);
        if (FILE* f = fopen(path, "w"))
            {
            fprintf(f, "{\"rank\": %d, \"size\": %d, \"allgathers\": %llu, \"end\": \"%s\"}\n", c->rank, c->size,
                    c->calls, how);
            fclose(f);
            }
        }
    }

void free_comm(FakeComm* c)
    {
    munmap((void*)c->sh, c->map_bytes);
    if (c->flags)
        (void)hipHostFree(c->flags);
    if (c->h_in)
        (void)hipHostFree(c->h_in);
    if (c->h_out)
        (void)hipHostFree(c->h_out);
    delete c;
    }
    } // namespace

extern "C" ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
    {
    memset(id, 0, sizeof(*id));
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 32) != 32)
        {
        if (fd >= 0)
            close(fd);
        return ncclSystemError;
        }
    close(fd);
    memcpy(id->internal + 32, "pgsd-fake-rccl", 14);
    return ncclSuccess;
    }

extern "C" ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
    {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks)
        return ncclInvalidArgument;
    char name[64];
    segment_name(id, name, sizeof(name));
    const size_t bytes = sizeof(Shared) + (size_t)nranks * SLOT;
    int fd = -1;
    if (rank == 0)
        {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0)
            return ncclSystemError;
        }
    else
        {
        const double t0 = now();
        struct stat st;
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < bytes)
            {
            if (fd >= 0)
                close(fd);
            if (now() - t0 > 60.0)
                return ncclSystemError;
            usleep(1000);
            }
        }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED)
        return ncclSystemError;
    FakeComm* c = new (std::nothrow) FakeComm;
    if (!c)
        return ncclSystemError;
    c->rank = rank;
    c->size = nranks;
    c->sh = (Shared*)p; // a fresh segment is zero-filled: counters start at 0
    c->slots = (char*)p + sizeof(Shared);
    c->map_bytes = bytes;
    c->calls = 0;
    c->flags = nullptr;
    c->h_in = c->h_out = nullptr;
    c->seq = 0;
    c->last_stream = nullptr;
    c->max_wait = 60.0;
    if (const char* w = getenv("PGSD_FAKE_RCCL_MAX_WAIT_S"))
        if (atof(w) > 0)
            c->max_wait = atof(w);
    c->stall_at = -1;
    const char* sr = getenv("PGSD_FAKE_RCCL_STALL_RANK");
    const char* sa = getenv("PGSD_FAKE_RCCL_STALL_AT");
    if (sr && *sr && atoi(sr) == rank)
        c->stall_at = sa ? atoi(sa) : 0;
    c->async_error.store((int)ncclSuccess);
    c->quit.store(false);
    const char* sy = getenv("PGSD_FAKE_RCCL_SYNC");
    c->sync = sy && atoi(sy) != 0;
    if (hipHostMalloc((void**)&c->flags, sizeof(Flags), hipHostMallocDefault) != hipSuccess
        || hipHostMalloc((void**)&c->h_in, SLOT, hipHostMallocDefault) != hipSuccess
        || hipHostMalloc((void**)&c->h_out, SLOT * (size_t)nranks, hipHostMallocDefault) != hipSuccess)
        {
        free_comm(c);
        return ncclUnhandledCudaError;
        }
    memset(c->flags, 0, sizeof(Flags));
    c->sh->attached.fetch_add(1);
    if (!meet(c))
        {
        free_comm(c);
        return ncclSystemError;
        }
    if (rank == 0)
        shm_unlink(name); // everybody is attached: the name can go
    if (!c->sync)
        c->helper = std::thread(helper_main, c);
    *comm = (ncclComm_t)c;
    return ncclSuccess;
    }

extern "C" ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
                                      ncclComm_t comm, hipStream_t stream)
    {
    FakeComm* c = (FakeComm*)comm;
    if (!c || !sendbuff || !recvbuff)
        return ncclInvalidArgument;
    size_t esz;
    switch (datatype)
        {
        case ncclInt8:
        case ncclUint8: esz = 1; break;
        case ncclInt32:
        case ncclUint32:
        case ncclFloat32: esz = 4; break;
        case ncclInt64:
        case ncclUint64:
        case ncclFloat64: esz = 8; break;
        default: return ncclInvalidArgument;
        }
    const size_t bytes = sendcount * esz;
    if (bytes > SLOT)
        return ncclInvalidArgument;
    if (c->sync)
        {
        // ranks as threads of one process: stream order by waiting for the stream, the exchange on this thread
        if (hipStreamSynchronize(stream) != hipSuccess
            || hipMemcpy(c->slots + (size_t)c->rank * SLOT, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return ncclUnhandledCudaError;
        if (!meet(c))
            return ncclSystemError;
        for (int r = 0; r < c->size; r++) // rank order, sendcount elements each
            memcpy(c->h_out + (size_t)r * bytes, c->slots + (size_t)r * SLOT, bytes);
        if (!meet(c)) // nobody overwrites a slot before everybody has read it
            return ncclSystemError;
        if (hipMemcpy(recvbuff, c->h_out, bytes * (size_t)c->size, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
        c->calls++;
        return ncclSuccess;
        }
    // everything in stream order, nothing waited for here: send bytes to the host, the kernel that waits for the
    // peers, the gathered bytes back into the receive buffer
    const uint32_t seq = ++c->seq;
    c->last_stream = stream;
    if (hipMemcpyAsync(c->h_in, sendbuff, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess)
        return ncclUnhandledCudaError;
    const unsigned long long ticks = (unsigned long long)(c->max_wait * 100e6);
    hipLaunchKernelGGL(fake_rccl_wait_kernel, dim3(1), dim3(1), 0, stream, c->flags, seq, ticks);
    if (hipGetLastError() != hipSuccess)
        return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recvbuff, c->h_out, bytes * (size_t)c->size, hipMemcpyHostToDevice, stream) != hipSuccess)
        return ncclUnhandledCudaError;
        {
        std::lock_guard<std::mutex> lk(c->m);
        c->jobs.push_back(Job {seq, bytes, c->stall_at >= 0 && c->calls == (unsigned long long)c->stall_at});
        }
    c->cv.notify_all();
    c->calls++;
    return ncclSuccess;
    }

extern "C" ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError)
    {
    FakeComm* c = (FakeComm*)comm;
    if (!c || !asyncError)
        return ncclInvalidArgument;
    *asyncError = load32(&c->flags->gave_up) ? ncclSystemError : (ncclResult_t)c->async_error.load();
    return ncclSuccess;
    }

// RCCL's abort: the kernels on the stream are told to leave, the communicator is freed without a word to the peers
extern "C" ncclResult_t ncclCommAbort(ncclComm_t comm)
    {
    FakeComm* c = (FakeComm*)comm;
    if (!c)
        return ncclInvalidArgument;
    __atomic_store_n(&c->flags->abort, 1u, __ATOMIC_RELEASE);
    stop_helper(c);
    if (c->seq)
        (void)hipStreamSynchronize(c->last_stream); // the waiting kernel has seen the word before the pinned memory goes
    write_log(c, "abort");
    free_comm(c);
    return ncclSuccess;
    }

extern "C" ncclResult_t ncclCommDestroy(ncclComm_t comm)
    {
    FakeComm* c = (FakeComm*)comm;
    if (!c)
        return ncclInvalidArgument;
    stop_helper(c);
    write_log(c, "destroy");
    free_comm(c);
    return ncclSuccess;
    }

extern "C" const char* ncclGetErrorString(ncclResult_t r)
    {
    switch (r)
        {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
        case ncclSystemError: return "fake rccl: peers did not meet";
        case ncclInvalidArgument: return "fake rccl: invalid argument";
        default: return "fake rccl: error";
        }
    }

// lets a test make sure which library answered
extern "C" int pgsd_fake_rccl_marker(void)
    {
    return 0x5047;
    }
