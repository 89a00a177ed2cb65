// fake_rccl.cpp -- TEST DOUBLE of the five librccl entry points the communicator uses (ncclGetUniqueId,
// ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString).
//
// Test infrastructure only; never loaded unless PGSD_RCCL_LIBRARY points at it.  Why it exists: the GPU boxes of
// this project have ONE GPU and RCCL refuses two ranks on one device, so pgsd_comm_rccl.cpp's multi-rank glue --
// who makes the unique id and how it travels, per-rank counts vs totals, the rank-ordered layout of the receive
// buffer, buffer growth, the barrier as a one-byte allgather, teardown order -- could never run at P > 1.  This
// library has RCCL's semantics for those calls (sendcount elements from every rank, gathered in rank order,
// ordered on the given HIP stream) with a trivial transport: ranks that share the box meet in a POSIX
// shared-memory segment named after the unique id, device buffers are staged through the host.  What it does
// NOT show is RCCL's own transport over xGMI; that needs an 8-GPU node.
//
//   hipcc -O2 -fPIC -shared --offload-arch=gfx950 fake_rccl.cpp -o libpgsd_fake_rccl.so -lrt
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <new>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
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

struct FakeComm
    {
    int rank, size;
    Shared* sh;
    char* slots;
    size_t map_bytes;
    unsigned long long calls;
    };

double now()
    {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

// sense-reversing barrier over the segment; gives up after 60 s (a peer died) instead of hanging the GPU box
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
        if (now() - t0 > 60.0)
            return false;
        }
    return true;
    }

void segment_name(const ncclUniqueId& id, char* out, size_t n)
    {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(id.internal); i++)
        h = (h ^ (unsigned char)id.internal[i]) * 1099511628211ull;
    snprintf(out, n, "/pgsd_fake_rccl_%016llx", h);
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
    c->sh->attached.fetch_add(1);
    if (!meet(c))
        return ncclSystemError;
    if (rank == 0)
        shm_unlink(name); // everybody is attached: the name can go
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
    // stream order: what was enqueued before the call is done before the buffers are touched ...
    if (hipStreamSynchronize(stream) != hipSuccess)
        return ncclUnhandledCudaError;
    if (hipMemcpy(c->slots + (size_t)c->rank * SLOT, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    if (!meet(c))
        return ncclSystemError;
    std::vector<char> all(bytes * (size_t)c->size);
    for (int r = 0; r < c->size; r++) // rank order, sendcount elements each
        memcpy(all.data() + (size_t)r * bytes, c->slots + (size_t)r * SLOT, bytes);
    if (!meet(c)) // nobody overwrites a slot before everybody has read it
        return ncclSystemError;
    // ... and the result is in place before anything enqueued after it runs (the copy is synchronous)
    if (hipMemcpy(recvbuff, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess)
        return ncclUnhandledCudaError;
    c->calls++;
    return ncclSuccess;
    }

extern "C" ncclResult_t ncclCommDestroy(ncclComm_t comm)
    {
    FakeComm* c = (FakeComm*)comm;
    if (!c)
        return ncclInvalidArgument;
    if (const char* log = getenv("PGSD_FAKE_RCCL_LOG")) // lets a test see who answered and how often
        {
        char path[1024];
        snprintf(path, sizeof(path), "%s.%d", log, c->rank);
        if (FILE* f = fopen(path, "w"))
            {
            fprintf(f, "{\"rank\": %d, \"size\": %d, \"allgathers\": %llu}\n", c->rank, c->size, c->calls);
            fclose(f);
            }
        }
    munmap((void*)c->sh, c->map_bytes);
    delete c;
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
