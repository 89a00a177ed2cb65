// pgsd_comm_rccl.cpp -- RCCL (xGMI) back end of the communicator.
//
// The per-frame exchange of the write path -- every rank's local row count, from which
// each rank derives its file offsets (MPI_Allgather in the reference's callers,
// benchmark-write.cc:41, and in pgsd.c:1126) -- runs as ONE ncclAllGather over the xGMI
// mesh of the node, on a private HIP stream, between two small device buffers; the
// host reads the P gathered values back from pinned memory.  With the frame exchange batched
// (pgsd_set_frame_exchange) it is also the only collective of a frame, and it is issued after the
// frame's pack kernel has been enqueued on the pipeline's own stream: the two run side by side, and
// the host waits for the gathered sizes only where the file offsets are needed (before the copies
// are handed to the writer).  Messages are a few bytes,
// so this is latency-bound; what matters is that it is a single device collective that
// can be ordered against kernels (e.g. a count produced by pgsd_select_rows) without a
// host round trip through MPI.
//
// librccl is opened at run time (dlopen) so that libpgsd_amd.so has no hard dependency on
// it: inside a PyTorch process the already-loaded RCCL (backend "nccl") is reused.
//
// Failure handling.  A peer that never enters the collective leaves RCCL's kernel spinning on the stream for good
// (MPI_Allgather in the reference waits for ever as well, pgsd.c:1126); here the wait is BOUNDED: the host polls an
// event recorded behind the device->host copy (hipEventQuery) against a deadline -- PGSD_COMM_TIMEOUT_S seconds,
// default 120 -- and polls ncclCommGetAsyncError beside it.  On either the communicator is aborted
// (ncclCommAbort: RCCL's kernels leave the stream), the call returns -1 (PGSD_ERROR_COMM at the C ABI) and the
// context stays broken: every later collective on it fails at once.  The process is never re-executed.
#include "pgsd_internal.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <sched.h>
#include <unistd.h>

namespace pgsd_amd
    {
struct RcclApi
    {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr; // optional: polled while waiting
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                        // optional: what ends a wait that timed out
    };

static RcclApi g_rccl;

// guarded: reachable from pgsd_comm_rccl_available and from several thread-ranks of one process at once
static std::mutex g_rccl_lock;

static bool load_rccl()
    {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    if (g_rccl.lib)
        return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    // PGSD_RCCL_LIBRARY: a particular RCCL build (or the tests' stand-in, tests/drivers/fake_rccl.cpp); its symbols
    // stay local, everything is reached through dlsym on the handle
    const char* chosen = getenv("PGSD_RCCL_LIBRARY");
    if (chosen && *chosen)
        {
        lib = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
        if (!lib)
            {
            set_last_error(std::string("cannot load PGSD_RCCL_LIBRARY: ") + dlerror());
            return false;
            }
        }
    for (const char* n : names)
        {
        if (lib)
            break;
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
    if (!lib)
        {
        set_last_error(std::string("cannot load librccl: ") + dlerror());
        return false;
        }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(lib, "ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    g_rccl.CommGetAsyncError = (decltype(g_rccl.CommGetAsyncError))dlsym(lib, "ncclCommGetAsyncError");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(lib, "ncclCommAbort");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather)
        {
        set_last_error("librccl lacks the expected nccl* symbols");
        return false;
        }
    g_rccl.lib = lib;
    return true;
    }

struct RcclCtx
    {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int device = 0, rank = 0, size = 1;
    char* d_send = nullptr; // device
    char* d_recv = nullptr; // device
    char* h_send = nullptr; // pinned
    char* h_recv = nullptr; // pinned
    size_t cap = 0;         // bytes per rank the buffers hold
    hipEvent_t done = nullptr; // recorded behind the device->host copy of an exchange: what the host polls
    double timeout_s = 120.0;  // PGSD_COMM_TIMEOUT_S
    bool broken = false;       // a collective timed out / failed: aborted, every later call fails at once
    bool stuck = false;        // the stream did not drain after the abort: its resources are left alone
    std::string why;           // the first failure, repeated by the later calls
    };

static const char* nccl_text(ncclResult_t r)
    {
    return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error";
    }

static double seconds_since(std::chrono::steady_clock::time_point t0)
    {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

// The communicator is given up: RCCL's kernels are told to leave (ncclCommAbort also frees the communicator),
// the stream gets a few seconds to drain what is left on it.
static int rccl_break(RcclCtx* c, const std::string& why)
    {
    c->broken = true;
    c->why = why;
    if (c->comm && g_rccl.CommAbort)
        (void)g_rccl.CommAbort(c->comm);
    if (c->comm && !g_rccl.CommAbort)
        c->stuck = true; // nothing can make the kernel leave: neither the communicator nor the stream is touched again
    c->comm = g_rccl.CommAbort ? nullptr : c->comm;
    if (!c->stuck && c->done)
        {
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipEventQuery(c->done)) == hipErrorNotReady && seconds_since(t0) < 5.0)
            usleep(1000);
        (void)hipGetLastError();
        if (q != hipSuccess)
            c->stuck = true;
        }
    set_last_error(why);
    return -1;
    }

// Wait for the exchange enqueued on the private stream, but not for ever: poll the event behind the last copy
// (spinning at first -- an exchange among ranks that are all there takes ~17 us --, then yielding, then sleeping),
// the communicator's asynchronous error beside it, and give up at the deadline.
static int rccl_wait_bounded(RcclCtx* c)
    {
    const auto t0 = std::chrono::steady_clock::now();
    double next_async_check = 1e-3;
    for (;;)
        {
        const hipError_t q = hipEventQuery(c->done);
        if (q == hipSuccess)
            return 0;
        (void)hipGetLastError(); // "not ready" is the answer, not an error a later hipGetLastError() of this thread should see
        if (q != hipErrorNotReady)
            return rccl_break(c, std::string("RCCL exchange: the stream reports ") + hipGetErrorString(q));
        const double el = seconds_since(t0);
        if (el >= next_async_check)
            {
            next_async_check = el < 0.05 ? el + 1e-3 : el + 0.05;
            ncclResult_t state = ncclSuccess;
            if (g_rccl.CommGetAsyncError && g_rccl.CommGetAsyncError(c->comm, &state) == ncclSuccess && state != ncclSuccess
                && state != ncclInProgress)
                return rccl_break(c, std::string("RCCL exchange failed asynchronously: ") + nccl_text(state));
            if (c->timeout_s > 0 && el > c->timeout_s)
                {
                char msg[200];
                snprintf(msg, sizeof(msg),
                         "RCCL exchange timed out after %.1f s (PGSD_COMM_TIMEOUT_S): a rank did not arrive; "
                         "communicator aborted", el);
                return rccl_break(c, msg);
                }
            }
        if (el < 200e-6)
            __builtin_ia32_pause();
        else if (el < 5e-3)
            sched_yield();
        else
            usleep(200);
        }
    }

static bool rccl_reserve(RcclCtx* c, size_t bytes)
    {
    if (bytes <= c->cap)
        return true;
    size_t cap = bytes < 256 ? 256 : bytes;
    // the old buffers go first and the context forgets them at once: a failed allocation below must not
    // leave a capacity that points at freed memory
    if (c->d_send)
        (void)hipFree(c->d_send);
    if (c->d_recv)
        (void)hipFree(c->d_recv);
    if (c->h_send)
        (void)hipHostFree(c->h_send);
    if (c->h_recv)
        (void)hipHostFree(c->h_recv);
    c->d_send = c->d_recv = c->h_send = c->h_recv = nullptr;
    c->cap = 0;
    if (hipMalloc((void**)&c->d_send, cap) != hipSuccess || hipMalloc((void**)&c->d_recv, cap * (size_t)c->size) != hipSuccess
        || hipHostMalloc((void**)&c->h_send, cap, hipHostMallocDefault) != hipSuccess
        || hipHostMalloc((void**)&c->h_recv, cap * (size_t)c->size, hipHostMallocDefault) != hipSuccess)
        {
        set_last_error("RCCL communicator: cannot allocate the exchange buffers");
        return false;
        }
    c->cap = cap;
    return true;
    }

static int rccl_allgather(void* p, const void* send, void* recv, size_t bytes)
    {
    RcclCtx* c = (RcclCtx*)p;
    if (c->broken)
        {
        set_last_error("the RCCL communicator is broken: " + c->why);
        return -1;
        }
    if (bytes == 0)
        return 0;
    if (hipSetDevice(c->device) != hipSuccess || !rccl_reserve(c, bytes))
        return -1;
    memcpy(c->h_send, send, bytes);
    if (hipMemcpyAsync(c->d_send, c->h_send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return rccl_break(c, "RCCL exchange: host->device copy failed");
    ncclResult_t r = g_rccl.AllGather(c->d_send, c->d_recv, bytes, ncclUint8, c->comm, c->stream);
    if (r != ncclSuccess)
        return rccl_break(c, std::string("ncclAllGather: ") + nccl_text(r));
    if (hipMemcpyAsync(c->h_recv, c->d_recv, bytes * (size_t)c->size, hipMemcpyDeviceToHost, c->stream) != hipSuccess
        || hipEventRecord(c->done, c->stream) != hipSuccess)
        return rccl_break(c, "RCCL exchange: device->host copy failed");
    if (rccl_wait_bounded(c) != 0)
        return -1;
    // the stream ran dry -- which it also does when RCCL gave an exchange up on its own: the bytes are trusted only when
    // the communicator reports no asynchronous error
    ncclResult_t state = ncclSuccess;
    if (g_rccl.CommGetAsyncError && g_rccl.CommGetAsyncError(c->comm, &state) == ncclSuccess && state != ncclSuccess
        && state != ncclInProgress)
        return rccl_break(c, std::string("RCCL exchange failed asynchronously: ") + nccl_text(state));
    memcpy(recv, c->h_recv, bytes * (size_t)c->size);
    return 0;
    }

static void rccl_destroy(void* p)
    {
    RcclCtx* c = (RcclCtx*)p;
    (void)hipSetDevice(c->device);
    if (c->stuck)
        {
        // a collective is still on the stream and nothing made it leave: the buffers it may touch, the stream and the
        // communicator are left to the end of the process rather than freed under it
        delete c;
        return;
        }
    if (c->comm)
        g_rccl.CommDestroy(c->comm);
    if (c->done)
        (void)hipEventDestroy(c->done);
    if (c->d_send)
        (void)hipFree(c->d_send);
    if (c->d_recv)
        (void)hipFree(c->d_recv);
    if (c->h_send)
        (void)hipHostFree(c->h_send);
    if (c->h_recv)
        (void)hipHostFree(c->h_recv);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    }
    } // namespace pgsd_amd

using namespace pgsd_amd;

extern "C" int pgsd_comm_rccl_unique_id(void* unique_id_128)
    try
    {
    if (!unique_id_128)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!load_rccl())
        return PGSD_ERROR_COMM;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess)
        {
        set_last_error("ncclGetUniqueId failed");
        return PGSD_ERROR_COMM;
        }
    memcpy(unique_id_128, &id, sizeof(id));
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

static int rccl_open_comm(const void* unique_id_128, int rank, int size, int device, pgsd_comm* out);

// What a rank can know by itself before it enters ncclCommInitRank (which waits for every rank): librccl loads with
// the entry points used here and the device can be selected.  The ranks agree on this first (pgsd.dist,
// benchmark_write.hip), so that no rank waits inside the bootstrap for one that could never have come.
extern "C" int pgsd_comm_rccl_available(int device)
    try
    {
    if (!pgsd_device_available())
        {
        set_last_error("pgsd_comm_rccl_available: no HIP device visible");
        return PGSD_ERROR_NO_DEVICE;
        }
    if (!load_rccl())
        return PGSD_ERROR_COMM;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || (device >= 0 && device >= n))
        {
        set_last_error("pgsd_comm_rccl_available: device " + std::to_string(device) + " of " + std::to_string(n));
        return PGSD_ERROR_DEVICE;
        }
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_comm_create_rccl(const void* unique_id_128, int rank, int size, int device, struct pgsd_comm* out)
    try
    {
    if (!out)
        return PGSD_ERROR_INVALID_ARGUMENT;
    return rccl_open_comm(unique_id_128, rank, size, device, out);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

static int rccl_open_comm(const void* unique_id_128, int rank, int size, int device, pgsd_comm* out)
    {
    if (!unique_id_128 || size < 1 || rank < 0 || rank >= size)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!pgsd_device_available())
        {
        set_last_error("pgsd_comm_create_rccl: no HIP device visible");
        return PGSD_ERROR_NO_DEVICE;
        }
    if (!load_rccl())
        return PGSD_ERROR_COMM;
    RcclCtx* c = new RcclCtx;
    c->rank = rank;
    c->size = size;
    if (device < 0)
        (void)hipGetDevice(&device);
    c->device = device;
    ncclUniqueId id;
    memcpy(&id, unique_id_128, sizeof(id));
    if (const char* t = getenv("PGSD_COMM_TIMEOUT_S"))
        {
        // seconds; 0 (or "off" / "inf"): no deadline -- wait for a peer as long as MPI_Allgather would (a rank may be
        // blocked on a slow file system or on the staging soft cap for longer than any fixed figure: ADVICE r4);
        // RCCL's own asynchronous errors still end the wait
        const double v = atof(t);
        if (v > 0)
            c->timeout_s = v;
        else if (strcmp(t, "0") == 0 || strcmp(t, "off") == 0 || strcmp(t, "inf") == 0)
            c->timeout_s = 0;
        }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess
        || hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess)
        {
        set_last_error("pgsd_comm_create_rccl: cannot select device / create stream");
        delete c;
        return PGSD_ERROR_DEVICE;
        }
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, size, id, rank);
    if (r != ncclSuccess)
        {
        set_last_error(std::string("ncclCommInitRank: ") + nccl_text(r));
        c->comm = nullptr;
        rccl_destroy(c);
        return PGSD_ERROR_COMM;
        }
    pgsd_comm pc;
    memset(&pc, 0, sizeof(pc));
    pc.ctx = c;
    pc.rank = rank;
    pc.size = size;
    pc.allgather = rccl_allgather;
    pc.barrier = nullptr; // 1-byte allgather
    pc.destroy = rccl_destroy;
    *out = pc;
    return PGSD_SUCCESS;
    }
