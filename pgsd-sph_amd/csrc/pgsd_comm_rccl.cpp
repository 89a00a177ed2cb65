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
#include "pgsd_internal.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <dlfcn.h>

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
    };

static RcclApi g_rccl;

static bool load_rccl()
    {
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
    };

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
    if (bytes == 0)
        return 0;
    if (hipSetDevice(c->device) != hipSuccess || !rccl_reserve(c, bytes))
        return -1;
    memcpy(c->h_send, send, bytes);
    if (hipMemcpyAsync(c->d_send, c->h_send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return -1;
    ncclResult_t r = g_rccl.AllGather(c->d_send, c->d_recv, bytes, ncclUint8, c->comm, c->stream);
    if (r != ncclSuccess)
        {
        set_last_error(std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
        return -1;
        }
    if (hipMemcpyAsync(c->h_recv, c->d_recv, bytes * (size_t)c->size, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess)
        return -1;
    memcpy(recv, c->h_recv, bytes * (size_t)c->size);
    return 0;
    }

static void rccl_destroy(void* p)
    {
    RcclCtx* c = (RcclCtx*)p;
    (void)hipSetDevice(c->device);
    if (c->comm)
        g_rccl.CommDestroy(c->comm);
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

extern "C" int pgsd_comm_init_rccl(const void* unique_id_128, int rank, int size, int device)
    try
    {
    pgsd_comm pc;
    int rc = rccl_open_comm(unique_id_128, rank, size, device, &pc);
    return rc == PGSD_SUCCESS ? pgsd_comm_set_default(&pc) : rc;
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
        set_last_error("pgsd_comm_init_rccl: no HIP device visible");
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
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        {
        set_last_error("pgsd_comm_init_rccl: cannot select device / create stream");
        delete c;
        return PGSD_ERROR_DEVICE;
        }
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, size, id, rank);
    if (r != ncclSuccess)
        {
        set_last_error(std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
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
