// pgsd_device_stub.cpp -- NOT part of libpgsd_amd.so.  Link-time stand-ins for the device
// pipeline so that the HOST file layer (pgsd_container.cpp, pgsd_placement.cpp, pgsd_read.cpp, pgsd_comm.cpp, pgsd_io.cpp) can be built
// with gcc -fsanitize=address,undefined and exercised by the scenario driver on a CPU box
// (`make asan`; GPU AddressSanitizer is not available on the pool).  Every device entry point
// fails with PGSD_ERROR_NO_DEVICE, exactly as the real library does without a GPU.
#include "pgsd_internal.hpp"

namespace pgsd_amd
    {
DevicePipeline* device_pipeline_create(const pgsd_device_config&, int, bool, std::string* err)
    {
    if (err)
        *err = "sanitizer build: no device pipeline";
    return nullptr;
    }
void device_pipeline_destroy(DevicePipeline*) { }
int device_pipeline_device(DevicePipeline*) { return -1; }
int device_pipeline_submit(DevicePipeline*, std::vector<DeviceChunk>&, uint64_t, std::string*) { return PGSD_ERROR_NO_DEVICE; }
int device_pipeline_stage(DevicePipeline*, std::vector<DeviceChunk>&, uint64_t, int*, std::string*) { return PGSD_ERROR_NO_DEVICE; }
int device_pipeline_commit(DevicePipeline*, int, size_t, long long, void*, std::string*) { return PGSD_ERROR_NO_DEVICE; }
int device_pipeline_compare(DevicePipeline*, int, size_t, size_t, const void* const*, const uint64_t*, uint8_t*, std::string*) { return PGSD_ERROR_NO_DEVICE; }
int device_pipeline_copy_staged(DevicePipeline*, int, size_t, size_t, void* const*, std::string*) { return PGSD_ERROR_NO_DEVICE; }
void device_pipeline_kick(DevicePipeline*) { }
void device_pipeline_write_host(DevicePipeline*, const void*, size_t, long long) { }
bool device_pipeline_single_writer(DevicePipeline*) { return false; }
int device_pipeline_wait_packed(DevicePipeline*, std::string*) { return PGSD_ERROR_NO_DEVICE; }
void device_pipeline_set_source_stream(DevicePipeline*, void*) { }
int device_pipeline_read(DevicePipeline*, long long, size_t, const pgsd_unpack_job&, uint64_t, std::string*) { return PGSD_ERROR_NO_DEVICE; }
int device_pipeline_wait_read(DevicePipeline*, std::string*) { return PGSD_ERROR_NO_DEVICE; }
int device_pipeline_drain(DevicePipeline*, std::string*) { return PGSD_ERROR_NO_DEVICE; }
void device_pipeline_stats(DevicePipeline*, pgsd_device_stats*, int) { }
    } // namespace pgsd_amd

extern "C" int pgsd_device_available(void) { return 0; }
extern "C" int pgsd_comm_rccl_unique_id(void*) { return PGSD_ERROR_NO_DEVICE; }
extern "C" int pgsd_comm_create_rccl(const void*, int, int, int, struct pgsd_comm*) { return PGSD_ERROR_NO_DEVICE; }
extern "C" int pgsd_comm_rccl_available(int) { return PGSD_ERROR_NO_DEVICE; }
extern "C" int pgsd_device_release_parked(void) { return 0; }
