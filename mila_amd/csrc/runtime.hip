// Runtime entry points of the C ABI: error text, device queries, streams, device memory.
// Counterpart of the reference's ExecutionContext<Cuda> resources
// (Compute/Devices/Cuda/CudaExecutionContext.ixx:106-368) and cudaCheck (Helpers/CudaUtils.h).
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <cstdlib>

#include "common.h"

namespace mila {

static thread_local char g_last_error[512] = "";

bool tuning_hooks_enabled()
{
    static const bool on = [] { const char* v = std::getenv("MILA_CDNA4_TUNING"); return v && v[0] == '1'; }();
    return on;
}

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

int check_hip(hipError_t e, const char* what)
{
    if (e == hipSuccess) return MILA_OK;
    return set_error(MILA_E_RUNTIME, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace mila

using namespace mila;

extern "C" {

const char* mila_cdna4_last_error(void) { return g_last_error; }

int mila_cdna4_abi_version(void) { return 4; }

int mila_cdna4_device_count(int* count)
{
    MILA_REQUIRE(count != nullptr, "device_count: null output");
    return check_hip(hipGetDeviceCount(count), "hipGetDeviceCount");
}

int mila_cdna4_set_device(int device) { return check_hip(hipSetDevice(device), "hipSetDevice"); }

int mila_cdna4_device_info(int device, char* name, int* compute_units, size_t* hbm_bytes)
{
    hipDeviceProp_t p;
    int rc = check_hip(hipGetDeviceProperties(&p, device), "hipGetDeviceProperties");
    if (rc != MILA_OK) return rc;
    if (name) { strncpy(name, p.gcnArchName, 63); name[63] = 0; }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return MILA_OK;
}

int mila_cdna4_stream_create(mila_stream_t* stream)
{
    MILA_REQUIRE(stream != nullptr, "stream_create: null output");
    hipStream_t s;
    int rc = check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    if (rc == MILA_OK) *stream = s;
    return rc;
}

int mila_cdna4_stream_destroy(mila_stream_t stream)
{
    return check_hip(hipStreamDestroy(as_stream(stream)), "hipStreamDestroy");
}

int mila_cdna4_stream_synchronize(mila_stream_t stream)
{
    return check_hip(hipStreamSynchronize(as_stream(stream)), "hipStreamSynchronize");
}

int mila_cdna4_malloc(void** ptr, size_t bytes)
{
    MILA_REQUIRE(ptr != nullptr, "malloc: null output");
    return check_hip(hipMalloc(ptr, bytes ? bytes : 1), "hipMalloc");
}

int mila_cdna4_free(void* ptr) { return check_hip(hipFree(ptr), "hipFree"); }

int mila_cdna4_host_alloc_pinned(void** host_ptr, size_t bytes)
{
    MILA_REQUIRE(host_ptr != nullptr, "host_alloc_pinned: null output");
    return check_hip(hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault), "hipHostMalloc");
}

int mila_cdna4_host_free_pinned(void* host_ptr) { return check_hip(hipHostFree(host_ptr), "hipHostFree"); }

int mila_cdna4_memcpy_h2d(void* dst, const void* host_src, size_t bytes, mila_stream_t stream)
{
    return check_hip(hipMemcpyAsync(dst, host_src, bytes, hipMemcpyHostToDevice, as_stream(stream)), "memcpy_h2d");
}

int mila_cdna4_memcpy_d2h(void* host_dst, const void* src, size_t bytes, mila_stream_t stream)
{
    return check_hip(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)), "memcpy_d2h");
}

int mila_cdna4_memcpy_d2d(void* dst, const void* src, size_t bytes, mila_stream_t stream)
{
    return check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)), "memcpy_d2d");
}

int mila_cdna4_memset_zero(void* dst, size_t bytes, mila_stream_t stream)
{
    return check_hip(hipMemsetAsync(dst, 0, bytes, as_stream(stream)), "memset_zero");
}

}  // extern "C"
