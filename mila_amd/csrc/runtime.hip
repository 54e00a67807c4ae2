// Runtime entry points of the C ABI: error text, device queries, streams, device memory.
// Counterpart of the reference's ExecutionContext<Cuda> resources
// (Compute/Devices/Cuda/CudaExecutionContext.ixx:106-368) and cudaCheck (Helpers/CudaUtils.h).
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <cstdlib>

#include "common.h"
#include "internal.h"

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

static TuneVar*& tune_head() { static TuneVar* head = nullptr; return head; }
void register_tune_var(TuneVar* v) { v->next = tune_head(); tune_head() = v; }
static TuneVar* find_tune(const char* name)
{
    for (TuneVar* v = tune_head(); v; v = v->next)
        if (std::strcmp(v->name, name) == 0) return v;
    return nullptr;
}

static thread_local char g_last_form[256] = "";
static thread_local size_t g_last_form_len = 0;
void note_form(const char* form)
{
    if (!tuning_hooks_enabled()) return;      // a record for tests and tools only: a product process keeps none
    // forms of one entry point accumulate, '+'-separated (a call may run a tile kernel on the leading rows and another form on the rest); cleared by last_form()
    const size_t n = std::strlen(form);
    if (g_last_form_len + n + 2 >= sizeof(g_last_form)) return;
    if (g_last_form_len) g_last_form[g_last_form_len++] = '+';
    std::memcpy(g_last_form + g_last_form_len, form, n + 1);
    g_last_form_len += n;
}

}  // namespace mila

using namespace mila;

extern "C" {

const char* mila_cdna4_last_error(void) { return g_last_error; }

int mila_cdna4_abi_version(void) { return 4; }

// ---- named tuning hooks (csrc/internal.h; tests / tools only) ----
#define MILA_TUNING_GATE() do { if (!::mila::tuning_hooks_enabled()) return ::mila::set_error(MILA_E_UNSUPPORTED, "%s: tuning hooks are inert unless MILA_CDNA4_TUNING=1 was set when the library was loaded", __func__); } while (0)
int mila_cdna4_tune(const char* name, int value)
{
    MILA_TUNING_GATE();
    MILA_REQUIRE(name != nullptr, "tune: null name");
    TuneVar* v = find_tune(name);
    MILA_REQUIRE(v != nullptr, "tune: no tuning variable named '%s' (mila_cdna4_tune_list names them)", name);
    *v->var = value;
    return MILA_OK;
}
int mila_cdna4_tune_get(const char* name, int* value)
{
    MILA_TUNING_GATE();
    MILA_REQUIRE(name && value, "tune_get: null argument");
    TuneVar* v = find_tune(name);
    MILA_REQUIRE(v != nullptr, "tune_get: no tuning variable named '%s'", name);
    *value = *v->var;
    return MILA_OK;
}
int mila_cdna4_tune_reset(void)
{
    MILA_TUNING_GATE();
    for (TuneVar* v = tune_head(); v; v = v->next) *v->var = v->def;
    return MILA_OK;
}
/* "name=value (default d)\n" per variable; returns the bytes needed including the terminator */
size_t mila_cdna4_tune_list(char* buf, size_t cap)
{
    size_t need = 1;
    for (TuneVar* v = tune_head(); v; v = v->next) need += (size_t)snprintf(nullptr, 0, "%s=%d (default %d)\n", v->name, *v->var, v->def);
    if (buf && cap)
    {
        size_t off = 0;
        buf[0] = 0;
        for (TuneVar* v = tune_head(); v && off + 1 < cap; v = v->next) off += (size_t)snprintf(buf + off, cap - off, "%s=%d (default %d)\n", v->name, *v->var, v->def);
    }
    return need;
}
/* the kernel forms noted by this thread's entry points since the last call, '+'-separated; clears the record.  Returns the bytes needed including the terminator */
size_t mila_cdna4_last_form(char* buf, size_t cap)
{
    const size_t need = g_last_form_len + 1;
    if (buf && cap) { std::strncpy(buf, g_last_form, cap - 1); buf[cap - 1] = 0; }
    g_last_form[0] = 0;
    g_last_form_len = 0;
    return need;
}

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
