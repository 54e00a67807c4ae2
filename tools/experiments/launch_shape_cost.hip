// Experiment (GPU box): what does the SHAPE of a launch cost?  Chains of dependent, nearly empty kernels replayed from a hipGraph;
// microseconds per launch by (workgroups, threads per workgroup, dynamic LDS).  Each kernel reads one word written by its predecessor.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void tiny(const float* in, float* out)
{
    extern __shared__ float lds[];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = in[0] + 1.0f;
    if (in[1] == 12345.0f) lds[threadIdx.x] = 1.0f;      // never true: keeps the LDS allocation alive
}

// a body shaped like the decode matvec's skeleton: every thread loads 16 B of x, the workgroup stages it in LDS, one barrier, a wave reduction, lane 0 stores
__global__ void skeleton(const float4* x, float* out, int n16)
{
    extern __shared__ float4 xs[];
    const int tid = threadIdx.x;
    for (int i = tid; i < n16; i += blockDim.x) xs[i] = x[i];
    __syncthreads();
    float a = 0.0f;
    for (int i = tid & 63; i < n16; i += 64 * 8) a += xs[i].x;
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((tid & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (tid >> 6)] = a;
}

int main()
{
    float* buf;
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMemset(buf, 0, 1 << 20));
    float4* x;
    CK(hipMalloc(&x, 65536));
    CK(hipMemset(x, 0, 65536));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tiny), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&skeleton), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int N = 96, REPS = 50;
    auto run = [&](const char* name, auto&& enqueue) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) enqueue(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(t0, s));
        for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(t1, s));
        CK(hipEventSynchronize(t1));
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("%-44s %6.2f us per launch\n", name, ms * 1e3 / (REPS * N));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return 0;
    };
    struct Shape { int wgs, threads, lds; } shapes[] = {{1, 1, 0}, {1, 64, 0}, {256, 64, 0}, {256, 256, 0}, {256, 512, 0}, {256, 1024, 0}, {256, 1024, 32768}, {256, 1024, 65536},
                                                        {512, 512, 0}, {1024, 256, 0}, {2048, 256, 0}, {240, 1024, 16384}, {512, 1024, 8192}};
    for (auto sh : shapes)
    {
        char name[96];
        snprintf(name, sizeof name, "tiny     %4d wg x %4d thr, %5d B LDS", sh.wgs, sh.threads, sh.lds);
        if (run(name, [&](int i) { hipLaunchKernelGGL(tiny, dim3(sh.wgs), dim3(sh.threads), sh.lds, s, buf + (i & 1), buf + ((i + 1) & 1)); })) return 1;
    }
    struct Sk { int wgs, threads, n16; } sks[] = {{256, 1024, 480}, {256, 512, 480}, {256, 256, 480}, {256, 1024, 1920}, {256, 512, 1920}, {512, 512, 480}};
    for (auto sk : sks)
    {
        char name[96];
        snprintf(name, sizeof name, "skeleton %4d wg x %4d thr, x = %5d B", sk.wgs, sk.threads, sk.n16 * 16);
        if (run(name, [&](int i) { hipLaunchKernelGGL(skeleton, dim3(sk.wgs), dim3(sk.threads), sk.n16 * 16, s, x, buf + 64 + (i & 1) * 16384, sk.n16); })) return 1;
    }
    return 0;
}
