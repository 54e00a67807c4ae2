// Experiment (GPU box): issue rate of the matrix-core instructions the prefill GEMMs use, on random operands held in registers.
// One workgroup per CU, 4 or 8 waves (one or two per SIMD), 16 independent 16 x 16 accumulators per wave (the GEMM's 8 per quadrant x 2), a loop of
// `iters` x 16 MFMAs; reports shader cycles per MFMA (s_memtime around the loop, median over workgroups), the in-kernel clock (s_memtime / s_memrealtime) and
// the wall-clock TFLOP/s of the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(const uint4* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ stamps, int iters)
{
    const int lane = threadIdx.x & 63;
    uint4 r0 = src[threadIdx.x], r1 = src[threadIdx.x + 512], r2 = src[threadIdx.x + 1024], r3 = src[threadIdx.x + 1536];
    f32x4 acc[16];
    f32x16 acc32[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
    {
        if constexpr (KIND == 0)          // v_mfma_f32_16x16x32_bf16
        {
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, r0), a1 = __builtin_bit_cast(bf16x8, r1), b0 = __builtin_bit_cast(bf16x8, r2), b1 = __builtin_bit_cast(bf16x8, r3);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0);
        }
        else if constexpr (KIND == 1)     // v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3, unit scales
        {
            struct P { uint4 lo, hi; };
            const i32x8 a0 = __builtin_bit_cast(i32x8, (P{r0, r1})), a1 = __builtin_bit_cast(i32x8, (P{r1, r2})), b0 = __builtin_bit_cast(i32x8, (P{r2, r3})), b1 = __builtin_bit_cast(i32x8, (P{r3, r0}));
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0, 127, 0, 127);
        }
        else if constexpr (KIND == 2)     // v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 x e4m3 (4 accumulators of 16 registers: the same 64 registers)
        {
            struct P { uint4 lo, hi; };
            const i32x8 a0 = __builtin_bit_cast(i32x8, (P{r0, r1})), a1 = __builtin_bit_cast(i32x8, (P{r1, r2})), b0 = __builtin_bit_cast(i32x8, (P{r2, r3})), b1 = __builtin_bit_cast(i32x8, (P{r3, r0}));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc32[i & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc32[i & 3], 0, 0, 0, 127, 0, 127);
        }
        else                               // v_mfma_f32_32x32x16_bf16
        {
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, r0), a1 = __builtin_bit_cast(bf16x8, r1), b0 = __builtin_bit_cast(bf16x8, r2), b1 = __builtin_bit_cast(bf16x8, r3);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc32[i & 3], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc32[i][0] + acc32[i][15];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
    (void)lane;
}

template <int KIND>
static int run(const char* name, double flops_per_mfma, int mfma_per_iter, int threads, const uint4* src, float* sink, unsigned long long* stamps, int iters)
{
    const int nwg = 256;
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(nwg), dim3(threads), 0, 0, src, sink, stamps, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, 0));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(nwg), dim3(threads), 0, 0, src, sink, stamps, iters);
    CK(hipEventRecord(t1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, t0, t1));
    std::vector<unsigned long long> h(2 * nwg);
    CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int i = 0; i < nwg; ++i) { cyc.push_back((double)h[2 * i]); clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }   // s_memrealtime ticks at 100 MHz: cycles per tick x 0.1 = GHz
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const int waves = threads / 64;
    const double per_simd = (double)iters * mfma_per_iter;      // wave 0's own stream; its SIMD runs waves / 4 of them
    const double tf = flops_per_mfma * iters * mfma_per_iter * waves * nwg / (ms / reps * 1e-3) / 1e12;
    printf("{\"mfma\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_mfma_of_one_wave\": %.2f, \"in_kernel_clock_GHz\": %.3f, \"wall_TFLOPs\": %.0f}\n", name, waves / 4,
           cyc[nwg / 2] / per_simd, clk[nwg / 2], tf);
    fflush(stdout);
    return 0;
}

int main()
{
    uint4* src; float* sink; unsigned long long* stamps;
    CK(hipMalloc(&src, 2048 * sizeof(uint4)));
    std::vector<unsigned> h(2048 * 4);
    srand(7);
    for (auto& v : h)
    {
        // random bf16 pairs of magnitude < 1 (also valid, finite e4m3 bytes: exponent field never all ones)
        unsigned w = 0;
        for (int b = 0; b < 4; ++b) { unsigned byte = rand() & 0xff; if ((byte & 0x78) == 0x78) byte &= ~0x40u; w |= byte << (8 * b); }
        v = (w & 0xBF7FBF7Fu);
    }
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&stamps, 512 * sizeof(unsigned long long)));
    const int iters = 20000;
    for (int threads : {256, 512})
    {
        if (run<0>("v_mfma_f32_16x16x32_bf16", 16.0 * 16 * 32 * 2, 16, threads, src, sink, stamps, iters)) return 1;
        if (run<3>("v_mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2, 8, threads, src, sink, stamps, iters)) return 1;
        if (run<1>("v_mfma_scale_f32_16x16x128_f8f6f4(e4m3)", 16.0 * 16 * 128 * 2, 16, threads, src, sink, stamps, iters)) return 1;
        if (run<2>("v_mfma_scale_f32_32x32x64_f8f6f4(e4m3)", 32.0 * 32 * 64 * 2, 8, threads, src, sink, stamps, iters)) return 1;
    }
    return 0;
}
