// Quantize-on-load: bf16 -> fp8 E4M3FN per output channel, bf16 -> packed fp4 E2M1 per group.
// Integer outputs are BIT-EXACT with the reference kernels:
//   OPS/Linear/Kernels/Quantization/CudaFp8WeightQuantization.cu:57-121
//     scale = absmax > 0 ? absmax / 448 : 1;  inv = 1.0f / scale;  q = e4m3_rne_satfinite(x * inv)
//   OPS/Linear/Kernels/Quantization/CudaFp4WeightQuantization.cu:54-144
//     scale = absmax > 0 ? absmax / 6 : 1;    inv = 1.0f / scale;  nibble = thresholds(|x * inv|)
// Both divisions are IEEE-correct (hipcc default, like nvcc without --use_fast_math); absmax is an
// exact reduction, so the reduction order does not matter.  The fp8 encode is done in integer
// arithmetic on the fp32 bits (RNE at mantissa bit 20, saturate to 0x7e) so it does not depend on
// the FP8 hardware-convert overflow mode.
#include "common.h"

namespace mila {

// OCP E4M3FN <- f32, RNE, saturate-to-finite, NaN -> 0x7f (== __nv_fp8_e4m3(float))
__device__ __forceinline__ uint32_t f32_to_e4m3_rne_sat(float v)
{
    const uint32_t u = __float_as_uint(v);
    const uint32_t sign = (u >> 24) & 0x80u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return sign | 0x7fu;
    if (a >= 0x43e80000u) return sign | 0x7eu;          // >= 464 (midpoint 448/480) or inf
    if (a < 0x3c800000u)                                 // < 2^-6: subnormal grid, step 2^-9
    {
        const float q = __builtin_rintf(__uint_as_float(a) * 512.0f);   // v_rndne_f32, 0..8
        return sign | (uint32_t)q;                                       // 8 == 0x08 == 2^-6
    }
    a += 0x7ffffu + ((a >> 20) & 1u);                    // RNE to 3 mantissa bits
    const uint32_t code = (((a >> 23) - 120u) << 3) | ((a >> 20) & 7u);
    return sign | (code > 0x7eu ? 0x7eu : code);
}

// CudaFp4WeightQuantization.cu:54-70: strict '<' breakpoints, sign from x < 0
__device__ __forceinline__ uint32_t f32_to_e2m1(float x)
{
    const uint32_t sign = (x < 0.0f) ? 8u : 0u;
    const float a = fabsf(x);
    uint32_t mag;
    if (a < 0.25f) mag = 0;
    else if (a < 0.75f) mag = 1;
    else if (a < 1.25f) mag = 2;
    else if (a < 1.75f) mag = 3;
    else if (a < 2.5f) mag = 4;
    else if (a < 3.5f) mag = 5;
    else if (a < 5.0f) mag = 6;
    else mag = 7;
    return sign | mag;
}

// one 256-thread workgroup per output channel
__global__ __launch_bounds__(256) void quantize_fp8_per_channel_kernel(uint8_t* __restrict__ dst,
                                                                       float* __restrict__ scales,
                                                                       const uint16_t* __restrict__ src, int K)
{
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const uint16_t* s = src + row * K;
    uint8_t* d = dst + row * K;
    const int nvec = K / 8;
    float m = 0.0f;
    for (int i = threadIdx.x; i < nvec; i += 256)
    {
        const u32x4 v = ld16(s + (size_t)i * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fmaxf(fabsf(bf16_lo(v[j])), fabsf(bf16_hi(v[j]))));
    }
    for (int i = nvec * 8 + threadIdx.x; i < K; i += 256) m = fmaxf(m, fabsf(bf16_bits_to_f32(s[i])));
    const float absmax = block_max<4>(m, red);
    const float scale = (absmax > 0.0f) ? (absmax / 448.0f) : 1.0f;
    const float inv = 1.0f / scale;
    if (threadIdx.x == 0) scales[row] = scale;
    for (int i = threadIdx.x; i < nvec; i += 256)
    {
        const u32x4 v = ld16(s + (size_t)i * 8);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
        {
            lo |= f32_to_e4m3_rne_sat(bf16_lo(v[j]) * inv) << (16 * j);
            lo |= f32_to_e4m3_rne_sat(bf16_hi(v[j]) * inv) << (16 * j + 8);
            hi |= f32_to_e4m3_rne_sat(bf16_lo(v[j + 2]) * inv) << (16 * j);
            hi |= f32_to_e4m3_rne_sat(bf16_hi(v[j + 2]) * inv) << (16 * j + 8);
        }
        *reinterpret_cast<u32x2*>(d + (size_t)i * 8) = u32x2{lo, hi};
    }
    for (int i = nvec * 8 + threadIdx.x; i < K; i += 256)
        d[i] = (uint8_t)f32_to_e4m3_rne_sat(bf16_bits_to_f32(s[i]) * inv);
}

// thread per column PAIR (one output byte); a group of G columns is G/2 consecutive lanes.
template <int G>
__global__ __launch_bounds__(256) void quantize_fp4_per_group_kernel(uint8_t* __restrict__ dst,
                                                                     float* __restrict__ scales,
                                                                     const uint16_t* __restrict__ src, int K,
                                                                     int64_t total_pairs)
{
    const int64_t pair = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = pair < total_pairs;
    const uint32_t packed = in ? *reinterpret_cast<const uint32_t*>(src + pair * 2) : 0u;
    const float v0 = bf16_lo(packed), v1 = bf16_hi(packed);
    float m = fmaxf(fabsf(v0), fabsf(v1));
#pragma unroll
    for (int off = G / 4; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const float scale = (m > 0.0f) ? (m / 6.0f) : 1.0f;
    const float inv = 1.0f / scale;
    if (!in) return;
    const uint32_t n0 = f32_to_e2m1(v0 * inv), n1 = f32_to_e2m1(v1 * inv);
    dst[pair] = (uint8_t)(n0 | (n1 << 4));
    if ((pair % (G / 2)) == 0) scales[pair / (G / 2)] = scale;   // [N, K/G] row-major == flat group index
    (void)K;
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_quantize_fp8_per_channel(uint8_t* dst, float* scales, const uint16_t* src_bf16, int N, int K,
                                        mila_stream_t stream)
{
    MILA_REQUIRE(dst && scales && src_bf16, "quantize_fp8_per_channel: null pointer");
    MILA_REQUIRE(N > 0 && K > 0, "quantize_fp8_per_channel: N and K must be positive (N=%d K=%d)", N, K);
    MILA_REQUIRE(K % 8 == 0, "quantize_fp8_per_channel: K=%d must be a multiple of 8", K);
    hipLaunchKernelGGL(quantize_fp8_per_channel_kernel, dim3(N), dim3(256), 0, as_stream(stream), dst, scales,
                       src_bf16, K);
    MILA_LAUNCH_CHECK("quantize_fp8_per_channel");
}

int mila_cdna4_quantize_fp4_per_group(uint8_t* dst_packed, float* scales, const uint16_t* src_bf16, int N, int K,
                                      int group, mila_stream_t stream)
{
    MILA_REQUIRE(dst_packed && scales && src_bf16, "quantize_fp4_per_group: null pointer");
    MILA_REQUIRE(N > 0 && K > 0, "quantize_fp4_per_group: N and K must be positive (N=%d K=%d)", N, K);
    MILA_REQUIRE(group == 64 || group == 128, "quantize_fp4_per_group: group size must be 64 or 128 (got %d)", group);
    MILA_REQUIRE(K % group == 0, "quantize_fp4_per_group: K=%d must be a multiple of the group size %d", K, group);
    const int64_t pairs = (int64_t)N * K / 2;
    const int blocks = ceil_div(pairs, 256);
    if (group == 128)
        hipLaunchKernelGGL(quantize_fp4_per_group_kernel<128>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           dst_packed, scales, src_bf16, K, pairs);
    else
        hipLaunchKernelGGL(quantize_fp4_per_group_kernel<64>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           dst_packed, scales, src_bf16, K, pairs);
    MILA_LAUNCH_CHECK("quantize_fp4_per_group");
}

}  // extern "C"
