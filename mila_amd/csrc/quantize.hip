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

// ---- W4A8 prefill staging (the reference's default fp4-policy prefill, CudaLinearOp.ixx:646-715) ----------------------
// weight_fp8_scale = max(max(group scale), 1e-12) * (6 / 448)          (CudaW4A16Gemm.cu:244-294; exact max, any order)
// Many workgroups, no scratch: f(m) = max(m, 1e-12) * (6 / 448) is non-decreasing and positive, and positive floats order like their bit
// patterns, so max over blocks of f(block max) -- an integer atomicMax on the zero-initialised output -- IS f(max over all scales), bit for bit.
__global__ __launch_bounds__(1024) void fp4_weight_fp8_scale_kernel(float* __restrict__ out, const float* __restrict__ scales, int64_t n)
{
    __shared__ float red[16];
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) m = fmaxf(m, scales[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
        atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(fmaxf(m, 1e-12f) * (6.0f / 448.0f)));
    }
}

// four finite f32 -> four OCP e4m3 bytes with the hardware convert (v_cvt_pk_fp8_f32: RNE), saturating to +-448 first as
// f32_to_e4m3_rne_sat does (|v| >= 464 -> 0x7e); inputs here are products of finite weights and scales, never NaN
__device__ __forceinline__ uint32_t f32x4_to_e4m3x4_hw(float a, float b, float c, float d)
{
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}

// packed fp4 -> e4m3: out[2b], out[2b+1] = e4m3(lut(nibble) * (group scale * (1 / weight_fp8_scale)))  (CudaW4A16Gemm.cu:300-323)
// One thread per 16 packed bytes (32 outputs = two 16-byte stores), grid-stride over the whole matrix.  The nibbles are decoded by
// v_cvt_scalef32_pk_f32_fp4 (scale 1.0: exact), multiplied in f32 and encoded by v_cvt_pk_fp8_f32: 1.5 VALU operations per
// element where the bit-arithmetic encoder needed ~25 (the kernel was VALU-bound at 2 TB/s); same bytes (tests/test_linear_gpu.py).
__global__ __launch_bounds__(256) void upcast_fp4_to_fp8_kernel(uint8_t* __restrict__ out, const uint8_t* __restrict__ packed,
                                                                const float* __restrict__ scales, const float* __restrict__ w_scale,
                                                                int K, int group_shift, int64_t nvec)
{
    const float inv_ws = 1.0f / w_scale[0];
    const int vec_per_row = K >> 5;                        // 32 elements per vector
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride)
    {
        const int64_t row = v / vec_per_row;
        const int k0 = (int)(v - row * vec_per_row) << 5;
        // a 32-element vector never straddles a group (group >= 64)
        const float sc = scales[row * (int64_t)(K >> group_shift) + (k0 >> group_shift)] * inv_ws;
        const u32x4 pk = *reinterpret_cast<const u32x4*>(packed + v * 16);
        u32x4 o[2];
#pragma unroll
        for (int w = 0; w < 4; ++w)
        {
            const f32x2 e01 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(pk[w], 1.0f, 0), e23 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(pk[w], 1.0f, 1);
            const f32x2 e45 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(pk[w], 1.0f, 2), e67 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(pk[w], 1.0f, 3);
            o[w >> 1][2 * (w & 1)] = f32x4_to_e4m3x4_hw(e01[0] * sc, e01[1] * sc, e23[0] * sc, e23[1] * sc);
            o[w >> 1][2 * (w & 1) + 1] = f32x4_to_e4m3x4_hw(e45[0] * sc, e45[1] * sc, e67[0] * sc, e67[1] * sc);
        }
        uint8_t* dst = out + v * 32;
        *reinterpret_cast<u32x4*>(dst) = o[0];
        *reinterpret_cast<u32x4*>(dst + 16) = o[1];
    }
}

// per-token activation quantization: scale = max(absmax(row), 1e-12) / 448, q = e4m3(x * (1 / scale))  (CudaFp8Prefill.cu:108-160)
// One workgroup per row; the row stays in registers between the absmax pass and the encoding pass (NCH 16-byte chunks per thread:
// K <= 2048 NCH; NCH = 0 re-reads the row for longer ones).
template <int NCH>
__global__ __launch_bounds__(256) void quantize_fp8_per_token_kernel(uint8_t* __restrict__ dst, float* __restrict__ scales,
                                                                     const uint16_t* __restrict__ src, int K)
{
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const uint16_t* s = src + row * K;
    uint8_t* d = dst + row * K;
    const int nvec = K / 8;
    float m = 0.0f;
    u32x4 keep[NCH > 0 ? NCH : 1];
    if constexpr (NCH > 0)
    {
#pragma unroll
        for (int k = 0; k < NCH; ++k)
        {
            const int i = threadIdx.x + 256 * k;
            keep[k] = (i < nvec) ? ld16(s + (size_t)i * 8) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(bf16_lo(keep[k][e])), fabsf(bf16_hi(keep[k][e]))));
    }
    else
    {
        for (int i = threadIdx.x; i < nvec; i += 256)
        {
            const u32x4 v = ld16(s + (size_t)i * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(bf16_lo(v[e])), fabsf(bf16_hi(v[e]))));
        }
    }
    m = block_max<4>(m, red);
    const float scale = fmaxf(m, 1e-12f) / 448.0f;
    if (threadIdx.x == 0) scales[row] = scale;
    const float inv = 1.0f / scale;
    // |x * inv| <= 448 (1 + 2^-23): the hardware convert (RNE) needs no clamp here, and a NaN stays a NaN (0x7f)
    auto encode = [&](const u32x4 v, int i) {
        u32x2 o;
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            int r = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_lo(v[2 * h]) * inv, bf16_hi(v[2 * h]) * inv, 0, false);
            r = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_lo(v[2 * h + 1]) * inv, bf16_hi(v[2 * h + 1]) * inv, r, true);
            o[h] = (uint32_t)r;
        }
        *reinterpret_cast<u32x2*>(d + (size_t)i * 8) = o;
    };
    if constexpr (NCH > 0)
    {
#pragma unroll
        for (int k = 0; k < NCH; ++k)
        {
            const int i = threadIdx.x + 256 * k;
            if (i < nvec) encode(keep[k], i);
        }
    }
    else
    {
        for (int i = threadIdx.x; i < nvec; i += 256) encode(ld16(s + (size_t)i * 8), i);
    }
}

int launch_gemm_fp8(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, const uint16_t* bias,
                    int M, int K, int N, hipStream_t s);
bool gemm256_geglu_applicable(int M, int K, int F);
int launch_gemm_fp8_geglu(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, int M, int K, int F,
                          hipStream_t s);
// gemm256.hip: the split-K forms behind a caller workspace
size_t gemm_fp8_ws_bytes(int M, int K, int N);
int launch_gemm_fp8_ws(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, const uint16_t* bias, int M, int K, int N, hipStream_t s,
                       void* ws);
bool gemm_fp8_geglu_steps_aside(int M, int K, int F);

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

int mila_cdna4_fp4_weight_fp8_scale(float* out_scale, const float* group_scales, int64_t num_scales, mila_stream_t stream)
{
    MILA_REQUIRE(out_scale && group_scales && num_scales > 0, "fp4_weight_fp8_scale: bad arguments");
    hipError_t e = hipMemsetAsync(out_scale, 0, sizeof(float), as_stream(stream));
    if (e != hipSuccess) return check_hip(e, "fp4_weight_fp8_scale: hipMemsetAsync");
    const int blocks = (int)std::min<int64_t>((num_scales + 4095) / 4096, 256);
    hipLaunchKernelGGL(fp4_weight_fp8_scale_kernel, dim3(blocks), dim3(1024), 0, as_stream(stream), out_scale, group_scales, num_scales);
    MILA_LAUNCH_CHECK("fp4_weight_fp8_scale");
}

int mila_cdna4_upcast_fp4_to_fp8(uint8_t* out, const uint8_t* packed, const float* scales, const float* weight_fp8_scale, int N, int K,
                                 int group, mila_stream_t stream)
{
    MILA_REQUIRE(out && packed && scales && weight_fp8_scale, "upcast_fp4_to_fp8: null pointer");
    MILA_REQUIRE(N > 0 && K > 0 && (group == 64 || group == 128) && K % group == 0 && K % 32 == 0, "upcast_fp4_to_fp8: bad sizes (N=%d K=%d group=%d)", N, K, group);
    const int64_t nvec = (int64_t)N * (K >> 5);
    const int blocks = (int)std::min<int64_t>((nvec + 255) / 256, 4096);
    hipLaunchKernelGGL(upcast_fp4_to_fp8_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), out, packed, scales, weight_fp8_scale, K, group == 128 ? 7 : 6,
                       nvec);
    MILA_LAUNCH_CHECK("upcast_fp4_to_fp8");
}

int mila_cdna4_quantize_fp8_per_token(uint8_t* dst, float* scales, const uint16_t* src, int M, int K, mila_stream_t stream)
{
    MILA_REQUIRE(dst && scales && src, "quantize_fp8_per_token: null pointer");
    MILA_REQUIRE(M > 0 && K > 0 && K % 8 == 0, "quantize_fp8_per_token: bad sizes (M=%d K=%d)", M, K);
    const int nch = (K / 8 + 255) / 256;
    if (nch <= 2) hipLaunchKernelGGL(quantize_fp8_per_token_kernel<2>, dim3(M), dim3(256), 0, as_stream(stream), dst, scales, src, K);
    else if (nch <= 4) hipLaunchKernelGGL(quantize_fp8_per_token_kernel<4>, dim3(M), dim3(256), 0, as_stream(stream), dst, scales, src, K);
    else if (nch <= 8) hipLaunchKernelGGL(quantize_fp8_per_token_kernel<8>, dim3(M), dim3(256), 0, as_stream(stream), dst, scales, src, K);
    else hipLaunchKernelGGL(quantize_fp8_per_token_kernel<0>, dim3(M), dim3(256), 0, as_stream(stream), dst, scales, src, K);
    MILA_LAUNCH_CHECK("quantize_fp8_per_token");
}

// every M: the LDS-DMA kernels on the leading multiple of 256 rows where one applies, the masked tail kernel on the rest (gemm256.hip: launch_gemm_fp8)
int mila_cdna4_gemm_fp8_applicable(int M, int K, int N) { return (M > 0 && K > 0 && N > 0 && K % 16 == 0) ? 1 : 0; }

int mila_cdna4_gemm_fp8_scaled(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, const float* weight_scale,
                               const uint16_t* bias, int M, int K, int N, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X8 && W8 && x_scales && weight_scale, "gemm_fp8_scaled: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_fp8_applicable(M, K, N), "gemm_fp8_scaled: shape (M=%d, K=%d, N=%d) has no fp8 MFMA kernel (ask gemm_fp8_applicable)", M, K, N);
    return launch_gemm_fp8(Y, X8, W8, x_scales, Fp8WScale{weight_scale, 0}, bias, M, K, N, as_stream(stream));
}

size_t mila_cdna4_gemm_fp8_workspace_bytes(int M, int K, int N)
{
    return (M > 0 && K > 0 && N > 0 && K % 16 == 0) ? gemm_fp8_ws_bytes(M, K, N) : 0;
}

int mila_cdna4_gemm_fp8_scaled_ws(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, const float* weight_scale, const uint16_t* bias, int M, int K, int N,
                                  void* workspace, size_t workspace_bytes, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X8 && W8 && x_scales && weight_scale, "gemm_fp8_scaled_ws: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_fp8_applicable(M, K, N), "gemm_fp8_scaled_ws: shape (M=%d, K=%d, N=%d) has no fp8 MFMA kernel (ask gemm_fp8_applicable)", M, K, N);
    const size_t need = gemm_fp8_ws_bytes(M, K, N);
    if (need && (!workspace || workspace_bytes < need))
        return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_fp8_scaled_ws: workspace %zu bytes < required %zu (ask gemm_fp8_workspace_bytes)", workspace_bytes, need);
    MILA_REQUIRE(!need || (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "gemm_fp8_scaled_ws: the workspace must be 16-byte aligned");
    return launch_gemm_fp8_ws(Y, X8, W8, x_scales, Fp8WScale{weight_scale, 0}, bias, M, K, N, as_stream(stream), workspace);
}

// [e4m3 weights | e4m3 activations | per-token scales | split-K workspace], each part 16-byte aligned
static size_t w4a8_ws_offset(int M, int K, int N)
{
    const size_t w = ((size_t)N * K + 15) & ~(size_t)15, x = ((size_t)M * K + 15) & ~(size_t)15, t = ((size_t)M * 4 + 15) & ~(size_t)15;
    return w + x + t;
}
size_t mila_cdna4_gemm_w4a8_scratch_bytes(int M, int K, int N)
{
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return w4a8_ws_offset(M, K, N) + (K % 16 == 0 ? gemm_fp8_ws_bytes(M, K, N) : 0);
}

int mila_cdna4_gemm_bf16_w4a8(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed, const float* scales, const float* weight_fp8_scale,
                              const uint16_t* bias, int M, int K, int N, int group, void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && W_packed && scales && weight_fp8_scale, "gemm_bf16_w4a8: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_fp8_applicable(M, K, N), "gemm_bf16_w4a8: shape (M=%d, K=%d, N=%d) has no fp8 MFMA kernel (ask gemm_fp8_applicable)", M, K, N);
    const size_t need = mila_cdna4_gemm_w4a8_scratch_bytes(M, K, N);
    if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_bf16_w4a8: scratch %zu bytes < required %zu", scratch_bytes, need);
    uint8_t* w8 = static_cast<uint8_t*>(scratch);
    uint8_t* x8 = w8 + (((size_t)N * K + 15) & ~(size_t)15);
    float* ts = reinterpret_cast<float*>(x8 + (((size_t)M * K + 15) & ~(size_t)15));
    int rc = mila_cdna4_upcast_fp4_to_fp8(w8, W_packed, scales, weight_fp8_scale, N, K, group, stream);
    if (rc) return rc;
    rc = mila_cdna4_quantize_fp8_per_token(x8, ts, X, M, K, stream);
    if (rc) return rc;
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "gemm_bf16_w4a8: the scratch must be 16-byte aligned");
    return launch_gemm_fp8_ws(Y, x8, w8, ts, Fp8WScale{weight_fp8_scale, 0}, bias, M, K, N, as_stream(stream), w8 + w4a8_ws_offset(M, K, N));
}

// (0 also where the plain W4A8 GEMM over [2F, K] would split K given a workspace: Linear + GeGLU as two calls is the faster pair there and keeps the bits of the unfused path)
int mila_cdna4_gemm_geglu_w4a8_applicable(int M, int K, int F) { return (M > 0 && K > 0 && F > 0 && K % 16 == 0 && !gemm_fp8_geglu_steps_aside(M, K, F)) ? 1 : 0; }

int mila_cdna4_gemm_geglu_fp8_scaled(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, const float* weight_scale, int M, int K, int F,
                                     mila_stream_t stream)
{
    MILA_REQUIRE(Y && X8 && W8 && x_scales && weight_scale, "gemm_geglu_fp8_scaled: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_geglu_w4a8_applicable(M, K, F), "gemm_geglu_fp8_scaled: shape (M=%d, K=%d, F=%d) is outside the fused kernel", M, K, F);
    return launch_gemm_fp8_geglu(Y, X8, W8, x_scales, Fp8WScale{weight_scale, 0}, M, K, F, as_stream(stream));
}

int mila_cdna4_gemm_geglu_bf16_w4a8(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed, const float* scales, const float* weight_fp8_scale,
                                    int M, int K, int F, int group, void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && W_packed && scales && weight_fp8_scale, "gemm_geglu_bf16_w4a8: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_geglu_w4a8_applicable(M, K, F), "gemm_geglu_bf16_w4a8: shape (M=%d, K=%d, F=%d) is outside the fused kernel", M, K, F);
    const int N = 2 * F;
    const size_t need = mila_cdna4_gemm_w4a8_scratch_bytes(M, K, N);
    if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_geglu_bf16_w4a8: scratch %zu bytes < required %zu", scratch_bytes, need);
    uint8_t* w8 = static_cast<uint8_t*>(scratch);
    uint8_t* x8 = w8 + (((size_t)N * K + 15) & ~(size_t)15);
    float* ts = reinterpret_cast<float*>(x8 + (((size_t)M * K + 15) & ~(size_t)15));
    int rc = mila_cdna4_upcast_fp4_to_fp8(w8, W_packed, scales, weight_fp8_scale, N, K, group, stream);
    if (rc) return rc;
    rc = mila_cdna4_quantize_fp8_per_token(x8, ts, X, M, K, stream);
    if (rc) return rc;
    return launch_gemm_fp8_geglu(Y, x8, w8, ts, Fp8WScale{weight_fp8_scale, 0}, M, K, F, as_stream(stream));
}

/* ---- W8A8: the PerChannelFp8<> policy's own e4m3 [N, K] weights + scale[N] consumed by the fp8 matrix cores as they lie in HBM (Quantization/Weight/Policies.ixx:39-40:
 * "FP8 matmul consumes weights and scales natively -- no dequantization on the forward hot path"); activations per token as on the W4A8 path
 * (Fp8Prefill/CudaFp8Prefill.cu:108-160).  The same kernels as the W4A8 forms above with the weight scale as a per-channel vector:
 *   y = bf16( (sum_k X8 W8) * scale[n] * s_m + bias )      (fp32, one rounding; common.h: w8a8_scale_bias)
 * Opt-in: the reference's arithmetic for this policy is W8A16 (CudaLinearOp.ixx:597-644), which stays the default of RocmLinearOp. ---- */
int mila_cdna4_gemm_fp8_w8a8_ws(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, const float* channel_scales, const uint16_t* bias, int M, int K, int N,
                                void* workspace, size_t workspace_bytes, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X8 && W8 && x_scales && channel_scales, "gemm_fp8_w8a8_ws: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_fp8_applicable(M, K, N), "gemm_fp8_w8a8_ws: shape (M=%d, K=%d, N=%d) has no fp8 MFMA kernel (ask gemm_fp8_applicable)", M, K, N);
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(channel_scales) & 15) == 0, "gemm_fp8_w8a8_ws: the per-channel scales must be 16-byte aligned");
    const size_t need = gemm_fp8_ws_bytes(M, K, N);
    if (need && (!workspace || workspace_bytes < need))
        return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_fp8_w8a8_ws: workspace %zu bytes < required %zu (ask gemm_fp8_workspace_bytes)", workspace_bytes, need);
    MILA_REQUIRE(!need || (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "gemm_fp8_w8a8_ws: the workspace must be 16-byte aligned");
    return launch_gemm_fp8_ws(Y, X8, W8, x_scales, Fp8WScale{channel_scales, 1}, bias, M, K, N, as_stream(stream), workspace);
}

int mila_cdna4_gemm_geglu_fp8_w8a8(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, const float* channel_scales, int M, int K, int F, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X8 && W8 && x_scales && channel_scales, "gemm_geglu_fp8_w8a8: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_geglu_w4a8_applicable(M, K, F), "gemm_geglu_fp8_w8a8: shape (M=%d, K=%d, F=%d) is outside the fused kernel (ask gemm_geglu_w4a8_applicable)", M, K, F);
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(channel_scales) & 15) == 0 && F % 4 == 0, "gemm_geglu_fp8_w8a8: the per-channel scales must be 16-byte aligned and F a multiple of 4");
    return launch_gemm_fp8_geglu(Y, X8, W8, x_scales, Fp8WScale{channel_scales, 1}, M, K, F, as_stream(stream));
}

// [e4m3 activations | per-token scales | split-K workspace], each part 16-byte aligned
static size_t w8a8_ws_offset(int M, int K)
{
    return (((size_t)M * K + 15) & ~(size_t)15) + (((size_t)M * 4 + 15) & ~(size_t)15);
}
size_t mila_cdna4_gemm_w8a8_scratch_bytes(int M, int K, int N)
{
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return w8a8_ws_offset(M, K) + (K % 16 == 0 ? gemm_fp8_ws_bytes(M, K, N) : 0);
}

int mila_cdna4_gemm_bf16_w8a8(uint16_t* Y, const uint16_t* X, const uint8_t* W8, const float* channel_scales, const uint16_t* bias, int M, int K, int N, void* scratch,
                              size_t scratch_bytes, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && W8 && channel_scales, "gemm_bf16_w8a8: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_fp8_applicable(M, K, N), "gemm_bf16_w8a8: shape (M=%d, K=%d, N=%d) has no fp8 MFMA kernel (ask gemm_fp8_applicable)", M, K, N);
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(channel_scales) & 15) == 0, "gemm_bf16_w8a8: the per-channel scales must be 16-byte aligned");
    const size_t need = mila_cdna4_gemm_w8a8_scratch_bytes(M, K, N);
    if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_bf16_w8a8: scratch %zu bytes < required %zu", scratch_bytes, need);
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "gemm_bf16_w8a8: the scratch must be 16-byte aligned");
    uint8_t* x8 = static_cast<uint8_t*>(scratch);
    float* ts = reinterpret_cast<float*>(x8 + (((size_t)M * K + 15) & ~(size_t)15));
    int rc = mila_cdna4_quantize_fp8_per_token(x8, ts, X, M, K, stream);
    if (rc) return rc;
    return launch_gemm_fp8_ws(Y, x8, W8, ts, Fp8WScale{channel_scales, 1}, bias, M, K, N, as_stream(stream), x8 + w8a8_ws_offset(M, K));
}

int mila_cdna4_gemm_geglu_bf16_w8a8(uint16_t* Y, const uint16_t* X, const uint8_t* W8, const float* channel_scales, int M, int K, int F, void* scratch, size_t scratch_bytes,
                                    mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && W8 && channel_scales, "gemm_geglu_bf16_w8a8: null pointer");
    MILA_REQUIRE(mila_cdna4_gemm_geglu_w4a8_applicable(M, K, F), "gemm_geglu_bf16_w8a8: shape (M=%d, K=%d, F=%d) is outside the fused kernel (ask gemm_geglu_w4a8_applicable)", M, K, F);
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(channel_scales) & 15) == 0 && F % 4 == 0, "gemm_geglu_bf16_w8a8: the per-channel scales must be 16-byte aligned and F a multiple of 4");
    const size_t need = w8a8_ws_offset(M, K);
    if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_geglu_bf16_w8a8: scratch %zu bytes < required %zu", scratch_bytes, need);
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "gemm_geglu_bf16_w8a8: the scratch must be 16-byte aligned");
    uint8_t* x8 = static_cast<uint8_t*>(scratch);
    float* ts = reinterpret_cast<float*>(x8 + (((size_t)M * K + 15) & ~(size_t)15));
    int rc = mila_cdna4_quantize_fp8_per_token(x8, ts, X, M, K, stream);
    if (rc) return rc;
    return launch_gemm_fp8_geglu(Y, x8, W8, ts, Fp8WScale{channel_scales, 1}, M, K, F, as_stream(stream));
}

}  // extern "C"
