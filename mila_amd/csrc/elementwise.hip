// Element-wise and gather kernels: GELU, GeGLU, residual add, scale, dtype convert, split3,
// token-embedding gather, GPT-2 token+position embedding.  All HBM-bound: 16-byte accesses,
// grid capped at 2048 workgroups with a grid-stride loop.
//   gelu:      OPS/Activations/Gelu/Kernels/Gelu.Fp32.cu:29-40 (functor: ElementwiseActivation.h:41-50)
//   geglu:     OPS/Activations/Geglu/Kernels/Geglu.cu:42-61
//   residual:  OPS/Residual/Kernels/Residual.Bf16.cu:15-40
//   scale:     Compute/Devices/Cuda/Tensors/Operations/Kernels/Math.Elementwise.cu:106-113
//   split3:    Compute/Devices/Cuda/Tensors/Operations/Kernels/Structural.cu:106
//   embedding: OPS/Embeddings/Kernels/TokenEmbedding.Bf16.cu:23-97 (+ TokenEmbedding.ixx:179-181 scale)
//   lpe:       OPS/Encodings/Lpe/Kernels/Lpe.Fp32.cu:33-124; CPU/CpuEncoderOp.ixx:255-330
#include "common.h"
#include "internal.h"

namespace mila {

static inline int grid_for(int64_t work_items, int per_block)
{
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename F>
__device__ __forceinline__ u32x4 map8(const u32x4 v, F f)
{
    u32x4 r;
#pragma unroll
    for (int d = 0; d < 4; ++d) r[d] = pack_bf16x2(f(bf16_lo(v[d])), f(bf16_hi(v[d])));
    return r;
}

__global__ __launch_bounds__(256) void gelu_bf16_kernel(uint16_t* __restrict__ Y, const uint16_t* __restrict__ X, int64_t n)
{
    const int64_t nvec = n / 8, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride)
        st16(Y + i * 8, map8(ld16(X + i * 8), [](float x) { return gelu_tanh(x); }));
    for (int64_t i = nvec * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        Y[i] = f32_to_bf16_bits(gelu_tanh(bf16_bits_to_f32(X[i])));
}

__global__ __launch_bounds__(256) void gelu_fp32_kernel(float* __restrict__ Y, const float* __restrict__ X, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) Y[i] = gelu_tanh_precise(X[i]);
}

// row = [gate(0..half) | up(half..2half)]; half % 8 == 0
__global__ __launch_bounds__(256) void geglu_bf16_kernel(uint16_t* __restrict__ Y, const uint16_t* __restrict__ X,
                                                         int64_t total_vec, int half_vec)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += stride)
    {
        const int64_t t = i / half_vec, c = i % half_vec;
        const uint16_t* row = X + t * (int64_t)half_vec * 16;
        const u32x4 g = ld16(row + c * 8), u = ld16(row + ((int64_t)half_vec + c) * 8);
        u32x4 r;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            r[d] = pack_bf16x2(gelu_tanh(bf16_lo(g[d])) * bf16_lo(u[d]), gelu_tanh(bf16_hi(g[d])) * bf16_hi(u[d]));
        st16(Y + i * 8, r);
    }
}

__global__ __launch_bounds__(256) void residual_bf16_kernel(uint16_t* __restrict__ Y, const uint16_t* __restrict__ A,
                                                            const uint16_t* __restrict__ B, int64_t n)
{
    const int64_t nvec = n / 8, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride)
    {
        const u32x4 a = ld16(A + i * 8), b = ld16(B + i * 8);
        u32x4 r;
#pragma unroll
        for (int d = 0; d < 4; ++d) r[d] = pack_bf16x2(bf16_lo(a[d]) + bf16_lo(b[d]), bf16_hi(a[d]) + bf16_hi(b[d]));
        st16(Y + i * 8, r);
    }
    for (int64_t i = nvec * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        Y[i] = f32_to_bf16_bits(bf16_bits_to_f32(A[i]) + bf16_bits_to_f32(B[i]));
}

__global__ __launch_bounds__(256) void residual_fp32_kernel(float* __restrict__ Y, const float* __restrict__ A,
                                                            const float* __restrict__ B, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) Y[i] = A[i] + B[i];
}

__global__ __launch_bounds__(256) void scale_bf16_kernel(uint16_t* __restrict__ Y, const uint16_t* __restrict__ X,
                                                         int64_t n, float s)
{
    const int64_t nvec = n / 8, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride)
        st16(Y + i * 8, map8(ld16(X + i * 8), [s](float x) { return x * s; }));
    for (int64_t i = nvec * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        Y[i] = f32_to_bf16_bits(bf16_bits_to_f32(X[i]) * s);
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(uint16_t* __restrict__ Y, const float* __restrict__ X, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) Y[i] = f32_to_bf16_bits(X[i]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(float* __restrict__ Y, const uint16_t* __restrict__ X, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) Y[i] = bf16_bits_to_f32(X[i]);
}

// split the last dim (na+nb+nc, all multiples of 8) into three tensors
__global__ __launch_bounds__(256) void split3_bf16_kernel(uint16_t* __restrict__ a, uint16_t* __restrict__ b,
                                                          uint16_t* __restrict__ c, const uint16_t* __restrict__ X,
                                                          int64_t total_vec, int va, int vb, int vc)
{
    const int vw = va + vb + vc;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += stride)
    {
        const int64_t r = i / vw;
        const int col = (int)(i % vw);
        const u32x4 v = ld16(X + i * 8);
        if (col < va) st16(a + (r * va + col) * 8, v);
        else if (col < va + vb) st16(b + (r * vb + (col - va)) * 8, v);
        else st16(c + (r * vc + (col - va - vb)) * 8, v);
    }
}

// one workgroup per token row; token id read from device memory; C % 8 == 0
__global__ __launch_bounds__(256) void embedding_gather_bf16_kernel(uint16_t* __restrict__ Y,
                                                                    const int32_t* __restrict__ tokens,
                                                                    const uint16_t* __restrict__ table, int C,
                                                                    int vocab, float scale, int32_t* error_flag)
{
    const int t = blockIdx.x;
    const int tok = tokens[t];
    if (tok < 0 || tok >= vocab)
    {
        if (threadIdx.x == 0 && error_flag) atomicExch(error_flag, 1 + t);
        return;
    }
    const uint16_t* src = table + (size_t)tok * C;
    uint16_t* dst = Y + (size_t)t * C;
    for (int i = threadIdx.x; i < C / 8; i += 256)
    {
        u32x4 v = ld16(src + (size_t)i * 8);
        if (scale != 0.0f) v = map8(v, [scale](float x) { return x * scale; });
        st16(dst + (size_t)i * 8, v);
    }
}

// FP8 tied table (per-vocab-row scale): y = bf16(float(e4m3) * scale[row]), then the optional
// embedding scale as a second bf16 rounding (OPS/Embeddings/Kernels/TokenEmbedding.Fp8.cu:33-69 +
// TokenEmbedding.ixx:179-181)
__global__ __launch_bounds__(256) void embedding_gather_bf16_qfp8_kernel(uint16_t* __restrict__ Y,
                                                                         const int32_t* __restrict__ tokens,
                                                                         const uint8_t* __restrict__ table,
                                                                         const float* __restrict__ row_scales, int C,
                                                                         int vocab, float scale, int32_t* error_flag)
{
    const int t = blockIdx.x;
    const int tok = tokens[t];
    if (tok < 0 || tok >= vocab)
    {
        if (threadIdx.x == 0 && error_flag) atomicExch(error_flag, 1 + t);
        return;
    }
    const float rs = row_scales[tok];
    const uint8_t* src = table + (size_t)tok * C;
    uint16_t* dst = Y + (size_t)t * C;
    for (int i = threadIdx.x; i < C / 8; i += 256)
    {
        const u32x2 raw = *reinterpret_cast<const u32x2*>(src + (size_t)i * 8);
        u32x4 v;
#pragma unroll
        for (int d = 0; d < 2; ++d)
        {
            const f32x2 a = fp8x2_to_f32x2(raw[d], false), b = fp8x2_to_f32x2(raw[d], true);
            v[2 * d] = pack_bf16x2(a[0] * rs, a[1] * rs);
            v[2 * d + 1] = pack_bf16x2(b[0] * rs, b[1] * rs);
        }
        if (scale != 0.0f) v = map8(v, [scale](float x) { return x * scale; });
        st16(dst + (size_t)i * 8, v);
    }
}

// counter-based synthetic data: element i of a tensor seeded `seed` is
// bf16(offset + amp * (2u - 1)), u = top 24 bits of splitmix64(seed + i * golden) / 2^24.
// Reproducible on the host (tests/synth.py) without shipping files.
__device__ __forceinline__ float synth_uniform(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__global__ __launch_bounds__(256) void fill_uniform_bf16_kernel(uint16_t* __restrict__ dst, int64_t n, uint64_t seed,
                                                                float amp, float offset)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        dst[i] = f32_to_bf16_bits(__fadd_rn(offset, __fmul_rn(amp, __fadd_rn(__fmul_rn(2.0f, synth_uniform(seed, (uint64_t)i)), -1.0f))));  // no FMA contraction: host-reproducible
}

__global__ __launch_bounds__(256) void lpe_bf16_kernel(uint16_t* __restrict__ Y, const int32_t* __restrict__ tokens,
                                                       const uint16_t* __restrict__ wte,
                                                       const uint16_t* __restrict__ wpe, int T, int C,
                                                       int out_stride_T, int vocab, int32_t* error_flag)
{
    const int bt = blockIdx.x;
    const int b = bt / T, t = bt % T;
    const int tok = tokens[bt];
    if (tok < 0 || tok >= vocab)
    {
        if (threadIdx.x == 0 && error_flag) atomicExch(error_flag, 1 + bt);
        return;
    }
    const uint16_t* we = wte + (size_t)tok * C;
    const uint16_t* wp = wpe + (size_t)t * C;
    uint16_t* dst = Y + ((size_t)b * out_stride_T + t) * C;
    for (int i = threadIdx.x; i < C / 8; i += 256)
    {
        const u32x4 a = ld16(we + (size_t)i * 8), p = ld16(wp + (size_t)i * 8);
        u32x4 r;
#pragma unroll
        for (int d = 0; d < 4; ++d) r[d] = pack_bf16x2(bf16_lo(a[d]) + bf16_lo(p[d]), bf16_hi(a[d]) + bf16_hi(p[d]));
        st16(dst + (size_t)i * 8, r);
    }
}

// ---- test / measurement hooks -------------------------------------------------------------------
__global__ void selftest_decode_kernel(float* out_fp8, float* out_fp4)
{
    const uint32_t b = threadIdx.x;   // 256 threads
    // fp8: byte value b placed in each of the 4 byte positions of a dword
    {
        const uint32_t w0 = b | (0x38u << 8);            // bytes 0,1 (byte1 = 1.0 marker)
        const uint32_t w1 = (0x38u << 16) | (b << 24);   // bytes 2,3
        bf16x2 lo = fp8x2_to_bf16x2(w0, false);          // (byte0, byte1)
        bf16x2 hi = fp8x2_to_bf16x2(w1, true);           // (byte2, byte3)
        bf16x2 lo2 = fp8x2_to_bf16x2(b << 8, false);     // byte1
        bf16x2 hi2 = fp8x2_to_bf16x2(b << 16, true);     // byte2
        out_fp8[0 * 256 + b] = (float)lo[0];
        out_fp8[1 * 256 + b] = (float)lo2[1];
        out_fp8[2 * 256 + b] = (float)hi2[0];
        out_fp8[3 * 256 + b] = (float)hi[1];
    }
    {
        bf16x2 v0 = fp4x2_to_bf16x2<0>(b);
        bf16x2 v1 = fp4x2_to_bf16x2<1>(b << 8);
        bf16x2 v2 = fp4x2_to_bf16x2<2>(b << 16);
        bf16x2 v3 = fp4x2_to_bf16x2<3>(b << 24);
        out_fp4[0 * 512 + 2 * b] = (float)v0[0]; out_fp4[0 * 512 + 2 * b + 1] = (float)v0[1];
        out_fp4[1 * 512 + 2 * b] = (float)v1[0]; out_fp4[1 * 512 + 2 * b + 1] = (float)v1[1];
        out_fp4[2 * 512 + 2 * b] = (float)v2[0]; out_fp4[2 * 512 + 2 * b + 1] = (float)v2[1];
        out_fp4[3 * 512 + 2 * b] = (float)v3[0]; out_fp4[3 * 512 + 2 * b + 1] = (float)v3[1];
    }
}

// one v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales): C[16,16] = A[16,128] B[16,128]^T with the operand
// layout the fp8 GEMM assumes: lane l holds row l & 15, bytes k = 32 (l >> 4) .. + 31 (mode 0) or the two-half layout
// k = 16 (l >> 4) + j, 64 + 16 (l >> 4) + j (mode 1).  tests/test_linear_gpu.py checks which one the hardware implements.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void selftest_mfma_fp8_kernel(float* C, const uint8_t* A, const uint8_t* B, int mode)
{
    const int lane = threadIdx.x & 63, row = lane & 15, kg = lane >> 4;
    i32x8 a, b;
#pragma unroll
    for (int d = 0; d < 8; ++d)
    {
        uint32_t wa = 0, wb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int byte = 4 * d + j;
            const int k = (mode == 0) ? 32 * kg + byte : (byte < 16 ? 16 * kg + byte : 64 + 16 * kg + (byte - 16));
            wa |= (uint32_t)A[row * 128 + k] << (8 * j);
            wb |= (uint32_t)B[row * 128 + k] << (8 * j);
        }
        a[d] = (int)wa;
        b[d] = (int)wb;
    }
    f32x4 c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(4 * kg + r) * 16 + row] = c[r];      // D[i = 4 (l >> 4) + r][j = l & 15]
}

__global__ void selftest_wave_reduce_kernel(float* out, const float* in)
{
    const float v = in[threadIdx.x];
    out[threadIdx.x] = wave_sum(v);
    out[64 + threadIdx.x] = wave_max(v);
    out[128 + threadIdx.x] = wave_sum_shfl(v);
}

__global__ __launch_bounds__(256) void stream_copy_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int64_t nvec)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) dst[i] = src[i];
}
// The streaming yardstick in the matvec's own launch shape: one 16-wave workgroup per CU, every wave pulls contiguous 8 KiB pieces (eight 1-KiB
// non-temporal wave-loads in flight, as the distance-2 pipeline of matvec_body.h keeps), pieces dealt round-robin over all waves of the grid.
__global__ __launch_bounds__(1024) void stream_read_kernel(float* __restrict__ sink, const u32x4* __restrict__ src, int64_t nvec)
{
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 16;
    uint32_t acc = 0;
    int64_t base = wave * (64 * U);
    for (; base + 64 * U <= nvec; base += nwaves * (64 * U))
    {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld16_nt(src + base + u * 64 + lane);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    for (int64_t i = base + lane; i < nvec && i < base + 64 * U; i += 64)
    {
        const u32x4 a = ld16_nt(src + i);
        acc ^= a[0] ^ a[1] ^ a[2] ^ a[3];
    }
    if (acc == 0x12345679u) sink[0] = 1.0f;   // never true in practice; keeps the loads alive
}


// Warm the 256 MiB Infinity Cache with a byte range a LATER kernel will stream (the decode step's next weight matrices), from a
// side stream while the current kernel runs: one dword per 128-byte line (the line is what moves), default cache policy, lines
// walked in address order by the whole grid, which is the order the matvec kernels consume their rows in.  Nothing is written.
__global__ __launch_bounds__(256) void prefetch_l3_kernel(float* __restrict__ sink, const uint8_t* __restrict__ src, int64_t nlines)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint32_t acc = 0;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < nlines; i += 8 * stride)
    {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint32_t*>(src + (i + u * stride) * 128);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    for (; i < nlines; i += stride) acc ^= *reinterpret_cast<const uint32_t*>(src + i * 128);
    if (acc == 0x12345679u && sink != nullptr) sink[0] = 1.0f;   // keeps the loads alive; sink is never written in practice
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_gelu_bf16(uint16_t* Y, const uint16_t* X, int64_t n, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && n >= 0, "gelu_bf16: bad arguments");
    if (n == 0) return MILA_OK;
    hipLaunchKernelGGL(gelu_bf16_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, as_stream(stream), Y, X, n);
    MILA_LAUNCH_CHECK("gelu_bf16");
}

int mila_cdna4_gelu_fp32(float* Y, const float* X, int64_t n, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && n >= 0, "gelu_fp32: bad arguments");
    if (n == 0) return MILA_OK;
    hipLaunchKernelGGL(gelu_fp32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), Y, X, n);
    MILA_LAUNCH_CHECK("gelu_fp32");
}

int mila_cdna4_geglu_bf16(uint16_t* Y, const uint16_t* X, int tokens, int half, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X, "geglu_bf16: null pointer");
    MILA_REQUIRE(tokens > 0 && half > 0, "geglu_bf16: tokens/half must be positive (%d,%d)", tokens, half);
    MILA_REQUIRE(half % 8 == 0, "geglu_bf16: half width %d must be a multiple of 8", half);
    const int64_t total_vec = (int64_t)tokens * (half / 8);
    hipLaunchKernelGGL(geglu_bf16_kernel, dim3(grid_for(total_vec, 256)), dim3(256), 0, as_stream(stream), Y, X,
                       total_vec, half / 8);
    MILA_LAUNCH_CHECK("geglu_bf16");
}

int mila_cdna4_residual_bf16(uint16_t* Y, const uint16_t* A, const uint16_t* B, int64_t n, mila_stream_t stream)
{
    MILA_REQUIRE(Y && A && B && n >= 0, "residual_bf16: bad arguments");
    if (n == 0) return MILA_OK;
    hipLaunchKernelGGL(residual_bf16_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, as_stream(stream), Y, A, B, n);
    MILA_LAUNCH_CHECK("residual_bf16");
}

int mila_cdna4_residual_fp32(float* Y, const float* A, const float* B, int64_t n, mila_stream_t stream)
{
    MILA_REQUIRE(Y && A && B && n >= 0, "residual_fp32: bad arguments");
    if (n == 0) return MILA_OK;
    hipLaunchKernelGGL(residual_fp32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), Y, A, B, n);
    MILA_LAUNCH_CHECK("residual_fp32");
}

int mila_cdna4_scale_bf16(uint16_t* Y, const uint16_t* X, int64_t n, float s, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && n >= 0, "scale_bf16: bad arguments");
    if (n == 0) return MILA_OK;
    hipLaunchKernelGGL(scale_bf16_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, as_stream(stream), Y, X, n, s);
    MILA_LAUNCH_CHECK("scale_bf16");
}

int mila_cdna4_convert_f32_to_bf16(uint16_t* Y, const float* X, int64_t n, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && n >= 0, "convert_f32_to_bf16: bad arguments");
    if (n == 0) return MILA_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), Y, X, n);
    MILA_LAUNCH_CHECK("convert_f32_to_bf16");
}

int mila_cdna4_convert_bf16_to_f32(float* Y, const uint16_t* X, int64_t n, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && n >= 0, "convert_bf16_to_f32: bad arguments");
    if (n == 0) return MILA_OK;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), Y, X, n);
    MILA_LAUNCH_CHECK("convert_bf16_to_f32");
}

int mila_cdna4_split3_bf16(uint16_t* a, uint16_t* b, uint16_t* c, const uint16_t* X, int rows, int na, int nb, int nc,
                           mila_stream_t stream)
{
    MILA_REQUIRE(a && b && X && (c || nc == 0), "split3_bf16: null pointer");
    MILA_REQUIRE(rows > 0 && na > 0 && nb > 0 && nc >= 0, "split3_bf16: bad sizes");
    MILA_REQUIRE(na % 8 == 0 && nb % 8 == 0 && nc % 8 == 0, "split3_bf16: widths must be multiples of 8 (%d,%d,%d)", na, nb, nc);
    const int64_t total_vec = (int64_t)rows * ((na + nb + nc) / 8);
    hipLaunchKernelGGL(split3_bf16_kernel, dim3(grid_for(total_vec, 256)), dim3(256), 0, as_stream(stream), a, b, c, X,
                       total_vec, na / 8, nb / 8, nc / 8);
    MILA_LAUNCH_CHECK("split3_bf16");
}

int mila_cdna4_embedding_gather_bf16(uint16_t* Y, const int32_t* tokens, const uint16_t* table, int n_tok, int C,
                                     int vocab, float scale, int32_t* error_flag, mila_stream_t stream)
{
    MILA_REQUIRE(Y && tokens && table, "embedding_gather_bf16: null pointer");
    MILA_REQUIRE(n_tok > 0 && C > 0 && vocab > 0, "embedding_gather_bf16: bad sizes");
    MILA_REQUIRE(C % 8 == 0, "embedding_gather_bf16: C=%d must be a multiple of 8", C);
    hipLaunchKernelGGL(embedding_gather_bf16_kernel, dim3(n_tok), dim3(256), 0, as_stream(stream), Y, tokens, table, C,
                       vocab, scale, error_flag);
    MILA_LAUNCH_CHECK("embedding_gather_bf16");
}

int mila_cdna4_embedding_gather_bf16_qfp8(uint16_t* Y, const int32_t* tokens, const uint8_t* table, const float* row_scales,
                                          int n_tok, int C, int vocab, float scale, int32_t* error_flag, mila_stream_t stream)
{
    MILA_REQUIRE(Y && tokens && table && row_scales, "embedding_gather_bf16_qfp8: null pointer");
    MILA_REQUIRE(n_tok > 0 && C > 0 && vocab > 0, "embedding_gather_bf16_qfp8: bad sizes");
    MILA_REQUIRE(C % 8 == 0, "embedding_gather_bf16_qfp8: C=%d must be a multiple of 8", C);
    hipLaunchKernelGGL(embedding_gather_bf16_qfp8_kernel, dim3(n_tok), dim3(256), 0, as_stream(stream), Y, tokens, table,
                       row_scales, C, vocab, scale, error_flag);
    MILA_LAUNCH_CHECK("embedding_gather_bf16_qfp8");
}

int mila_cdna4_fill_uniform_bf16(uint16_t* dst, int64_t n, uint64_t seed, float amp, float offset, mila_stream_t stream)
{
    MILA_REQUIRE(dst && n >= 0, "fill_uniform_bf16: bad arguments");
    if (n == 0) return MILA_OK;
    hipLaunchKernelGGL(fill_uniform_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), dst, n, seed, amp,
                       offset);
    MILA_LAUNCH_CHECK("fill_uniform_bf16");
}

int mila_cdna4_lpe_bf16(uint16_t* Y, const int32_t* tokens, const uint16_t* wte, const uint16_t* wpe, int B, int T,
                        int C, int out_stride_T, int vocab, int32_t* error_flag, mila_stream_t stream)
{
    MILA_REQUIRE(Y && tokens && wte && wpe, "lpe_bf16: null pointer");
    MILA_REQUIRE(B > 0 && T > 0 && C > 0 && vocab > 0, "lpe_bf16: bad sizes");
    MILA_REQUIRE(out_stride_T >= T, "lpe_bf16: output row stride %d is shorter than T=%d", out_stride_T, T);
    MILA_REQUIRE(C % 8 == 0, "lpe_bf16: C=%d must be a multiple of 8", C);
    hipLaunchKernelGGL(lpe_bf16_kernel, dim3(B * T), dim3(256), 0, as_stream(stream), Y, tokens, wte, wpe, T, C,
                       out_stride_T, vocab, error_flag);
    MILA_LAUNCH_CHECK("lpe_bf16");
}

int mila_cdna4_selftest_decode(float* out_fp8, float* out_fp4, mila_stream_t stream)
{
    MILA_REQUIRE(out_fp8 && out_fp4, "selftest_decode: null pointer");
    hipLaunchKernelGGL(selftest_decode_kernel, dim3(1), dim3(256), 0, as_stream(stream), out_fp8, out_fp4);
    MILA_LAUNCH_CHECK("selftest_decode");
}

int mila_cdna4_selftest_mfma_fp8(float* C, const uint8_t* A, const uint8_t* B, int mode, mila_stream_t stream)
{
    MILA_REQUIRE(C && A && B, "selftest_mfma_fp8: null pointer");
    hipLaunchKernelGGL(selftest_mfma_fp8_kernel, dim3(1), dim3(64), 0, as_stream(stream), C, A, B, mode);
    MILA_LAUNCH_CHECK("selftest_mfma_fp8");
}

int mila_cdna4_selftest_wave_reduce(float* out, const float* in, mila_stream_t stream)
{
    MILA_REQUIRE(out && in, "selftest_wave_reduce: null pointer");
    hipLaunchKernelGGL(selftest_wave_reduce_kernel, dim3(1), dim3(64), 0, as_stream(stream), out, in);
    MILA_LAUNCH_CHECK("selftest_wave_reduce");
}

int mila_cdna4_stream_copy(void* dst, const void* src, size_t bytes, mila_stream_t stream)
{
    MILA_REQUIRE(dst && src && bytes % 16 == 0, "stream_copy: bad arguments");
    hipLaunchKernelGGL(stream_copy_kernel, dim3(2048), dim3(256), 0, as_stream(stream), (u32x4*)dst, (const u32x4*)src,
                       (int64_t)(bytes / 16));
    MILA_LAUNCH_CHECK("stream_copy");
}

int mila_cdna4_stream_read(float* sink, const void* src, size_t bytes, mila_stream_t stream)
{
    MILA_REQUIRE(sink && src && bytes % 16 == 0, "stream_read: bad arguments");
    hipLaunchKernelGGL(stream_read_kernel, dim3(kNumCU), dim3(1024), 0, as_stream(stream), sink, (const u32x4*)src,
                       (int64_t)(bytes / 16));
    MILA_LAUNCH_CHECK("stream_read");
}

int mila_cdna4_prefetch_l3(const void* src, size_t bytes, int workgroups, float* sink, mila_stream_t stream)
{
    MILA_REQUIRE(src && sink && workgroups > 0 && workgroups <= 4096, "prefetch_l3: bad arguments");
    const int64_t nlines = (int64_t)(bytes / 128);
    if (nlines == 0) return 0;
    hipLaunchKernelGGL(prefetch_l3_kernel, dim3(workgroups), dim3(256), 0, as_stream(stream), sink, (const uint8_t*)src, nlines);
    MILA_LAUNCH_CHECK("prefetch_l3");
}

}  // extern "C"
