// RMSNorm / LayerNorm / Softmax.
//   rmsnorm:   OPS/Normalizations/RmsNorm/Kernels/RmsNorm.Bf16.cu:20-73 (warp per slice) -> here
//              a 256-thread workgroup per long row (model-width rows), a wave per short row
//              (per-head rows), 16-byte loads; strided (inner > 1) slices take the generic kernel.
//   layernorm: CPU/CpuLayerNormOp.ixx:187-258 semantics (biased variance, two-pass), fp32 math.
//   softmax:   CPU/CpuSoftmaxOp.ixx:167-215 / OPS/Normalizations/Softmax/Kernels/Softmax.Fp32.cu:56-88
//              (thread per row there; a wave per row here, online max+sum in one pass).
#include "common.h"
#include "rms_common.h"

namespace mila {

// ---- RMSNorm ----------------------------------------------------------------------------------
constexpr int kRmsBlockMaxGroups = 256;   // rows up to 131072 elements through the workgroup-per-row kernel

__global__ __launch_bounds__(256) void rmsnorm_block_kernel(uint16_t* __restrict__ Y, uint16_t* __restrict__ rstd_out,
                                                            const uint16_t* __restrict__ X,
                                                            const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ b, int dim, float eps,
                                                            float w_offset)
{
    __shared__ float red[kRmsBlockMaxGroups];
    const size_t row = blockIdx.x;
    const uint16_t* x = X + row * dim;
    uint16_t* y = Y + row * dim;
    const float rstd = rms_rstd_block<4>(x, dim, eps, red);
    if (threadIdx.x == 0 && rstd_out) rstd_out[row] = f32_to_bf16_bits(rstd);
    for (int i = threadIdx.x; i < dim / 8; i += 256)
    {
        const u32x4 xv = ld16(x + (size_t)i * 8);
        u32x4 r;
        if (w == nullptr) r = rms_apply8_now(xv, rstd);
        else if (b == nullptr) r = rms_apply8(xv, ld16(w + (size_t)i * 8), rstd, w_offset);
        else r = rms_apply8_bias(xv, ld16(w + (size_t)i * 8), ld16(b + (size_t)i * 8), rstd, w_offset);
        st16(y + (size_t)i * 8, r);
    }
}

// Sandwich tail over T rows (prefill form of the decode matvec prologue): per row
//   r  = bf16(bf16(res + bf16(rmsnorm(a; post_w))) * post_scale)          -> R
//   xn = bf16(rmsnorm(r; next_w))                                         -> XN (when next_w != NULL)
// replaces RmsNorm + Residual (+ scale) + RmsNorm of Gemma.Block.ixx:339-356 / :287-289; same canonical reductions
// (rms_rstd_block) and element functions, so the outputs carry the same bits as the unfused launches.
// One workgroup per row, dim <= 8 * 256 * kTailChunks.
constexpr int kTailChunks = 4;
// NCH = 16-byte chunks per thread actually needed (ceil(dim / 8 / 256)): a row of 3840 takes 2, not kTailChunks -- the groups a
// smaller NCH leaves out are exactly the empty ones (g >= G), so the sums and their order do not change
// QUANT: XN's row is also written as per-token e4m3 (XQ) with its scale (XS) -- the arithmetic of quantize_fp8_per_token_kernel (csrc/quantize.hip) on the values
// XN receives (the maximum is exact in any order, the scale and the converts are the same expressions), so the W4A8 Linear that consumes XN needs no
// quantization launch of its own.
template <int NCH, bool QUANT = false>
__global__ __launch_bounds__(256) void tail_norm_kernel(uint16_t* __restrict__ R, uint16_t* __restrict__ XN,
                                                        const uint16_t* __restrict__ A, const uint16_t* __restrict__ RES,
                                                        const uint16_t* __restrict__ post_w, const uint16_t* __restrict__ next_w,
                                                        int dim, float post_scale, float eps, uint8_t* __restrict__ XQ = nullptr,
                                                        float* __restrict__ XS = nullptr)
{
    __shared__ float red_a[kRmsBlockMaxGroups], red_b[kRmsBlockMaxGroups];
    __shared__ float red_q[4];
    const size_t row = blockIdx.x;
    const uint16_t* a = A + row * dim;
    const uint16_t* res = RES + row * dim;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int nx16 = dim / 8, G = (nx16 + 63) / 64;
    // chunk c = 64 g + lane of group g = wib + 4 k: the assignment rms_rstd_block<4> uses
    // every operand of the row is requested up front: the residual and the two weight rows arrive under the first reduction instead of costing a second
    // memory round trip behind its barrier
    u32x4 av[NCH], rv[NCH], resv[NCH], pwv[NCH], nwv[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
    {
        const int c = 64 * (wib + 4 * k) + lane;
        const size_t e = (size_t)min(c, nx16 - 1) * 8;
        av[k] = ld16(a + e);
        resv[k] = ld16(res + e);
        pwv[k] = ld16(post_w + e);
        nwv[k] = next_w != nullptr ? ld16(next_w + e) : u32x4{0u, 0u, 0u, 0u};
    }
    {
#pragma unroll
        for (int k = 0; k < NCH; ++k)
        {
            const int g = wib + 4 * k, c = 64 * g + lane;
            float s = c < nx16 ? sumsq8(av[k], 0.0f) : 0.0f;
            s = wave_sum(s);
            if (lane == 0 && g < G) red_a[g] = s;
        }
        __syncthreads();
    }
    float t = 0.0f;
    for (int g = 0; g < G; ++g) t += red_a[g];
    const float rstd_a = rsqrtf(t / (float)dim + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
    {
        const int g = wib + 4 * k, c = 64 * g + lane;
        const size_t e = (size_t)min(c, nx16 - 1) * 8;
        rv[k] = sandwich_tail8(rms_apply8(av[k], pwv[k], rstd_a, 0.0f), resv[k], post_scale);
        if (c < nx16) st16(R + row * dim + e, rv[k]);
        if (next_w != nullptr)
        {
            float s = c < nx16 ? sumsq8(rv[k], 0.0f) : 0.0f;
            s = wave_sum(s);
            if (lane == 0 && g < G) red_b[g] = s;
        }
    }
    if (next_w == nullptr) return;
    __syncthreads();
    float t2 = 0.0f;
    for (int g = 0; g < G; ++g) t2 += red_b[g];
    const float rstd_r = rsqrtf(t2 / (float)dim + eps);
    u32x4 xn[NCH];
    float qm = 0.0f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
    {
        const int c = 64 * (wib + 4 * k) + lane;
        xn[k] = rms_apply8(rv[k], nwv[k], rstd_r, 0.0f);
        if (c < nx16)
        {
            st16(XN + row * dim + (size_t)c * 8, xn[k]);
            if constexpr (QUANT)
            {
#pragma unroll
                for (int e = 0; e < 4; ++e) qm = fmaxf(qm, fmaxf(fabsf(bf16_lo(xn[k][e])), fabsf(bf16_hi(xn[k][e]))));
            }
        }
    }
    if constexpr (QUANT)
    {
        qm = block_max<4>(qm, red_q);
        const float scale = fmaxf(qm, 1e-12f) / 448.0f;
        if (threadIdx.x == 0) XS[row] = scale;
        const float inv = 1.0f / scale;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
        {
            const int c = 64 * (wib + 4 * k) + lane;
            if (c < nx16)
            {
                u32x2 o;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                {
                    int r = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_lo(xn[k][2 * h]) * inv, bf16_hi(xn[k][2 * h]) * inv, 0, false);
                    r = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_lo(xn[k][2 * h + 1]) * inv, bf16_hi(xn[k][2 * h + 1]) * inv, r, true);
                    o[h] = (uint32_t)r;
                }
                *reinterpret_cast<u32x2*>(XQ + row * dim + (size_t)c * 8) = o;
            }
        }
    }
}

// wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void rmsnorm_wave_kernel(uint16_t* __restrict__ Y, uint16_t* __restrict__ rstd_out,
                                                           const uint16_t* __restrict__ X,
                                                           const uint16_t* __restrict__ w,
                                                           const uint16_t* __restrict__ b, int rows, int dim,
                                                           float eps, float w_offset)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const uint16_t* x = X + (size_t)row * dim;
    uint16_t* y = Y + (size_t)row * dim;
    const float rstd = rms_rstd_wave(x, dim, eps);
    if (lane == 0 && rstd_out) rstd_out[row] = f32_to_bf16_bits(rstd);
    for (int i = lane; i < dim / 8; i += 64)
    {
        const u32x4 xv = ld16(x + (size_t)i * 8);
        u32x4 r;
        if (w == nullptr) r = rms_apply8_now(xv, rstd);
        else if (b == nullptr) r = rms_apply8(xv, ld16(w + (size_t)i * 8), rstd, w_offset);
        else r = rms_apply8_bias(xv, ld16(w + (size_t)i * 8), ld16(b + (size_t)i * 8), rstd, w_offset);
        st16(y + (size_t)i * 8, r);
    }
}

// generic: slice = (outer, inner), elements strided by `inner`; wave per slice
__global__ __launch_bounds__(256) void rmsnorm_strided_kernel(uint16_t* __restrict__ Y, uint16_t* __restrict__ rstd_out,
                                                              const uint16_t* __restrict__ X,
                                                              const uint16_t* __restrict__ w,
                                                              const uint16_t* __restrict__ b, int slices, int dim,
                                                              int inner, float eps, float w_offset)
{
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= slices) return;
    const int lane = threadIdx.x & 63;
    const size_t base = (size_t)(idx / inner) * dim * inner + (idx % inner);
    float ss = 0.0f;
    for (int i = lane; i < dim; i += 64)
    {
        const float v = bf16_bits_to_f32(X[base + (size_t)i * inner]);
        ss = fmaf(v, v, ss);
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)dim + eps);
    if (lane == 0 && rstd_out) rstd_out[idx] = f32_to_bf16_bits(rstd);
    for (int i = lane; i < dim; i += 64)
    {
        const float xv = bf16_bits_to_f32(X[base + (size_t)i * inner]);
        const float ww = w ? bf16_bits_to_f32(w[i]) : 1.0f;
        const float bb = b ? bf16_bits_to_f32(b[i]) : 0.0f;
        Y[base + (size_t)i * inner] = f32_to_bf16_bits(rms_apply1(xv, ww, rstd, w ? w_offset : 0.0f, bb));
    }
}

// ---- LayerNorm (row contiguous); T = float or bf16 bits ----------------------------------------
template <typename T> __device__ __forceinline__ float load_f(const T* p, size_t i);
template <> __device__ __forceinline__ float load_f<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float load_f<uint16_t>(const uint16_t* p, size_t i) { return bf16_bits_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void store_f(T* p, size_t i, float v);
template <> __device__ __forceinline__ void store_f<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void store_f<uint16_t>(uint16_t* p, size_t i, float v) { p[i] = f32_to_bf16_bits(v); }

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(T* __restrict__ Y, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out, const T* __restrict__ X,
                                                        const T* __restrict__ w, const T* __restrict__ b, int dim,
                                                        float eps)
{
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const T* x = X + row * dim;
    T* y = Y + row * dim;
    float s = 0.0f;
    for (int i = threadIdx.x; i < dim; i += 256) s += load_f(x, i);
    const float mean = block_sum<4>(s, red) / (float)dim;
    float v = 0.0f;
    for (int i = threadIdx.x; i < dim; i += 256)
    {
        const float d = load_f(x, i) - mean;
        v = fmaf(d, d, v);
    }
    const float var = block_sum<4>(v, red) / (float)dim;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0)
    {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    for (int i = threadIdx.x; i < dim; i += 256)
    {
        float n = rstd * (load_f(x, i) - mean);
        if (w) n *= load_f(w, i);
        if (b) n += load_f(b, i);
        store_f(y, i, n);
    }
}

// bf16 rows of up to 4096 elements: one WAVE per row (four rows per workgroup), the row held in registers between the three passes (16-byte loads, one global
// read and one write per element), 64-lane butterflies instead of two workgroup barriers per reduction.  GPT-2's [8192, 768] rows: 20 us -> ~7 us per call
// (25 calls per forward).  Same two-pass arithmetic (mean, then the biased variance of the deviations) as layernorm_kernel; only the order of the fp32 sums differs.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_wave_bf16_kernel(uint16_t* __restrict__ Y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                  const uint16_t* __restrict__ X, const uint16_t* __restrict__ w, const uint16_t* __restrict__ b,
                                                                  int rows, int dim, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* x = X + (size_t)row * dim;
    const int nvec = dim / 8;
    u32x4 v[NV];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
    {
        const int c = lane + 64 * k;
        v[k] = c < nvec ? ld16(x + (size_t)c * 8) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) s += bf16_lo(v[k][e]) + bf16_hi(v[k][e]);
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane + 64 * k < nvec)
        {
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                const float d0 = bf16_lo(v[k][e]) - mean, d1 = bf16_hi(v[k][e]) - mean;
                q = fmaf(d0, d0, q);
                q = fmaf(d1, d1, q);
            }
        }
    const float var = wave_sum(q) / (float)dim;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0)
    {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    uint16_t* y = Y + (size_t)row * dim;
#pragma unroll
    for (int k = 0; k < NV; ++k)
    {
        const int c = lane + 64 * k;
        if (c >= nvec) continue;
        u32x4 wv = u32x4{0u, 0u, 0u, 0u}, bv = wv, o;
        if (w) wv = ld16(w + (size_t)c * 8);
        if (b) bv = ld16(b + (size_t)c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            float n0 = rstd * (bf16_lo(v[k][e]) - mean), n1 = rstd * (bf16_hi(v[k][e]) - mean);
            if (w) { n0 *= bf16_lo(wv[e]); n1 *= bf16_hi(wv[e]); }
            if (b) { n0 += bf16_lo(bv[e]); n1 += bf16_hi(bv[e]); }
            o[e] = pack_bf16x2(n0, n1);
        }
        st16(y + (size_t)c * 8, o);
    }
}

// ---- Softmax along `dim` with stride `inner`; wave per slice -----------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(T* __restrict__ Y, const T* __restrict__ X, int slices, int dim,
                                                      int inner)
{
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= slices) return;
    const int lane = threadIdx.x & 63;
    const size_t base = (size_t)(idx / inner) * dim * inner + (idx % inner);
    float m = -INFINITY;
    for (int i = lane; i < dim; i += 64) m = fmaxf(m, load_f(X, base + (size_t)i * inner));
    m = wave_max(m);
    float s = 0.0f;
    for (int i = lane; i < dim; i += 64) s += expf(load_f(X, base + (size_t)i * inner) - m);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int i = lane; i < dim; i += 64)
        store_f(Y, base + (size_t)i * inner, expf(load_f(X, base + (size_t)i * inner) - m) * inv);
}

// ---- RMSNorm, fp32 row (RmsNorm.Fp32.cu:20-86: a warp per slice there, a 64-lane wave per slice here) --------------------------
// y = (x * rstd) * (w + w_offset) + b with rstd = rsqrtf(sum x^2 * (1 / dim) + eps); slices of a [outer, dim, inner] tensor, element i of
// slice (o, j) at ((o * dim) + i) * inner + j.  VEC: inner == 1 and dim % 4 == 0 -> 16-byte accesses.
template <bool VEC>
__global__ __launch_bounds__(256) void rmsnorm_fp32_kernel(float* __restrict__ Y, float* __restrict__ rstd_out, const float* __restrict__ X,
                                                           const float* __restrict__ w, const float* __restrict__ b, int slices, int dim, int inner,
                                                           float eps, float w_offset)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= slices) return;
    const int o = idx / inner, j = idx - o * inner;
    const float* x = X + (size_t)o * dim * inner + j;
    float* y = Y + (size_t)o * dim * inner + j;
    float m2 = 0.0f;
    if constexpr (VEC)
    {
        for (int i = lane; i < dim / 4; i += 64)
        {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)i * 4);
            m2 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
    }
    else
    {
        for (int i = lane; i < dim; i += 64)
        {
            const float v = x[(size_t)i * inner];
            m2 += v * v;
        }
    }
    m2 = wave_sum(m2);
    const float rs = rsqrtf(m2 * (1.0f / (float)dim) + eps);
    if (lane == 0 && rstd_out) rstd_out[idx] = rs;
    if constexpr (VEC)
    {
        for (int i = lane; i < dim / 4; i += 64)
        {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)i * 4);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                const float wv = w ? w[i * 4 + e] + w_offset : 1.0f;
                const float bv = b ? b[i * 4 + e] : 0.0f;
                r[e] = (v[e] * rs) * wv + bv;
            }
            *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = r;
        }
    }
    else
    {
        for (int i = lane; i < dim; i += 64)
        {
            const size_t off = (size_t)i * inner;
            const float wv = w ? w[i] + w_offset : 1.0f;
            const float bv = b ? b[i] : 0.0f;
            y[off] = (x[off] * rs) * wv + bv;
        }
    }
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_rmsnorm_bf16(uint16_t* Y, uint16_t* rstd, const uint16_t* X, const uint16_t* w, const uint16_t* b,
                            int outer, int inner, int dim, float eps, float w_offset, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X, "rmsnorm_bf16: null pointer");
    MILA_REQUIRE(outer > 0 && dim > 0 && inner > 0, "rmsnorm_bf16: outer/inner/dim must be positive (%d,%d,%d)", outer, inner, dim);
    MILA_REQUIRE(!(b && !w), "rmsnorm_bf16: bias without weight is not a reference configuration");
    hipStream_t s = as_stream(stream);
    if (inner == 1 && dim % 8 == 0)
    {
        if (dim > 1024 && dim <= kRmsBlockMaxGroups * 512)
            hipLaunchKernelGGL(rmsnorm_block_kernel, dim3(outer), dim3(256), 0, s, Y, rstd, X, w, b, dim, eps, w_offset);
        else
            hipLaunchKernelGGL(rmsnorm_wave_kernel, dim3(ceil_div(outer, 4)), dim3(256), 0, s, Y, rstd, X, w, b, outer,
                               dim, eps, w_offset);
    }
    else
    {
        const int64_t slices = (int64_t)outer * inner;
        MILA_REQUIRE(slices < (1ll << 31), "rmsnorm_bf16: too many slices");
        hipLaunchKernelGGL(rmsnorm_strided_kernel, dim3(ceil_div(slices, 4)), dim3(256), 0, s, Y, rstd, X, w, b,
                           (int)slices, dim, inner, eps, w_offset);
    }
    MILA_LAUNCH_CHECK("rmsnorm_bf16");
}

int mila_cdna4_rmsnorm_fp32(float* Y, float* rstd, const float* X, const float* w, const float* b, int outer, int inner, int dim, float eps,
                            float w_offset, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X, "rmsnorm_fp32: null pointer");
    MILA_REQUIRE(outer > 0 && dim > 0 && inner > 0, "rmsnorm_fp32: outer/inner/dim must be positive (%d,%d,%d)", outer, inner, dim);
    MILA_REQUIRE(!(b && !w), "rmsnorm_fp32: bias without weight is not a reference configuration");
    const int64_t slices = (int64_t)outer * inner;
    MILA_REQUIRE(slices < (1ll << 31), "rmsnorm_fp32: too many slices");
    hipStream_t s = as_stream(stream);
    if (inner == 1 && dim % 4 == 0)
        hipLaunchKernelGGL(rmsnorm_fp32_kernel<true>, dim3(ceil_div(slices, 4)), dim3(256), 0, s, Y, rstd, X, w, b, (int)slices, dim, inner, eps, w_offset);
    else
        hipLaunchKernelGGL(rmsnorm_fp32_kernel<false>, dim3(ceil_div(slices, 4)), dim3(256), 0, s, Y, rstd, X, w, b, (int)slices, dim, inner, eps, w_offset);
    MILA_LAUNCH_CHECK("rmsnorm_fp32");
}

int mila_cdna4_fused_tail_norm_quant_bf16(uint16_t* R, uint16_t* XN, uint8_t* XQ, float* XS, const uint16_t* A, const uint16_t* RES, const uint16_t* post_w,
                                          const uint16_t* next_w, int rows, int dim, float post_scale, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(R && XN && XQ && XS && A && RES && post_w && next_w, "fused_tail_norm_quant_bf16: null pointer");
    MILA_REQUIRE(rows > 0 && dim > 0, "fused_tail_norm_quant_bf16: rows/dim must be positive (%d,%d)", rows, dim);
    MILA_REQUIRE(dim % 8 == 0 && dim > 1024 && dim <= 8 * 256 * kTailChunks, "fused_tail_norm_quant_bf16: dim=%d must be a multiple of 8 in (1024, %d]", dim, 8 * 256 * kTailChunks);
    const int nch = (dim / 8 + 255) / 256;
    if (nch <= 1) hipLaunchKernelGGL((tail_norm_kernel<1, true>), dim3(rows), dim3(256), 0, as_stream(stream), R, XN, A, RES, post_w, next_w, dim, post_scale, eps, XQ, XS);
    else if (nch == 2) hipLaunchKernelGGL((tail_norm_kernel<2, true>), dim3(rows), dim3(256), 0, as_stream(stream), R, XN, A, RES, post_w, next_w, dim, post_scale, eps, XQ, XS);
    else if (nch == 3) hipLaunchKernelGGL((tail_norm_kernel<3, true>), dim3(rows), dim3(256), 0, as_stream(stream), R, XN, A, RES, post_w, next_w, dim, post_scale, eps, XQ, XS);
    else hipLaunchKernelGGL((tail_norm_kernel<4, true>), dim3(rows), dim3(256), 0, as_stream(stream), R, XN, A, RES, post_w, next_w, dim, post_scale, eps, XQ, XS);
    MILA_LAUNCH_CHECK("fused_tail_norm_quant_bf16");
}

int mila_cdna4_fused_tail_norm_bf16(uint16_t* R, uint16_t* XN, const uint16_t* A, const uint16_t* RES, const uint16_t* post_w,
                                    const uint16_t* next_w, int rows, int dim, float post_scale, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(R && A && RES && post_w, "fused_tail_norm_bf16: null pointer");
    MILA_REQUIRE((XN == nullptr) == (next_w == nullptr), "fused_tail_norm_bf16: XN and next_w go together");
    MILA_REQUIRE(rows > 0 && dim > 0, "fused_tail_norm_bf16: rows/dim must be positive (%d,%d)", rows, dim);
    MILA_REQUIRE(dim % 8 == 0 && dim > 1024 && dim <= 8 * 256 * kTailChunks, "fused_tail_norm_bf16: dim=%d must be a multiple of 8 in (1024, %d]", dim, 8 * 256 * kTailChunks);
    const int nch = (dim / 8 + 255) / 256;
    if (nch <= 1) hipLaunchKernelGGL(tail_norm_kernel<1>, dim3(rows), dim3(256), 0, as_stream(stream), R, XN, A, RES, post_w, next_w, dim, post_scale, eps, (uint8_t*)nullptr, (float*)nullptr);
    else if (nch == 2) hipLaunchKernelGGL(tail_norm_kernel<2>, dim3(rows), dim3(256), 0, as_stream(stream), R, XN, A, RES, post_w, next_w, dim, post_scale, eps, (uint8_t*)nullptr, (float*)nullptr);
    else if (nch == 3) hipLaunchKernelGGL(tail_norm_kernel<3>, dim3(rows), dim3(256), 0, as_stream(stream), R, XN, A, RES, post_w, next_w, dim, post_scale, eps, (uint8_t*)nullptr, (float*)nullptr);
    else hipLaunchKernelGGL(tail_norm_kernel<4>, dim3(rows), dim3(256), 0, as_stream(stream), R, XN, A, RES, post_w, next_w, dim, post_scale, eps, (uint8_t*)nullptr, (float*)nullptr);
    MILA_LAUNCH_CHECK("fused_tail_norm_bf16");
}

int mila_cdna4_layernorm_bf16(uint16_t* Y, float* mean, float* rstd, const uint16_t* X, const uint16_t* w,
                              const uint16_t* b, int outer, int dim, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X, "layernorm_bf16: null pointer");
    MILA_REQUIRE(outer > 0 && dim > 0, "layernorm_bf16: outer/dim must be positive (%d,%d)", outer, dim);
    if (dim % 8 == 0 && dim <= 4096)
    {
        const int nv = (dim / 8 + 63) / 64, blocks = ceil_div(outer, 4);
        hipStream_t s = as_stream(stream);
        if (nv <= 1) hipLaunchKernelGGL(layernorm_wave_bf16_kernel<1>, dim3(blocks), dim3(256), 0, s, Y, mean, rstd, X, w, b, outer, dim, eps);
        else if (nv == 2) hipLaunchKernelGGL(layernorm_wave_bf16_kernel<2>, dim3(blocks), dim3(256), 0, s, Y, mean, rstd, X, w, b, outer, dim, eps);
        else if (nv <= 4) hipLaunchKernelGGL(layernorm_wave_bf16_kernel<4>, dim3(blocks), dim3(256), 0, s, Y, mean, rstd, X, w, b, outer, dim, eps);
        else hipLaunchKernelGGL(layernorm_wave_bf16_kernel<8>, dim3(blocks), dim3(256), 0, s, Y, mean, rstd, X, w, b, outer, dim, eps);
        MILA_LAUNCH_CHECK("layernorm_bf16");
    }
    hipLaunchKernelGGL(layernorm_kernel<uint16_t>, dim3(outer), dim3(256), 0, as_stream(stream), Y, mean, rstd, X, w, b,
                       dim, eps);
    MILA_LAUNCH_CHECK("layernorm_bf16");
}

int mila_cdna4_layernorm_fp32(float* Y, float* mean, float* rstd, const float* X, const float* w, const float* b,
                              int outer, int dim, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X, "layernorm_fp32: null pointer");
    MILA_REQUIRE(outer > 0 && dim > 0, "layernorm_fp32: outer/dim must be positive (%d,%d)", outer, dim);
    hipLaunchKernelGGL(layernorm_kernel<float>, dim3(outer), dim3(256), 0, as_stream(stream), Y, mean, rstd, X, w, b,
                       dim, eps);
    MILA_LAUNCH_CHECK("layernorm_fp32");
}

int mila_cdna4_softmax_fp32(float* Y, const float* X, int outer, int dim, int inner, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X, "softmax_fp32: null pointer");
    MILA_REQUIRE(outer > 0 && dim > 0 && inner > 0, "softmax_fp32: outer/dim/inner must be positive");
    const int64_t slices = (int64_t)outer * inner;
    MILA_REQUIRE(slices < (1ll << 31), "softmax_fp32: too many slices");
    hipLaunchKernelGGL(softmax_kernel<float>, dim3(ceil_div(slices, 4)), dim3(256), 0, as_stream(stream), Y, X,
                       (int)slices, dim, inner);
    MILA_LAUNCH_CHECK("softmax_fp32");
}

int mila_cdna4_softmax_bf16(uint16_t* Y, const uint16_t* X, int outer, int dim, int inner, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X, "softmax_bf16: null pointer");
    MILA_REQUIRE(outer > 0 && dim > 0 && inner > 0, "softmax_bf16: outer/dim/inner must be positive");
    const int64_t slices = (int64_t)outer * inner;
    MILA_REQUIRE(slices < (1ll << 31), "softmax_bf16: too many slices");
    hipLaunchKernelGGL(softmax_kernel<uint16_t>, dim3(ceil_div(slices, 4)), dim3(256), 0, as_stream(stream), Y, X,
                       (int)slices, dim, inner);
    MILA_LAUNCH_CHECK("softmax_bf16");
}

}  // extern "C"
