// RoPE: fp32 cos/sin cache build and half-split (NeoX) rotation on bf16.
//   cache:    OPS/Encodings/Rope/Kernels/Rope.Fp32.cu:27-59,288-321
//             theta_i = base^(-2i/HS) for i < rope_pairs, else (cos,sin) = (1,0);
//             angle = float(pos) * theta (fp32), cosf/sinf.
//             The reference evaluates theta with the fast __powf; here theta is base^(-2i/HS)
//             evaluated in double and rounded once to fp32 (the correctly rounded value __powf
//             approximates), so the cache agrees with the host reference of the reference's own
//             test (Tests/.../Rope.Cuda.cpp:51-96) to fp32 rounding.
//   rotation: OPS/Encodings/Rope/Kernels/Rope.Bf16.cu:28-118: r0 = x0 c - x1 s, r1 = x0 s + x1 c.
#include "common.h"
#include "rope_common.h"

namespace mila {

__global__ __launch_bounds__(256) void rope_build_cache_kernel(float* __restrict__ cos_out, float* __restrict__ sin_out,
                                                               int half, int max_seq, float base, int rope_pairs)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)max_seq * half) return;
    const int pos = (int)(idx / half), i = (int)(idx % half);
    if (i < rope_pairs)
    {
        const float theta = (float)pow((double)base, -2.0 * (double)i / (double)(half * 2));
        const float angle = (float)pos * theta;
        cos_out[idx] = cosf(angle);
        sin_out[idx] = sinf(angle);
    }
    else
    {
        cos_out[idx] = 1.0f;
        sin_out[idx] = 0.0f;
    }
}

// one thread per 8 rotation pairs (two 16-byte loads: x[i..i+8), x[i+half..i+half+8))
__global__ __launch_bounds__(256) void rope_rotate_bf16_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in,
                                                               const float* __restrict__ cos_c,
                                                               const float* __restrict__ sin_c, int64_t total_vec,
                                                               int half, int T, int n_heads, int pos_offset)
{
    const int hv = half / 8;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total_vec; v += stride)
    {
        const int64_t bth = v / hv;               // (b*T + t)*n_heads + h
        const int i = (int)(v % hv) * 8;
        const int t = (int)((bth / n_heads) % T);
        const int pos = t + pos_offset;
        const size_t base = (size_t)bth * half * 2;
        rope_rotate8(out + base, in + base, cos_c + (size_t)pos * half, sin_c + (size_t)pos * half, i, half);
    }
}

// any even head size (HS % 16 != 0: the reference's own test geometry is HS = 8, Tests/.../Rope.Cuda.cpp:42): one thread per
// rotation pair, the same arithmetic and rounding as rope_rotate8_vals
__global__ __launch_bounds__(256) void rope_rotate_bf16_pair_kernel(uint16_t* out, const uint16_t* in,   /* may alias (in place) */
                                                                    const float* __restrict__ cos_c, const float* __restrict__ sin_c,
                                                                    int64_t total_pairs, int half, int T, int n_heads, int pos_offset)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total_pairs; v += stride)
    {
        const int64_t bth = v / half;
        const int i = (int)(v % half);
        const int pos = (int)((bth / n_heads) % T) + pos_offset;
        const size_t base = (size_t)bth * half * 2;
        const float c = cos_c[(size_t)pos * half + i], s = sin_c[(size_t)pos * half + i];
        const float x0 = bf16_bits_to_f32(in[base + i]), x1 = bf16_bits_to_f32(in[base + i + half]);
        const uint32_t lo = pack_bf16x2(x0 * c - x1 * s, 0.0f), hi = pack_bf16x2(x0 * s + x1 * c, 0.0f);
        out[base + i] = (uint16_t)(lo & 0xffffu);
        out[base + i + half] = (uint16_t)(hi & 0xffffu);
    }
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_rope_build_cache(float* cos_cache, float* sin_cache, int max_seq, int HS, float base, int rotary_dim,
                                mila_stream_t stream)
{
    MILA_REQUIRE(cos_cache && sin_cache, "rope_build_cache: null pointer");
    MILA_REQUIRE(max_seq > 0 && HS > 0 && HS % 2 == 0, "rope_build_cache: bad sizes (max_seq=%d HS=%d)", max_seq, HS);
    const int half = HS / 2;
    const int pairs = (rotary_dim > 0 && rotary_dim < HS) ? rotary_dim / 2 : half;
    const int64_t n = (int64_t)max_seq * half;
    hipLaunchKernelGGL(rope_build_cache_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), cos_cache,
                       sin_cache, half, max_seq, base, pairs);
    MILA_LAUNCH_CHECK("rope_build_cache");
}

int mila_cdna4_rope_forward_bf16(uint16_t* Qout, uint16_t* Kout, const uint16_t* Qin, const uint16_t* Kin,
                                 const float* cos_cache, const float* sin_cache, int B, int T, int NH, int NKV, int HS,
                                 int pos_offset, int max_seq, mila_stream_t stream)
{
    MILA_REQUIRE(cos_cache && sin_cache, "rope_forward_bf16: null cache");
    MILA_REQUIRE((Qout && Qin) || (Kout && Kin), "rope_forward_bf16: nothing to rotate");
    MILA_REQUIRE(B > 0 && T > 0 && HS > 0 && HS % 2 == 0, "rope_forward_bf16: bad sizes (B=%d T=%d HS=%d)", B, T, HS);
    MILA_REQUIRE(pos_offset >= 0 && pos_offset + T <= max_seq,
                 "rope_forward_bf16: positions [%d,%d) exceed the cache length %d", pos_offset, pos_offset + T, max_seq);
    const int half = HS / 2;
    hipStream_t s = as_stream(stream);
    if (HS % 16 != 0)
    {
        for (int which = 0; which < 2; ++which)
        {
            uint16_t* o = which ? Kout : Qout;
            const uint16_t* in = which ? Kin : Qin;
            if (!o || !in) continue;
            const int heads = which ? NKV : NH;
            const int64_t tp = (int64_t)B * T * heads * half;
            int blocks = ceil_div(tp, 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(rope_rotate_bf16_pair_kernel, dim3(blocks), dim3(256), 0, s, o, in, cos_cache, sin_cache, tp, half, T, heads, pos_offset);
        }
        MILA_LAUNCH_CHECK("rope_forward_bf16");
    }
    if (Qout && Qin)
    {
        const int64_t tv = (int64_t)B * T * NH * (half / 8);
        int blocks = ceil_div(tv, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(rope_rotate_bf16_kernel, dim3(blocks), dim3(256), 0, s, Qout, Qin, cos_cache, sin_cache, tv, half,
                           T, NH, pos_offset);
    }
    if (Kout && Kin)
    {
        const int64_t tv = (int64_t)B * T * NKV * (half / 8);
        int blocks = ceil_div(tv, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(rope_rotate_bf16_kernel, dim3(blocks), dim3(256), 0, s, Kout, Kin, cos_cache, sin_cache, tv, half,
                           T, NKV, pos_offset);
    }
    MILA_LAUNCH_CHECK("rope_forward_bf16");
}

}  // extern "C"
