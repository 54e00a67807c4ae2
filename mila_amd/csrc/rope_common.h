// Canonical RoPE rotation of 8 pairs, shared by rope.hip and the fused q/k post-processing kernel
// so both produce identical bits.  OPS/Encodings/Rope/Kernels/Rope.Bf16.cu:45-69.
#pragma once
#include "common.h"

namespace mila {

// rotate pairs (i+j, i+j+half), j = 0..7, of one head row; in-place safe (reads before writes)
// the rotation on cache values the caller already holds (a caller that wants them requested early, with its other operands)
__device__ __forceinline__ void rope_rotate8_regs(u32x4& lo, u32x4& hi, f32x4 c0, f32x4 c1, f32x4 s0, f32x4 s1);
__device__ __forceinline__ void rope_rotate8_vals(u32x4& lo, u32x4& hi, const float* __restrict__ cos_row,
                                                  const float* __restrict__ sin_row, int i)
{
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cos_row + i), c1 = *reinterpret_cast<const f32x4*>(cos_row + i + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sin_row + i), s1 = *reinterpret_cast<const f32x4*>(sin_row + i + 4);
    rope_rotate8_regs(lo, hi, c0, c1, s0, s1);
}
__device__ __forceinline__ void rope_rotate8_regs(u32x4& lo, u32x4& hi, f32x4 c0, f32x4 c1, f32x4 s0, f32x4 s1)
{
    const float c[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    u32x4 rlo, rhi;
#pragma unroll
    for (int d = 0; d < 4; ++d)
    {
        const float x0a = bf16_lo(lo[d]), x0b = bf16_hi(lo[d]);
        const float x1a = bf16_lo(hi[d]), x1b = bf16_hi(hi[d]);
        // one product rounded, the other fused into the sum -- spelled out, so that every kernel sharing this helper rounds alike whatever the compiler
        // would have contracted around it (the fused q/k post-processing requests its cache rows early; the standalone kernel does not)
        rlo[d] = pack_bf16x2(__fmaf_rn(x0a, c[2 * d], -__fmul_rn(x1a, s[2 * d])), __fmaf_rn(x0b, c[2 * d + 1], -__fmul_rn(x1b, s[2 * d + 1])));
        rhi[d] = pack_bf16x2(__fmaf_rn(x0a, s[2 * d], __fmul_rn(x1a, c[2 * d])), __fmaf_rn(x0b, s[2 * d + 1], __fmul_rn(x1b, c[2 * d + 1])));
    }
    lo = rlo;
    hi = rhi;
}

__device__ __forceinline__ void rope_rotate8(uint16_t* __restrict__ out_row, const uint16_t* __restrict__ in_row,
                                             const float* __restrict__ cos_row, const float* __restrict__ sin_row, int i,
                                             int half)
{
    u32x4 lo = ld16(in_row + i), hi = ld16(in_row + i + half);
    rope_rotate8_vals(lo, hi, cos_row, sin_row, i);
    st16(out_row + i, lo);
    st16(out_row + i + half, hi);
}

}  // namespace mila
