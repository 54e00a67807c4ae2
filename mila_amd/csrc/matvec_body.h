// Body of the decode (M == 1) Linear kernels, shared by the one-launch-per-Linear kernel (matvec.hip) and the
// multi-phase decode chain (chain.hip).  See matvec.hip for the design notes.
#pragma once
#include "common.h"
#include "rms_common.h"

namespace mila {

enum { FMT_BF16 = 0, FMT_FP8 = 1, FMT_FP4 = 2 };

template <int FMT> struct Fmt;
template <> struct Fmt<FMT_BF16> { static constexpr int kElemsPerChunk = 8; };
template <> struct Fmt<FMT_FP8> { static constexpr int kElemsPerChunk = 16; };
template <> struct Fmt<FMT_FP4> { static constexpr int kElemsPerChunk = 32; };

// dot of one 16-byte weight chunk with the matching x values (bf16 pairs in LDS, 16-byte units)
template <int FMT>
__device__ __forceinline__ float chunk_dot(const u32x4 w, const u32x4* __restrict__ xs, int c, float acc)
{
    if constexpr (FMT == FMT_BF16)
    {
        const u32x4 xv = xs[c];
#pragma unroll
        for (int d = 0; d < 4; ++d) acc = dot2_bf16(as_bf16x2(w[d]), as_bf16x2(xv[d]), acc);
    }
    else if constexpr (FMT == FMT_FP8)
    {
        const u32x4 x0 = xs[2 * c], x1 = xs[2 * c + 1];
        acc = dot2_bf16(fp8x2_to_bf16x2(w[0], false), as_bf16x2(x0[0]), acc);
        acc = dot2_bf16(fp8x2_to_bf16x2(w[0], true), as_bf16x2(x0[1]), acc);
        acc = dot2_bf16(fp8x2_to_bf16x2(w[1], false), as_bf16x2(x0[2]), acc);
        acc = dot2_bf16(fp8x2_to_bf16x2(w[1], true), as_bf16x2(x0[3]), acc);
        acc = dot2_bf16(fp8x2_to_bf16x2(w[2], false), as_bf16x2(x1[0]), acc);
        acc = dot2_bf16(fp8x2_to_bf16x2(w[2], true), as_bf16x2(x1[1]), acc);
        acc = dot2_bf16(fp8x2_to_bf16x2(w[3], false), as_bf16x2(x1[2]), acc);
        acc = dot2_bf16(fp8x2_to_bf16x2(w[3], true), as_bf16x2(x1[3]), acc);
    }
    else
    {
#pragma unroll
        for (int d = 0; d < 4; ++d)
        {
            const u32x4 xv = xs[4 * c + d];
            acc = dot2_bf16(fp4x2_to_bf16x2<0>(w[d]), as_bf16x2(xv[0]), acc);
            acc = dot2_bf16(fp4x2_to_bf16x2<1>(w[d]), as_bf16x2(xv[1]), acc);
            acc = dot2_bf16(fp4x2_to_bf16x2<2>(w[d]), as_bf16x2(xv[2]), acc);
            acc = dot2_bf16(fp4x2_to_bf16x2<3>(w[d]), as_bf16x2(xv[3]), acc);
        }
    }
    return acc;
}

struct MatvecParams
{
    void* y;
    const uint16_t* x;
    const uint8_t* W;
    const float* scales;
    const uint16_t* bias;
    // prologue operands (PRO != 0)
    const uint16_t* norm_w;
    const uint16_t* post_w;
    const uint16_t* res;
    uint16_t* res_out;
    float post_scale, eps;
    int K, N, group;
    int comb_splits, comb_heads;   // X_COMBINE: x = combine of the decode-attention split partials [heads, splits, HS + 4] at p.x
    // Y_F32 (lm_head) only, both or neither: the greedy sampler's FIRST stage in this kernel's epilogue -- workgroup b writes the largest of its rows' outputs and its
    // column (ties to the lowest column, Sampling.cu:23-75) to amax_v[b] / amax_i[b]; the sampler's final stage reduces the gridDim.x partials (one launch fewer per token)
    float* amax_v = nullptr;
    int* amax_i = nullptr;
};

// the sampler's order: strictly larger wins, equal values go to the lower index, NaN never wins (csrc/sampling.hip: better)
__device__ __forceinline__ void argmax_better(float& bv, int& bi, float v, int i)
{
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

constexpr int kMatvecWaves = 16;   // 1024 threads

template <int FMT, int U, int NR>
struct WBuf
{
    u32x4 w[U][NR];
    float sc[U][NR];
};

// uniform 32-bit load through the scalar cache (read-only data: scales, bias)
__device__ __forceinline__ uint32_t sload32(const void* p_uniform)
{
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p_uniform) : "memory");
    return v;
}

// ---- in-launch hand-off vectors (chain.hip) ----------------------------------------------------
// A hand-off vector holds one activation element per 32-bit word (the bf16 bits in the low half).  It is
// written with 4-byte write-through (sc1) stores and read with 8-byte sc1 loads, the measured store / load
// forms of MI355X_MICROARCH.md "visibility" table row 1; see chain.hip for the signalling around them.
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
// generic -> global address space (the hand-off words live in hipMalloc memory): global_ instructions, never flat_
__device__ __forceinline__ gu32* as_global(uint32_t* p) { return (gu32*)p; }
__device__ __forceinline__ const gu32* as_global(const uint32_t* p) { return (const gu32*)p; }
__device__ __forceinline__ gu64* as_global(unsigned long long* p) { return (gu64*)p; }
__device__ __forceinline__ const gu64* as_global(const unsigned long long* p) { return (const gu64*)p; }
__device__ __forceinline__ void handoff_store(uint32_t* v, int i, uint32_t bf16_bits)
{
    __hip_atomic_store(as_global(v) + i, bf16_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// elements 8 c .. 8 c + 7 packed as a 16-byte bf16 chunk
__device__ __forceinline__ u32x4 handoff_load8(const uint32_t* v, int c)
{
    const gu64* q = as_global(reinterpret_cast<const unsigned long long*>(v)) + (size_t)c * 4;
    u32x4 r;
#pragma unroll
    for (int d = 0; d < 4; ++d)
    {
        const unsigned long long t = __hip_atomic_load(q + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r[d] = ((uint32_t)t & 0xffffu) | ((uint32_t)(t >> 32) << 16);
    }
    return r;
}

struct NoWait { __device__ __forceinline__ void operator()() const {} };

enum { Y_BF16 = 0, Y_F32 = 1, Y_HANDOFF = 2 };     // where a phase writes its outputs
enum { X_PLAIN = 0, X_HANDOFF = 1, X_COMBINE_1 = 2, X_COMBINE_2 = 3 };   // where x comes from (p.x reinterpreted); COMBINE_n: n heads per wave
enum { RES_MEM = 0, RES_REG = 1 };                 // residual operand of the sandwich tail

// PRO: 0 = x as is; 1 = x <- rmsnorm(x; norm_w); 2 = sandwich tail (see mila_cdna4.h)
// ROWS = R output columns per wave; with GEGLU each column reads two weight rows (n, N + n).
// XC   = 16-byte x chunks each thread preloads (1024 * 8 * XC >= K): 1 covers K <= 8192, 2 covers K <= 16384.
// `wait()` runs after the weight prefetch and before any X_HANDOFF load (the chain's grid barrier).
// `rkeep`: residual in (RES_REG) / sandwich-tail result r out (PRO == 2), chunk tid + 1024 k per thread.
template <int FMT, int R, int U, int PRO, bool GEGLU, int YDST, int XC, int XSRC, int RSRC, class Wait>
__device__ __forceinline__ void matvec_body(const MatvecParams& p, u32x4* xs, float* red_a, float* red_b,
                                            const int block, const int nblocks, u32x4 (&rkeep)[XC], Wait wait)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = p.K, N = p.N;
    const int nx16 = K / 8;   // 16-byte units of x

    constexpr int EPC = Fmt<FMT>::kElemsPerChunk;
    constexpr int XPC = EPC / 8;                       // 16-byte x units per weight chunk
    constexpr int NR = GEGLU ? 2 * R : R;              // weight rows per wave step
    const int nchunks = K / EPC;                       // 16-byte chunks per weight row
    const size_t row_bytes = (size_t)nchunks * 16;
    const int S = (nchunks + 64 * U - 1) / (64 * U);   // pipeline steps per row-group
    const int nx16_pad = S * 64 * U * XPC;             // x units covered by the chunk positions of S steps
    const int ngroups = (FMT == FMT_FP4) ? K / p.group : 0;
    const int cpg_shift = (FMT == FMT_FP4) ? (p.group == 128 ? 2 : 1) : 0;   // chunks per group = group / 32
    const int total_waves = nblocks * kMatvecWaves;
    const int n_rg = (N + R - 1) / R;
    const int wave_g = block * kMatvecWaves + wib;
    const int nrg_w = wave_g < n_rg ? (n_rg - 1 - wave_g) / total_waves + 1 : 0;
    const int T = nrg_w * S;                           // pipeline steps of this wave

    // ---- x / prologue operands first (everything that does not wait for another workgroup) ----
    u32x4 px[XC], pnw[PRO != 0 ? XC : 1], ppw[PRO == 2 ? XC : 1], pres[PRO == 2 ? XC : 1];
#pragma unroll
    for (int k = 0; k < XC; ++k)
    {
        const size_t e = (size_t)min(tid + 1024 * k, nx16 - 1) * 8;
        if constexpr (XSRC == X_PLAIN) px[k] = ld16(p.x + e);
        if constexpr (PRO != 0) pnw[k] = ld16(p.norm_w + e);
        if constexpr (PRO == 2)
        {
            ppw[k] = ld16(p.post_w + e);
            if constexpr (RSRC == RES_MEM) pres[k] = ld16(p.res + e);
            else pres[k] = rkeep[k];
        }
    }

    // issue the loads of one pipeline step: row-group rg, step s (chunk positions 64 U s + lane + 64 u)
    auto issue = [&](WBuf<FMT, U, NR>& b, int rg, int s) {
        const int c0 = s * (64 * U) + lane;
#pragma unroll
        for (int j = 0; j < NR; ++j)
        {
            const int col = min(rg * R + (GEGLU ? (j >> 1) : j), N - 1);
            const int row = (GEGLU && (j & 1)) ? (N + col) : col;
            const uint8_t* wrow = p.W + (size_t)row * row_bytes;
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                const int c = min(c0 + 64 * u, nchunks - 1);
                b.w[u][j] = ld16_nt(wrow + (size_t)c * 16);
                if constexpr (FMT == FMT_FP4) b.sc[u][j] = p.scales[(size_t)row * ngroups + (c >> cpg_shift)];
            }
        }
    };
    // wave-uniform cursors over (row-group, step): `ci` issues, two steps ahead of `cc`, which computes
    int ci_rg = wave_g, ci_s = 0, cc_rg = wave_g, cc_s = 0;
    auto advance = [&](int& rg_, int& s_) {
        const bool wrap = (s_ + 1 == S);
        s_ = wrap ? 0 : s_ + 1;
        rg_ = wrap ? rg_ + total_waves : rg_;
    };
    WBuf<FMT, U, NR> ba, bb;
    issue(ba, ci_rg, ci_s); advance(ci_rg, ci_s);
    issue(bb, ci_rg, ci_s); advance(ci_rg, ci_s);

    // ---- x = attention output combined from the flash-decode split partials (attn_combine_kernel's arithmetic, element for
    //      element: M = max m_s, f_s = exp(m_s - M), L = sum l_s f_s (64-lane butterflies with lane = split), acc = fma chain over
    //      the splits in order, y = bf16(acc / L)); a wave's 64 chunks span HPW = 512 / HS heads.  Runs under the weight prefetch.
    if constexpr (XSRC == X_COMBINE_1 || XSRC == X_COMBINE_2)
    {
        constexpr int HPW = (XSRC == X_COMBINE_1) ? 1 : 2;
        const float* part = reinterpret_cast<const float*>(p.x);
        const int splits = p.comb_splits, HS = 512 / HPW, STR = HS + 4;
#pragma unroll
        for (int k = 0; k < XC; ++k)
        {
            const int c = tid + 1024 * k;
            const int h0 = ((c & ~63) * 8) / HS;                       // first head of this wave's 64 chunks
            const int myj = (HPW == 1) ? 0 : (lane >> 5);              // this lane's head within the wave
            float fs[HPW], Lh[HPW];
#pragma unroll
            for (int j = 0; j < HPW; ++j)
            {
                const int h = min(h0 + j, p.comb_heads - 1);
                const float* hb = part + (size_t)h * splits * STR;
                const float ms = (lane < splits) ? hb[(size_t)lane * STR + HS] : -INFINITY;
                const float ls = (lane < splits) ? hb[(size_t)lane * STR + HS + 1] : 0.0f;
                const float M = wave_max(ms);
                fs[j] = (ms == -INFINITY) ? 0.0f : __expf(ms - M);
                Lh[j] = wave_sum(ls * fs[j]);
            }
            const int h = min(h0 + myj, p.comb_heads - 1);
            const float* vb = part + (size_t)h * splits * STR + (size_t)((c * 8) % HS);
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
            for (int s0 = 0; s0 < splits; s0 += 8)
            {
                f32x4 va[8], vb2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                {
                    const float* src = vb + (size_t)min(s0 + u, splits - 1) * STR;
                    va[u] = *reinterpret_cast<const f32x4*>(src);
                    vb2[u] = *reinterpret_cast<const f32x4*>(src + 4);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                {
                    const int si = min(s0 + u, 63);
                    float f = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fs[0]), si));
                    if constexpr (HPW == 2)
                    {
                        const float f1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fs[1]), si));
                        f = myj ? f1 : f;
                    }
                    if (s0 + u >= splits) f = 0.0f;                     // past the last split: fma(x, 0, acc) leaves acc as it is
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                    {
                        acc[e] = fmaf(va[u][e], f, acc[e]);
                        acc[4 + e] = fmaf(vb2[u][e], f, acc[4 + e]);
                    }
                }
            }
            float L = Lh[0];
            if constexpr (HPW == 2) L = myj ? Lh[1] : Lh[0];
#pragma unroll
            for (int d = 0; d < 4; ++d)
                px[k][d] = pack_bf16x2(L > 0.0f ? acc[2 * d] / L : 0.0f, L > 0.0f ? acc[2 * d + 1] / L : 0.0f);
        }
    }

    // ---- the previous phase's outputs become readable here (chain); no-op for the plain kernel ----
    wait();
    if constexpr (XSRC == X_HANDOFF)
    {
#pragma unroll
        for (int k = 0; k < XC; ++k)
            px[k] = handoff_load8(reinterpret_cast<const uint32_t*>(p.x), min(tid + 1024 * k, nx16 - 1));
    }

    // ---- stage x into LDS (optionally through the fused RMSNorm prologue), zero-pad the tail ----
    for (int i = nx16 + tid; i < nx16_pad; i += 1024) xs[i] = u32x4{0u, 0u, 0u, 0u};
    if constexpr (PRO == 0)
    {
#pragma unroll
        for (int k = 0; k < XC; ++k)
            if (tid + 1024 * k < nx16) xs[tid + 1024 * k] = px[k];
    }
    else
    {
        // the canonical order of rms_rstd_block (rms_common.h): chunk tid + 1024 k belongs to group wib + 16 k
        const int G = (nx16 + 63) / 64;
        auto rstd_of = [&](const u32x4* v, float* red) {
#pragma unroll
            for (int k = 0; k < XC; ++k)
            {
                float s = (tid + 1024 * k < nx16) ? sumsq8(v[k], 0.0f) : 0.0f;
                s = wave_sum(s);
                if (lane == 0) red[wib + kMatvecWaves * k] = s;
            }
            __syncthreads();
            float t = 0.0f;
            for (int g = 0; g < G; ++g) t += red[g];
            return rsqrtf(t / (float)K + p.eps);
        };
        if constexpr (PRO == 1)
        {
            const float rstd = rstd_of(px, red_a);
#pragma unroll
            for (int k = 0; k < XC; ++k)
                if (tid + 1024 * k < nx16) xs[tid + 1024 * k] = rms_apply8(px[k], pnw[k], rstd, 0.0f);
        }
        else
        {
            const float rstd_a = rstd_of(px, red_a);
#pragma unroll
            for (int k = 0; k < XC; ++k) rkeep[k] = sandwich_tail8(rms_apply8(px[k], ppw[k], rstd_a, 0.0f), pres[k], p.post_scale);
            if (block == 0 && p.res_out != nullptr)
            {
#pragma unroll
                for (int k = 0; k < XC; ++k)
                    if (tid + 1024 * k < nx16) st16(p.res_out + (size_t)(tid + 1024 * k) * 8, rkeep[k]);
            }
            const float rstd_r = rstd_of(rkeep, red_b);
#pragma unroll
            for (int k = 0; k < XC; ++k)
                if (tid + 1024 * k < nx16) xs[tid + 1024 * k] = rms_apply8(rkeep[k], pnw[k], rstd_r, 0.0f);
        }
    }
    __syncthreads();

    float acc[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[j] = 0.0f;

    auto compute = [&](const WBuf<FMT, U, NR>& b, int s) {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const int c = s * (64 * U) + lane + 64 * u;    // < S * 64 * U: inside the zero-padded x
#pragma unroll
            for (int j = 0; j < NR; ++j)
            {
                if constexpr (FMT == FMT_FP4)
                    acc[j] = fmaf(b.sc[u][j], chunk_dot<FMT>(b.w[u][j], xs, c, 0.0f), acc[j]);
                else
                    acc[j] = chunk_dot<FMT>(b.w[u][j], xs, c, acc[j]);
            }
        }
    };

    float amax_bv = -3.402823466e+38f;      // lane 0 of each wave: the best (value, column) among the rows this wave has stored (Y_F32 with p.amax_v)
    int amax_bi = 0x7fffffff;
    auto finish = [&](int rg_) {
        const int col0 = rg_ * R;
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[j] = wave_sum(acc[j]);
#pragma unroll
        for (int j = 0; j < NR; ++j)
        {
            const int col = min(col0 + (GEGLU ? (j >> 1) : j), N - 1);
            const int row = (GEGLU && (j & 1)) ? (N + col) : col;
            float v = acc[j];
            if constexpr (FMT == FMT_FP8) v = __builtin_bit_cast(float, sload32(p.scales + row)) * v;
            if (p.bias)
            {
                const uint32_t pair = sload32(reinterpret_cast<const uint32_t*>(p.bias) + (row >> 1));
                v += bf16_bits_to_f32((uint16_t)((row & 1) ? (pair >> 16) : (pair & 0xffffu)));
            }
            acc[j] = v;
        }
        if (lane == 0)
        {
#pragma unroll
            for (int r = 0; r < R; ++r)
            {
                const int col = col0 + r;
                if (col >= N) continue;
                float v = acc[r];
                if constexpr (GEGLU)
                {
                    // unfused chain: gate/up stored as bf16 by the Linear, then
                    // bf16(gelu_tanh(gate) * up) by the GeGLU kernel
                    const float g = round_bf16(acc[2 * r]), up = round_bf16(acc[2 * r + 1]);
                    v = gelu_tanh(g) * up;
                }
                if constexpr (YDST == Y_F32)
                {
                    reinterpret_cast<float*>(p.y)[col] = v;
                    if (p.amax_v) argmax_better(amax_bv, amax_bi, v, col);
                }
                else if constexpr (YDST == Y_HANDOFF) handoff_store(reinterpret_cast<uint32_t*>(p.y), col, f32_to_bf16_bits(v));
                else reinterpret_cast<uint16_t*>(p.y)[col] = f32_to_bf16_bits(v);
            }
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[j] = 0.0f;
    };

    // consume one step from BUF; at the end of a row-group reduce and store it
    auto consume = [&](const WBuf<FMT, U, NR>& b) {
        compute(b, cc_s);
        const int rg_prev = cc_rg;
        advance(cc_rg, cc_s);
        if (cc_rg != rg_prev) finish(rg_prev);
    };

    int t = 0;
    for (; t + 4 <= T; t += 2)      // steps t + 2 and t + 3 exist: refill unconditionally
    {
        consume(ba);
        issue(ba, ci_rg, ci_s); advance(ci_rg, ci_s);
        consume(bb);
        issue(bb, ci_rg, ci_s); advance(ci_rg, ci_s);
    }
    const int rem = T - t;          // 0 (idle wave), 1, 2 or 3
    if (rem >= 1)
    {
        consume(ba);
        if (rem == 3) issue(ba, ci_rg, ci_s);
    }
    if (rem >= 2) consume(bb);
    if (rem == 3) consume(ba);

    if constexpr (YDST == Y_F32)
    {
        // the sampler's first stage: the workgroup's 16 per-wave candidates through LDS (red_a / red_b are free since the prologue), one partial per workgroup
        if (p.amax_v)
        {
            __syncthreads();
            if (lane == 0) { red_a[wib] = amax_bv; red_b[wib] = __int_as_float(amax_bi); }
            __syncthreads();
            if (tid == 0)
            {
                float bv = red_a[0];
                int bi = __float_as_int(red_b[0]);
#pragma unroll
                for (int w = 1; w < kMatvecWaves; ++w) argmax_better(bv, bi, red_a[w], __float_as_int(red_b[w]));
                p.amax_v[block] = bv;
                p.amax_i[block] = bi;
            }
        }
    }
}

}  // namespace mila
