// Decode (M == 1) Linear: y = W x (+ bias) for three weight formats, plus the fused
// RMSNorm-prologue / GeGLU-epilogue variants used by the Gemma decode schedule.
//
// Replaces OPS/Linear/Kernels/MatVec/CudaMatVecBias.Bf16.cu:134-181 (bf16), :198-251 (fp8
// per-channel, scale after the reduction) and :271-508 (fp4 per-group) -- re-derived for CDNA4:
//   * HBM-bound: every weight byte is read exactly once with 16-byte non-temporal loads
//     (1 KiB per wave-instruction), R rows x U chunk positions in flight per lane;
//   * x is staged once per workgroup (one 16-wave workgroup per CU) in LDS and re-read with ds_read_b128,
//     so the vector L1 only ever sees the weight stream;
//   * one wave owns R whole rows (no cross-wave reduction, no barrier in the streaming loop);
//     the 64-lane reduction is a xor butterfly;
//   * fp8 / fp4 are expanded to bf16 pairs by the gfx950 v_cvt_scalef32_pk_bf16_{fp8,fp4}
//     converts (scale operand 1.0 => exact) and multiplied with v_dot2_f32_bf16, fp32 accumulate;
//     the fp4 group scale is folded with one FMA per 32-element chunk.
#include "common.h"
#include "internal.h"
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
};

// PRO: 0 = x as is; 1 = x <- rmsnorm(x; norm_w); 2 = sandwich tail (see mila_cdna4.h)
// ROWS = R output columns per wave; with GEGLU each column reads two weight rows (n, N + n).
// XC   = 16-byte x chunks each thread preloads (1024 * 8 * XC >= K): 1 covers K <= 8192, 2 covers K <= 16384.
// One 1024-thread workgroup (16 waves) per CU: x is staged (and the prologue computed) once per CU, so the
// L2 -> LDS staging traffic is 256 * 2K bytes whatever the weight format.
//
// The kernel is written so that the compiler's own s_waitcnt accounting stays COUNTED (vmcnt(n), n > 0):
//   * there is no load under divergent control flow -- out-of-range rows / chunk positions are clamped to a
//     valid address (x is zero-padded in LDS up to the last chunk position, so a clamped chunk contributes 0)
//     and per-row scalars (fp8 channel scale, bias) come through the scalar cache (s_load, lgkmcnt);
//   * x and the prologue operands are requested FIRST, into registers (vector memory returns in order:
//     anything issued behind the weight prefetch only becomes usable after the weights have landed), then
//     two pipeline steps of weights; the prologue's latency chain (reductions, barriers) runs under that
//     first HBM round trip;
//   * the weight stream is software-pipelined per wave over the flattened (row-group, chunk-position) space
//     with two named register buffers and a straight-line steady-state loop: consume step t, refill the same
//     buffer with step t + 2; the last <= 3 steps are peeled so the loop body needs no validity test.
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

template <int FMT, int R, int U, int PRO, bool GEGLU, bool F32OUT, int XC>
__global__ __launch_bounds__(1024) void matvec_kernel(const MatvecParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
    __shared__ float red_a[16 * XC], red_b[16 * XC];

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
    const int total_waves = gridDim.x * kMatvecWaves;
    const int n_rg = (N + R - 1) / R;
    const int wave_g = blockIdx.x * kMatvecWaves + wib;
    const int nrg_w = wave_g < n_rg ? (n_rg - 1 - wave_g) / total_waves + 1 : 0;
    const int T = nrg_w * S;                           // pipeline steps of this wave

    // ---- x / prologue operands first ----
    u32x4 px[XC], pnw[PRO != 0 ? XC : 1], ppw[PRO == 2 ? XC : 1], pres[PRO == 2 ? XC : 1];
#pragma unroll
    for (int k = 0; k < XC; ++k)
    {
        const size_t e = (size_t)min(tid + 1024 * k, nx16 - 1) * 8;
        px[k] = ld16(p.x + e);
        if constexpr (PRO != 0) pnw[k] = ld16(p.norm_w + e);
        if constexpr (PRO == 2)
        {
            ppw[k] = ld16(p.post_w + e);
            pres[k] = ld16(p.res + e);
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

    // r = bf16(bf16(res + a) * post_scale) on 8 elements (sandwich tail)
    auto tail8 = [&](const u32x4 a, const u32x4 rr) {
        u32x4 r;
#pragma unroll
        for (int d = 0; d < 4; ++d)
        {
            float lo = round_bf16(bf16_lo(rr[d]) + bf16_lo(a[d]));
            float hi = round_bf16(bf16_hi(rr[d]) + bf16_hi(a[d]));
            if (p.post_scale != 1.0f) { lo = lo * p.post_scale; hi = hi * p.post_scale; }
            r[d] = pack_bf16x2(lo, hi);
        }
        return r;
    };

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
            u32x4 r[XC];
#pragma unroll
            for (int k = 0; k < XC; ++k) r[k] = tail8(rms_apply8(px[k], ppw[k], rstd_a, 0.0f), pres[k]);
            if (blockIdx.x == 0)
            {
#pragma unroll
                for (int k = 0; k < XC; ++k)
                    if (tid + 1024 * k < nx16) st16(p.res_out + (size_t)(tid + 1024 * k) * 8, r[k]);
            }
            const float rstd_r = rstd_of(r, red_b);
#pragma unroll
            for (int k = 0; k < XC; ++k)
                if (tid + 1024 * k < nx16) xs[tid + 1024 * k] = rms_apply8(r[k], pnw[k], rstd_r, 0.0f);
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
                if constexpr (GEGLU)
                {
                    // unfused chain: gate/up stored as bf16 by the Linear, then
                    // bf16(gelu_tanh(gate) * up) by the GeGLU kernel
                    const float g = round_bf16(acc[2 * r]), up = round_bf16(acc[2 * r + 1]);
                    reinterpret_cast<uint16_t*>(p.y)[col] = f32_to_bf16_bits(gelu_tanh(g) * up);
                }
                else if constexpr (F32OUT)
                    reinterpret_cast<float*>(p.y)[col] = acc[r];
                else
                    reinterpret_cast<uint16_t*>(p.y)[col] = f32_to_bf16_bits(acc[r]);
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
}

// ---- host side ------------------------------------------------------------------------------
static int g_tune_R = 0, g_tune_U = 0, g_tune_blocks = 0;

template <int FMT, int R, int U, int PRO, bool GEGLU, bool F32OUT, int XC>
static int launch_xc(const MatvecParams& p, int max_blocks, hipStream_t s)
{
    const int n_rg = (p.N + R - 1) / R;
    constexpr int EPC = Fmt<FMT>::kElemsPerChunk;
    const int nchunks = p.K / EPC;
    const int S = (nchunks + 64 * U - 1) / (64 * U);
    const size_t lds = (size_t)S * 64 * U * EPC * 2;
    MILA_REQUIRE(lds <= 65536, "matvec: K=%d needs %zu bytes of LDS for x (limit 65536)", p.K, lds);
    int blocks = (n_rg + kMatvecWaves - 1) / kMatvecWaves;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((matvec_kernel<FMT, R, U, PRO, GEGLU, F32OUT, XC>), dim3(blocks), dim3(64 * kMatvecWaves), lds, s, p);
    MILA_LAUNCH_CHECK("matvec");
}

template <int FMT, int R, int U, int PRO, bool GEGLU, bool F32OUT>
static int launch(const MatvecParams& p, int max_blocks, hipStream_t s)
{
    if (p.K <= 8192) return launch_xc<FMT, R, U, PRO, GEGLU, F32OUT, 1>(p, max_blocks, s);
    if (p.K <= 16384) return launch_xc<FMT, R, U, PRO, GEGLU, F32OUT, 2>(p, max_blocks, s);
    return set_error(MILA_E_INVALID_ARGUMENT, "matvec: K=%d exceeds the register-staged x limit (16384)", p.K);
}

template <int FMT, int PRO, bool GEGLU, bool F32OUT>
static int dispatch_RU(const MatvecParams& p, hipStream_t s)
{
    // Launch shape (measured on MI355X with tools/bench_matvec.py): weight rows per wave step
    // `rows_per_step` and chunk positions per step U by weight format and matrix height.  Small matrices want
    // the most waves (1 row each) and, for bf16, the deepest per-wave stream; tall ones amortise the LDS x
    // reads over more rows and take two workgroups per CU so the hardware balances the tail.
    const int rows = GEGLU ? 2 * p.N : p.N;
    const bool tall = rows >= 16384, huge = rows >= 100000;
    int rows_per_step, U;
    if (FMT == FMT_BF16) { rows_per_step = tall ? 4 : 1; U = tall ? 2 : 4; }
    else if (FMT == FMT_FP8) { rows_per_step = huge ? 4 : (tall ? 2 : 1); U = 2; }
    else { rows_per_step = tall ? 2 : 1; U = huge ? 2 : 1; }
    int R = GEGLU ? (rows_per_step >= 2 ? rows_per_step / 2 : 1) : rows_per_step;
    int max_blocks = tall ? 2 * kNumCU : kNumCU;
    if (g_tune_R > 0) R = g_tune_R;
    if (g_tune_U > 0) U = g_tune_U;
    if (g_tune_blocks > 0) max_blocks = g_tune_blocks;
    if (GEGLU && R > 2) R = 2;
    if (R == 1 && U == 1) return launch<FMT, 1, 1, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if (R == 1 && U == 2) return launch<FMT, 1, 2, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if (R == 1) return launch<FMT, 1, 4, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if (R == 2 && U == 1) return launch<FMT, 2, 1, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if (R == 2) return launch<FMT, 2, 2, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if constexpr (!GEGLU) return launch<FMT, 4, 2, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    return launch<FMT, 2, 2, PRO, GEGLU, F32OUT>(p, max_blocks, s);
}

static int validate(const char* who, const void* y, const void* x, const void* W, const float* scales,
                    int fmt, int K, int N, int group)
{
    MILA_REQUIRE(y && x && W, "%s: null pointer (y=%p x=%p W=%p)", who, y, x, W);
    MILA_REQUIRE(K > 0 && N > 0, "%s: K and N must be positive (K=%d N=%d)", who, K, N);
    MILA_REQUIRE(K <= 16384, "%s: K=%d exceeds the register-staged x limit (16384)", who, K);
    if (fmt == FMT_BF16) MILA_REQUIRE(K % 8 == 0, "%s: K=%d must be a multiple of 8 for bf16 weights", who, K);
    if (fmt == FMT_FP8)
    {
        MILA_REQUIRE(K % 16 == 0, "%s: K=%d must be a multiple of 16 for fp8 weights", who, K);
        MILA_REQUIRE(scales != nullptr, "%s: fp8 weights need per-channel scales", who);
    }
    if (fmt == FMT_FP4)
    {
        MILA_REQUIRE(group == 64 || group == 128, "%s: fp4 group size must be 64 or 128 (got %d)", who, group);
        MILA_REQUIRE(K % 32 == 0 && K % group == 0, "%s: K=%d must be a multiple of 32 and of the group size %d", who, K, group);
        MILA_REQUIRE(scales != nullptr, "%s: fp4 weights need per-group scales", who);
    }
    MILA_REQUIRE(fmt >= 0 && fmt <= 2, "%s: unknown weight format %d", who, fmt);
    return MILA_OK;
}

template <int PRO, bool GEGLU, bool F32OUT>
static int dispatch_fmt(int fmt, const MatvecParams& p, hipStream_t s)
{
    switch (fmt)
    {
        case FMT_BF16: return dispatch_RU<FMT_BF16, PRO, GEGLU, F32OUT>(p, s);
        case FMT_FP8: return dispatch_RU<FMT_FP8, PRO, GEGLU, F32OUT>(p, s);
        default: return dispatch_RU<FMT_FP4, PRO, GEGLU, F32OUT>(p, s);
    }
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_tune_matvec(int R, int U, int max_blocks)
{
    g_tune_R = R;
    g_tune_U = U;
    g_tune_blocks = max_blocks;
    return MILA_OK;
}

int mila_cdna4_matvec_bf16(uint16_t* y, const uint16_t* x, const uint16_t* W, const uint16_t* bias, int K,
                           int N, mila_stream_t stream)
{
    int rc = validate("matvec_bf16", y, x, W, nullptr, FMT_BF16, K, N, 0);
    if (rc) return rc;
    MatvecParams p{y, x, reinterpret_cast<const uint8_t*>(W), nullptr, bias, nullptr, nullptr, nullptr, nullptr, 1.0f, 0.0f, K, N, 0};
    return dispatch_fmt<0, false, false>(FMT_BF16, p, as_stream(stream));
}

int mila_cdna4_matvec_bf16_qfp8(uint16_t* y, const uint16_t* x, const uint8_t* W, const float* scales,
                                const uint16_t* bias, int K, int N, mila_stream_t stream)
{
    int rc = validate("matvec_bf16_qfp8", y, x, W, scales, FMT_FP8, K, N, 0);
    if (rc) return rc;
    MatvecParams p{y, x, W, scales, bias, nullptr, nullptr, nullptr, nullptr, 1.0f, 0.0f, K, N, 0};
    return dispatch_fmt<0, false, false>(FMT_FP8, p, as_stream(stream));
}

int mila_cdna4_matvec_bf16_qfp4(uint16_t* y, const uint16_t* x, const uint8_t* W_packed, const float* scales,
                                const uint16_t* bias, int K, int N, int group, mila_stream_t stream)
{
    int rc = validate("matvec_bf16_qfp4", y, x, W_packed, scales, FMT_FP4, K, N, group);
    if (rc) return rc;
    MatvecParams p{y, x, W_packed, scales, bias, nullptr, nullptr, nullptr, nullptr, 1.0f, 0.0f, K, N, group};
    return dispatch_fmt<0, false, false>(FMT_FP4, p, as_stream(stream));
}

int mila_cdna4_matvec_f32out(float* y, const uint16_t* x, const void* W, const float* scales, int fmt, int K,
                             int N, int group, mila_stream_t stream)
{
    int rc = validate("matvec_f32out", y, x, W, scales, fmt, K, N, group);
    if (rc) return rc;
    MatvecParams p{y, x, reinterpret_cast<const uint8_t*>(W), scales, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, 0.0f, K, N, group};
    return dispatch_fmt<0, false, true>(fmt, p, as_stream(stream));
}

int mila_cdna4_fused_norm_matvec(const mila_fused_matvec_args* a, mila_stream_t stream)
{
    MILA_REQUIRE(a != nullptr, "fused_norm_matvec: null args");
    const int rows = a->geglu ? 2 * a->N : a->N;
    (void)rows;
    int rc = validate("fused_norm_matvec", a->y, a->x, a->W, a->scales, a->fmt, a->K, a->N, a->group);
    if (rc) return rc;
    MILA_REQUIRE(a->norm_w != nullptr, "fused_norm_matvec: norm_w is required");
    MILA_REQUIRE((a->res == nullptr) == (a->post_w == nullptr), "fused_norm_matvec: res and post_w go together");
    if (a->res)
    {
        MILA_REQUIRE(a->res_out != nullptr, "fused_norm_matvec: res_out is required with res");
        MILA_REQUIRE(a->res_out != a->res && a->res_out != a->x, "fused_norm_matvec: res_out must not alias res or x");
    }
    MatvecParams p{a->y, a->x, reinterpret_cast<const uint8_t*>(a->W), a->scales, nullptr, a->norm_w, a->post_w,
                   a->res, a->res_out, a->post_scale, a->eps, a->K, a->N, a->group};
    hipStream_t s = as_stream(stream);
    MILA_REQUIRE(!(a->geglu && a->f32_out), "fused_norm_matvec: geglu and f32_out are exclusive");
    if (a->f32_out)
        return a->res ? dispatch_fmt<2, false, true>(a->fmt, p, s) : dispatch_fmt<1, false, true>(a->fmt, p, s);
    if (a->res)
        return a->geglu ? dispatch_fmt<2, true, false>(a->fmt, p, s) : dispatch_fmt<2, false, false>(a->fmt, p, s);
    return a->geglu ? dispatch_fmt<1, true, false>(a->fmt, p, s) : dispatch_fmt<1, false, false>(a->fmt, p, s);
}

}  // extern "C"
