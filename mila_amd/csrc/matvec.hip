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
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "internal.h"
#include "rms_common.h"
#include "matvec_body.h"

namespace mila {

// PRO / R / U / XC: see matvec_body.h.
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
// One 1024-thread workgroup (16 waves) per CU: x is staged (and the prologue computed) once per CU, so the
// L2 -> LDS staging traffic is 256 * 2K bytes whatever the weight format.
template <int FMT, int R, int U, int PRO, bool GEGLU, bool F32OUT, int XC, int XSRC = X_PLAIN>
__global__ __launch_bounds__(1024) void matvec_kernel(const MatvecParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ float red_a[16 * XC], red_b[16 * XC];
    u32x4 rkeep[XC];
    matvec_body<FMT, R, U, PRO, GEGLU, F32OUT ? Y_F32 : Y_BF16, XC, XSRC, RES_MEM>(
        p, reinterpret_cast<u32x4*>(smem_raw), red_a, red_b, (int)blockIdx.x, (int)gridDim.x, rkeep, NoWait{});
}

// ---- host side ------------------------------------------------------------------------------
static int g_tune_R = 0, g_tune_U = 0, g_tune_blocks = 0;      // named tuning variables (0 = the default rule): rows per wave, chunk positions in flight, maximum workgroups
MILA_TUNE("matvec.rows_per_wave", g_tune_R);
MILA_TUNE("matvec.chunks_in_flight", g_tune_U);
MILA_TUNE("matvec.max_workgroups", g_tune_blocks);
static thread_local int t_last_matvec_blocks = 0;      // workgroups of this thread's last matvec launch (= the sampler partials a lm_head launch wrote)

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
    t_last_matvec_blocks = blocks;
    note_form("matvec");
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

struct ShapeOverrides
{
    struct E { bool set = false; int rows = 1, U = 2, blocks = kNumCU; } e[3][3];
};
static ShapeOverrides parse_shape_overrides()
{
    ShapeOverrides o;
    const char* env = getenv("MILA_MATVEC_SHAPE");
    if (!env) return o;
    int f, t, r, u, b, n = 0;
    while (*env && sscanf(env, "f%dt%d=%d,%d,%d%n", &f, &t, &r, &u, &b, &n) == 5)
    {
        if (f >= 0 && f < 3 && t >= 0 && t < 3 && r >= 1 && u >= 1 && b >= 1) { o.e[f][t].set = true; o.e[f][t].rows = r; o.e[f][t].U = u; o.e[f][t].blocks = b; }
        env += n;
        if (*env == ';') ++env;
    }
    return o;
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
    if (FMT == FMT_BF16) { rows_per_step = huge ? 4 : (tall ? 2 : 1); U = tall ? 2 : 4; }
    else if (FMT == FMT_FP8) { rows_per_step = huge ? 4 : (tall ? 2 : 1); U = 2; }
    else { rows_per_step = tall ? 2 : 1; U = huge ? 2 : 1; }
    // two workgroups per CU (dynamic tail balance) pay for the bf16 matrices and for the lm_head; the quantized
    // tall matrices (fc_gate_up) are faster with one resident workgroup per CU (in-situ sweep, tools/tune_shapes.sh)
    int max_blocks = (huge || (tall && FMT == FMT_BF16)) ? 2 * kNumCU : kNumCU;
    {
        // tuning hook (tools/, never set by the product path): MILA_MATVEC_SHAPE="f<fmt>t<0 short|1 tall|2 huge>=rows,U,blocks;..."
        static const ShapeOverrides ov = parse_shape_overrides();
        const ShapeOverrides::E& e = ov.e[FMT][huge ? 2 : (tall ? 1 : 0)];
        if (e.set) { rows_per_step = e.rows; U = e.U; max_blocks = e.blocks; }
    }
    int R = GEGLU ? (rows_per_step >= 2 ? rows_per_step / 2 : 1) : rows_per_step;
    if (g_tune_R > 0) R = g_tune_R;
    if (g_tune_U > 0) U = g_tune_U;
    if (g_tune_blocks > 0) max_blocks = g_tune_blocks;
    // the argmax epilogue writes one (value, index) partial per WORKGROUP into the sampler's scratch of kArgmaxPartials slots: the grid is capped BEFORE the launch
    // (the count was only checked afterwards -- a tuned or future workgroup rule beyond 512 would have overrun the scratch first)
    if (p.amax_v && max_blocks > kArgmaxPartials) max_blocks = kArgmaxPartials;
    if (GEGLU && R > 2) R = 2;
    if (R == 1 && U == 1) return launch<FMT, 1, 1, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if (R == 1 && U == 2) return launch<FMT, 1, 2, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if (R == 1) return launch<FMT, 1, 4, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if (R == 2 && U == 1) return launch<FMT, 2, 1, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if (R == 2) return launch<FMT, 2, 2, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    if constexpr (!GEGLU) return launch<FMT, 4, 2, PRO, GEGLU, F32OUT>(p, max_blocks, s);
    return launch<FMT, 2, 2, PRO, GEGLU, F32OUT>(p, max_blocks, s);
}

// o_proj with the attention combine as its prologue: the short-matrix launch shape of the format, x chunks per thread 1
template <int FMT, int U, int XSRC>
static int launch_combine_t(const MatvecParams& p, hipStream_t s)
{
    constexpr int EPC = Fmt<FMT>::kElemsPerChunk;
    const int nchunks = p.K / EPC;
    const int S = (nchunks + 64 * U - 1) / (64 * U);
    const size_t lds = (size_t)S * 64 * U * EPC * 2;
    MILA_REQUIRE(lds <= 65536, "matvec_attn_combine: K=%d needs %zu bytes of LDS for x (limit 65536)", p.K, lds);
    int blocks = (p.N + kMatvecWaves - 1) / kMatvecWaves;
    if (blocks > kNumCU) blocks = kNumCU;
    hipLaunchKernelGGL((matvec_kernel<FMT, 1, U, 0, false, false, 1, XSRC>), dim3(blocks), dim3(64 * kMatvecWaves), lds, s, p);
    MILA_LAUNCH_CHECK("matvec_attn_combine");
}
template <int XSRC>
static int launch_combine(int fmt, const MatvecParams& p, hipStream_t s)
{
    switch (fmt)
    {
        case FMT_BF16: return launch_combine_t<FMT_BF16, 4, XSRC>(p, s);
        case FMT_FP8: return launch_combine_t<FMT_FP8, 2, XSRC>(p, s);
        default: return launch_combine_t<FMT_FP4, 1, XSRC>(p, s);
    }
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

int mila_cdna4_matvec_bf16(uint16_t* y, const uint16_t* x, const uint16_t* W, const uint16_t* bias, int K,
                           int N, mila_stream_t stream)
{
    int rc = validate("matvec_bf16", y, x, W, nullptr, FMT_BF16, K, N, 0);
    if (rc) return rc;
    MatvecParams p{y, x, reinterpret_cast<const uint8_t*>(W), nullptr, bias, nullptr, nullptr, nullptr, nullptr, 1.0f, 0.0f, K, N, 0, 0, 0};
    return dispatch_fmt<0, false, false>(FMT_BF16, p, as_stream(stream));
}

int mila_cdna4_matvec_bf16_qfp8(uint16_t* y, const uint16_t* x, const uint8_t* W, const float* scales,
                                const uint16_t* bias, int K, int N, mila_stream_t stream)
{
    int rc = validate("matvec_bf16_qfp8", y, x, W, scales, FMT_FP8, K, N, 0);
    if (rc) return rc;
    MatvecParams p{y, x, W, scales, bias, nullptr, nullptr, nullptr, nullptr, 1.0f, 0.0f, K, N, 0, 0, 0};
    return dispatch_fmt<0, false, false>(FMT_FP8, p, as_stream(stream));
}

int mila_cdna4_matvec_bf16_qfp4(uint16_t* y, const uint16_t* x, const uint8_t* W_packed, const float* scales,
                                const uint16_t* bias, int K, int N, int group, mila_stream_t stream)
{
    int rc = validate("matvec_bf16_qfp4", y, x, W_packed, scales, FMT_FP4, K, N, group);
    if (rc) return rc;
    MatvecParams p{y, x, W_packed, scales, bias, nullptr, nullptr, nullptr, nullptr, 1.0f, 0.0f, K, N, group, 0, 0};
    return dispatch_fmt<0, false, false>(FMT_FP4, p, as_stream(stream));
}

int mila_cdna4_matvec_f32out(float* y, const uint16_t* x, const void* W, const float* scales, int fmt, int K,
                             int N, int group, mila_stream_t stream)
{
    int rc = validate("matvec_f32out", y, x, W, scales, fmt, K, N, group);
    if (rc) return rc;
    MatvecParams p{y, x, reinterpret_cast<const uint8_t*>(W), scales, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, 0.0f, K, N, group, 0, 0};
    return dispatch_fmt<0, false, true>(fmt, p, as_stream(stream));
}

int mila_cdna4_matvec_attn_combine(uint16_t* y, const void* partials, int splits, int NH, int HS, const void* W, const float* scales,
                                   int fmt, int N, int group, mila_stream_t stream)
{
    MILA_REQUIRE(partials != nullptr, "matvec_attn_combine: null partials");
    MILA_REQUIRE(HS == 256 || HS == 512, "matvec_attn_combine: head size %d must be 256 or 512 (a wave's 64 chunks span 2 or 1 heads)", HS);
    MILA_REQUIRE(NH > 0 && splits > 1 && splits <= 64, "matvec_attn_combine: need NH > 0 and 1 < splits <= 64 (NH=%d splits=%d)", NH, splits);
    const int K = NH * HS;
    MILA_REQUIRE(K <= 8192, "matvec_attn_combine: NH * HS = %d exceeds 8192", K);
    int rc = validate("matvec_attn_combine", y, partials, W, scales, fmt, K, N, group);
    if (rc) return rc;
    MatvecParams p{y, reinterpret_cast<const uint16_t*>(partials), reinterpret_cast<const uint8_t*>(W), scales, nullptr, nullptr, nullptr, nullptr,
                   nullptr, 1.0f, 0.0f, K, N, group, splits, NH};
    return HS == 512 ? launch_combine<X_COMBINE_1>(fmt, p, as_stream(stream)) : launch_combine<X_COMBINE_2>(fmt, p, as_stream(stream));
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
                   a->res, a->res_out, a->post_scale, a->eps, a->K, a->N, a->group, 0, 0};
    hipStream_t s = as_stream(stream);
    MILA_REQUIRE(!(a->geglu && a->f32_out), "fused_norm_matvec: geglu and f32_out are exclusive");
    if (a->argmax_scratch)
    {
        // the greedy sampler's first stage in the lm_head's epilogue: one (value, index) partial per workgroup, laid out as sample_argmax_fp32 lays its own
        // ([kArgmaxPartials] floats, then [kArgmaxPartials] ints); sample_argmax_final_advance reduces *argmax_blocks of them
        MILA_REQUIRE(a->f32_out, "fused_norm_matvec: argmax_scratch needs f32_out (the logits)");
        MILA_REQUIRE(a->argmax_blocks != nullptr, "fused_norm_matvec: argmax_blocks is required with argmax_scratch");
        if (a->argmax_scratch_bytes < mila_cdna4_sample_scratch_bytes())
            return set_error(MILA_E_SCRATCH_TOO_SMALL, "fused_norm_matvec: argmax scratch %zu bytes < required %zu", a->argmax_scratch_bytes, mila_cdna4_sample_scratch_bytes());
        p.amax_v = reinterpret_cast<float*>(a->argmax_scratch);
        p.amax_i = reinterpret_cast<int*>(p.amax_v + kArgmaxPartials);
    }
    if (a->f32_out)
    {
        rc = a->res ? dispatch_fmt<2, false, true>(a->fmt, p, s) : dispatch_fmt<1, false, true>(a->fmt, p, s);
        if (rc == MILA_OK && a->argmax_scratch)
        {
            if (t_last_matvec_blocks > kArgmaxPartials) return set_error(MILA_E_RUNTIME, "fused_norm_matvec: %d workgroups exceed the %d sampler partials", t_last_matvec_blocks, kArgmaxPartials);
            *a->argmax_blocks = t_last_matvec_blocks;
        }
        return rc;
    }
    if (a->res)
        return a->geglu ? dispatch_fmt<2, true, false>(a->fmt, p, s) : dispatch_fmt<2, false, false>(a->fmt, p, s);
    return a->geglu ? dispatch_fmt<1, true, false>(a->fmt, p, s) : dispatch_fmt<1, false, false>(a->fmt, p, s);
}

}  // extern "C"
