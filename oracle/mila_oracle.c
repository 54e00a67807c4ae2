/*
 * mila_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never measured as the
 * product).  See mila_oracle.h for the contract and the parity-pinning statement.
 *
 * Every function cites the reference file:line it follows; paths are relative to
 * /root/reference/Mila/Src/Dnn unless they start with Tests/.
 *   CPU/  = Compute/Devices/Cpu/Operations/
 *   OPS/  = Compute/Devices/Cuda/Operations/
 */
#include "mila_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===========================================================================================
 * Scalar formats
 * ========================================================================================= */

/* bf16 <- f32, round-to-nearest-even; what __float2bfloat16 / __float2bfloat16_rn do
 * (OPS/Linear/Kernels/MatVec/CudaMatVecBias.Bf16.cu:179). */
uint16_t orc_f32_to_bf16(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

float orc_bf16_to_f32(uint16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void orc_round_bf16_inplace(float* x, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) x[i] = orc_bf16_to_f32(orc_f32_to_bf16(x[i]));
}

void orc_f32_to_bf16_array(uint16_t* dst, const float* src, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) dst[i] = orc_f32_to_bf16(src[i]);
}

void orc_bf16_to_f32_array(float* dst, const uint16_t* src, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) dst[i] = orc_bf16_to_f32(src[i]);
}

/* OCP FP8 E4M3FN <- f32: RNE, saturate-to-finite, NaN -> 0x7f.  This is the semantics of the
 * __nv_fp8_e4m3(float) constructor used at
 * OPS/Linear/Kernels/Quantization/CudaFp8WeightQuantization.cu:120 (cuda_fp8.hpp:
 * __nv_cvt_float_to_fp8(x, __NV_SATFINITE, __NV_E4M3)).  Bias 7, 3 mantissa bits, max 448,
 * subnormal step 2^-9 (decoder: Tests/Dnn/Components/Linear/Linear.Cuda.cpp:929-951). */
uint8_t orc_f32_to_e4m3(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint8_t)(sign | 0x7fu);
    const double a = fabs((double)x);
    if (a == 0.0) return sign;
    if (a >= 464.0) return (uint8_t)(sign | 0x7eu); /* >= midpoint(448,480) or inf: saturate */

    int e;
    (void)frexp(a, &e);   /* a = f * 2^e, f in [0.5,1)  => a in [2^(e-1), 2^e) */
    int E = e - 1;        /* unbiased exponent of the leading one */
    if (E < -6)
    {
        /* subnormal range: multiples of 2^-9, RNE (nearbyint under default rounding) */
        const double q = nearbyint(ldexp(a, 9));
        const int m = (int)q;           /* 0..8; 8 == smallest normal 2^-6 == code 0x08 */
        return (uint8_t)(sign | (uint8_t)m);
    }
    double frac = ldexp(a, -E) - 1.0;                 /* [0,1) */
    int mant = (int)nearbyint(frac * 8.0);            /* 0..8, RNE (ties exact in double) */
    if (mant == 8) { mant = 0; E += 1; }
    int biased = E + 7;
    if (biased > 15 || (biased == 15 && mant == 7)) return (uint8_t)(sign | 0x7eu);
    return (uint8_t)(sign | (uint8_t)(biased << 3) | (uint8_t)mant);
}

/* Tests/Dnn/Components/Linear/Linear.Cuda.cpp:929-951 (decodeFp8E4M3); NaN code 0x7f/0xff. */
float orc_e4m3_to_f32(uint8_t b)
{
    const uint32_t sign = (b >> 7) & 1u, ex = (b >> 3) & 0xfu, man = b & 7u;
    float mag;
    if (ex == 0xfu && man == 7u) return NAN;
    if (ex == 0) mag = ldexpf((float)man, -9);
    else mag = ldexpf(1.0f + (float)man / 8.0f, (int)ex - 7);
    return sign ? -mag : mag;
}

/* OPS/Linear/Kernels/Quantization/CudaFp4WeightQuantization.cu:54-70 (fp4_e2m1_quantize):
 * strict '<' breakpoints, sign from x < 0 (so -0.0f -> nibble 0). */
uint8_t orc_f32_to_e2m1(float x)
{
    const uint8_t sign = (x < 0.0f) ? 8u : 0u;
    const float a = fabsf(x);
    uint8_t mag;
    if (a < 0.25f) mag = 0;
    else if (a < 0.75f) mag = 1;
    else if (a < 1.25f) mag = 2;
    else if (a < 1.75f) mag = 3;
    else if (a < 2.5f) mag = 4;
    else if (a < 3.5f) mag = 5;
    else if (a < 5.0f) mag = 6;
    else mag = 7;
    return (uint8_t)(sign | mag);
}

/* OPS/Linear/Kernels/MatVec/CudaMatVecBias.Bf16.cu:20-25 (fp4_e2m1_decode). */
float orc_e2m1_to_f32(uint8_t nibble)
{
    static const float lut[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};
    const float mag = lut[nibble & 7u];
    return (nibble & 8u) ? -mag : mag;
}

/* ===========================================================================================
 * Reference CPU backend ops (FP32) -- line-faithful
 * ========================================================================================= */

/* The reference's own `#pragma omp` placements (the CPU ops of CPU/, enabled by MILA_ENABLE_OPENMP, OFF by default: CMakeLists.txt:78).
 * Compiled in only for libmila_oracle_omp.so (-fopenmp -DORC_OPENMP=1), the "all host cores" leg of bench.py's cpu_baseline;
 * each iteration computes exactly what the serial build computes, so results are identical.  Note what the placement in
 * CpuLinearOp means: the parallel loop runs over BATCH rows, so a batch-1 decode Linear stays on one core whatever the build. */
#ifdef ORC_OPENMP
#define ORC_OMP(x) _Pragma(#x)
#else
#define ORC_OMP(x)
#endif

/* CPU/CpuLinearOp.ixx:384-411 (forwardNaive): long double accumulation, bias added last. */
void orc_cpu_linear_naive(float* Y, const float* X, const float* W, const float* B,
                          int64_t batch, int64_t in_features, int64_t out_features)
{
    ORC_OMP(omp parallel for)                                       /* CpuLinearOp.ixx:389 */
    for (int64_t idx = 0; idx < batch; ++idx)
    {
        const int64_t in_base = idx * in_features;
        const int64_t out_base = idx * out_features;
        for (int64_t o = 0; o < out_features; ++o)
        {
            long double acc = 0.0L;
            for (int64_t i = 0; i < in_features; ++i)
                acc += (long double)X[in_base + i] * (long double)W[o * in_features + i];
            if (B) acc += (long double)B[o];
            Y[out_base + o] = (float)acc;
        }
    }
}

/* CPU/CpuLinearOp.ixx:418-456 (forwardUnrolled): LOOP_UNROLL = 8 rows at a time, float
 * accumulators seeded with the bias, product rounded then added (no fused multiply-add is
 * requested by the source; -ffp-contract=off in oracle/Makefile keeps it that way). */
void orc_cpu_linear_unrolled(float* Y, const float* X, const float* W, const float* B,
                             int64_t batch, int64_t in_features, int64_t out_features)
{
    enum { LOOP_UNROLL = 8 };
    ORC_OMP(omp parallel for)                                       /* CpuLinearOp.ixx:423 */
    for (int64_t out_idx = 0; out_idx < batch; out_idx += LOOP_UNROLL)
    {
        for (int64_t o = 0; o < out_features; ++o)
        {
            float result[LOOP_UNROLL];
            for (int r = 0; r < LOOP_UNROLL; ++r) result[r] = B ? B[o] : 0.0f;
            for (int64_t i = 0; i < in_features; ++i)
            {
                const float w = W[o * in_features + i];
                for (int r = 0; r < LOOP_UNROLL; ++r)
                    result[r] += X[(out_idx + r) * in_features + i] * w;
            }
            for (int r = 0; r < LOOP_UNROLL; ++r) Y[(out_idx + r) * out_features + o] = result[r];
        }
    }
}

/* CPU/CpuLinearOp.ixx:248-266: use_loop_unroll_ iff batch % LOOP_UNROLL == 0 (set in build). */
void orc_cpu_linear(float* Y, const float* X, const float* W, const float* B,
                    int64_t batch, int64_t in_features, int64_t out_features)
{
    if (batch % 8 == 0) orc_cpu_linear_unrolled(Y, X, W, B, batch, in_features, out_features);
    else orc_cpu_linear_naive(Y, X, W, B, batch, in_features, out_features);
}

/* CPU/CpuGeluOp.ixx:142-148; GELU_SCALING_FACTOR = sqrtf(2/pi) = 0.7978845608f. */
void orc_cpu_gelu(float* Y, const float* X, int64_t n)
{
    const float k = 0.7978845608f;
    ORC_OMP(omp parallel for if (n > 1000))                         /* CpuGeluOp.ixx:143 */
    for (int64_t i = 0; i < n; ++i)
    {
        const float x = X[i];
        const float cube = 0.044715f * x * x * x;
        Y[i] = 0.5f * x * (1.0f + tanhf(k * (x + cube)));
    }
}

/* CPU/CpuSoftmaxOp.ixx:167-215: float max, long double expl and sum, 1/sum multiply. */
void orc_cpu_softmax(float* Y, const float* X, int64_t outer, int64_t dim, int64_t inner)
{
    ORC_OMP(omp parallel for collapse(2))                           /* CpuSoftmaxOp.ixx:176 */
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t in = 0; in < inner; ++in)
        {
            const float* si = X + o * dim * inner + in;
            float* so = Y + o * dim * inner + in;
            float max_val = -INFINITY;
            for (int64_t i = 0; i < dim; ++i)
                if (si[i * inner] > max_val) max_val = si[i * inner];
            long double sum = 0.0L;
            for (int64_t i = 0; i < dim; ++i)
            {
                long double val = expl((long double)(si[i * inner] - max_val));
                so[i * inner] = (float)val;
                sum += val;
            }
            long double inv = 1.0L / sum;
            for (int64_t i = 0; i < dim; ++i) so[i * inner] = (float)((long double)so[i * inner] * inv);
        }
}

/* CPU/CpuLayerNormOp.ixx:187-258: long double mean, biased variance, 1/sqrt(v+eps). */
void orc_cpu_layernorm(float* Y, float* mean, float* rstd, const float* X, const float* w,
                       const float* b, int64_t outer, int64_t dim, int64_t inner, float eps)
{
    ORC_OMP(omp parallel for collapse(2) if ((size_t)outer * (size_t)inner > 100))      /* CpuLayerNormOp.ixx:214 */
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t in = 0; in < inner; ++in)
        {
            const float* si = X + o * dim * inner + in;
            float* so = Y + o * dim * inner + in;
            long double m = 0.0L;
            for (int64_t i = 0; i < dim; ++i) m += (long double)si[i * inner];
            m /= (long double)dim;
            long double v = 0.0L;
            for (int64_t i = 0; i < dim; ++i)
            {
                long double d = (long double)si[i * inner] - m;
                v += d * d;
            }
            v /= (long double)dim;
            long double s = 1.0L / sqrtl(v + (long double)eps);
            for (int64_t i = 0; i < dim; ++i)
            {
                long double n = s * ((long double)si[i * inner] - m);
                if (w) n *= (long double)w[i];
                if (b) n += (long double)b[i];
                so[i * inner] = (float)n;
            }
            if (mean) mean[o * inner + in] = (float)m;
            if (rstd) rstd[o * inner + in] = (float)s;
        }
}

/* CPU/CpuResidualOp.ixx:83-101. */
void orc_cpu_residual(float* Y, const float* A, const float* B, int64_t n)
{
    ORC_OMP(omp parallel for if (n > 1000))                         /* CpuResidualOp.ixx:96 */
    for (int64_t i = 0; i < n; ++i) Y[i] = A[i] + B[i];
}

/* CPU/CpuEncoderOp.ixx:255-330: wte[tok] + wpe[t]; destination row stride is the built maximum
 * sequence length (out_stride_T), positions come from the input; bad token ids are an error. */
int orc_cpu_lpe(float* Y, const int32_t* tokens, const float* wte, const float* wpe,
                int64_t B, int64_t T, int64_t C, int64_t out_stride_T, int64_t vocab)
{
    for (int64_t i = 0; i < B * T; ++i)      /* the reference throws std::out_of_range from inside its loop (:296-302); checked up front here */
        if (tokens[i] < 0 || tokens[i] >= vocab) return -1;
    ORC_OMP(omp parallel for collapse(2))                           /* CpuEncoderOp.ixx:289 */
    for (int64_t b = 0; b < B; ++b)
        for (int64_t t = 0; t < T; ++t)
        {
            const int32_t tok = tokens[b * T + t];
            float* out = Y + b * out_stride_T * C + t * C;
            const float* we = wte + (int64_t)tok * C;
            const float* wp = wpe + t * C;
            for (int64_t c = 0; c < C; ++c) out[c] = we[c] + wp[c];
        }
    return 0;
}

/* CPU/CpuAttentionOp.ixx:133-151,310-460: permute -> q.k * 1/sqrt(HS) (float) -> causal float
 * softmax with expsum>0 guard -> att.v (float, sum over ALL j, masked entries are 0) ->
 * unpermute. */
void orc_cpu_mha(float* Y, const float* X, int B, int T, int C, int NH)
{
    const int HS = C / NH;
    const float scale = 1.0f / sqrtf((float)HS);
    const size_t nq = (size_t)B * NH * T * HS, na = (size_t)B * NH * T * T;
    float* q = (float*)malloc(nq * 4);
    float* k = (float*)malloc(nq * 4);
    float* v = (float*)malloc(nq * 4);
    float* vo = (float*)malloc(nq * 4);
    float* pre = (float*)malloc(na * 4);
    float* att = (float*)malloc(na * 4);
    const int qkv = 3 * C;
    ORC_OMP(omp parallel for collapse(2))                           /* CpuAttentionOp.ixx:312 (permuteQKV) */
    for (int b = 0; b < B; b++)
        for (int h = 0; h < NH; h++)
            for (int t = 0; t < T; t++)
                for (int d = 0; d < HS; d++)
                {
                    const int emb = h * HS + d;
                    const size_t base = (size_t)(b * T + t) * qkv;
                    const size_t idx = ((size_t)(b * NH + h) * T + t) * HS + d;
                    q[idx] = X[base + emb];
                    k[idx] = X[base + C + emb];
                    v[idx] = X[base + 2 * C + emb];
                }
    /* :336 scores, :366 softmax, :410 att.v -- each `parallel for collapse(2..3)` over (b, h[, t]) in the reference; one region over
     * (b, h) here does the same per-head work on the same threads with two barriers fewer */
    ORC_OMP(omp parallel for collapse(2))
    for (int b = 0; b < B; b++)
        for (int h = 0; h < NH; h++)
        {
            const size_t so = (size_t)(b * NH + h) * T * T;
            const size_t ho = (size_t)(b * NH + h) * T * HS;
            for (int i = 0; i < T; i++)
                for (int j = 0; j < T; j++)
                {
                    float sum = 0.0f;
                    for (int d = 0; d < HS; d++) sum += q[ho + (size_t)i * HS + d] * k[ho + (size_t)j * HS + d];
                    pre[so + (size_t)i * T + j] = sum * scale;
                }
            for (int t = 0; t < T; t++)
            {
                const float* sr = pre + so + (size_t)t * T;
                float* ar = att + so + (size_t)t * T;
                float maxval = -INFINITY;
                for (int t2 = 0; t2 <= t; t2++) if (sr[t2] > maxval) maxval = sr[t2];
                float expsum = 0.0f;
                for (int t2 = 0; t2 <= t; t2++)
                {
                    float ev = expf(sr[t2] - maxval);
                    ar[t2] = ev;
                    expsum += ev;
                }
                const float inv = (expsum > 0.0f) ? (1.0f / expsum) : 0.0f;
                for (int t2 = 0; t2 <= t; t2++) ar[t2] *= inv;
                for (int t2 = t + 1; t2 < T; t2++) ar[t2] = 0.0f;
            }
            for (int i = 0; i < T; i++)
                for (int d = 0; d < HS; d++)
                {
                    float sum = 0.0f;
                    for (int j = 0; j < T; j++) sum += att[so + (size_t)i * T + j] * v[ho + (size_t)j * HS + d];
                    vo[ho + (size_t)i * HS + d] = sum;
                }
        }
    ORC_OMP(omp parallel for collapse(2))                           /* CpuAttentionOp.ixx:439 (unpermute) */
    for (int b = 0; b < B; b++)
        for (int i = 0; i < T; i++)
            for (int h = 0; h < NH; h++)
                for (int d = 0; d < HS; d++)
                    Y[(size_t)(b * T + i) * C + h * HS + d] = vo[((size_t)(b * NH + h) * T + i) * HS + d];
    free(q); free(k); free(v); free(vo); free(pre); free(att);
}

/* GptTransformer::forward (Components/Transformers/Gpt/GptTransformer.ixx:221-254) over
 * GptBlock::forward (GptBlock.ixx:148-184) and MLP::forward (Components/FFN/MLP/MLP.ixx:148-161).
 * LayerNorm eps 1e-5 (LayerNorm.Config.ixx:270); lm_head has no bias (GptTransformer.ixx:854-855).
 * params: [0]=wte[V,C] [1]=wpe[maxT,C]; per layer l (12 entries from 2+12*l):
 *   ln1.w ln1.b qkv.w[3C,C] qkv.b attn_out.w[C,C] attn_out.b ln2.w ln2.b fc1.w[4C,C] fc1.b
 *   fc2.w[C,4C] fc2.b;  then lnf.w lnf.b lm_head.w[V,C]. */
void orc_cpu_gpt2_forward(float* logits, const int32_t* tokens, const float* const* params,
                          int B, int T, int C, int L, int NH, int V, int maxT)
{
    const int64_t M = (int64_t)B * T;
    (void)maxT;
    float* x = (float*)malloc((size_t)M * C * 4);
    float* ln = (float*)malloc((size_t)M * C * 4);
    float* qkv = (float*)malloc((size_t)M * 3 * C * 4);
    float* att = (float*)malloc((size_t)M * C * 4);
    float* proj = (float*)malloc((size_t)M * C * 4);
    float* res1 = (float*)malloc((size_t)M * C * 4);
    float* h1 = (float*)malloc((size_t)M * 4 * C * 4);
    float* h2 = (float*)malloc((size_t)M * 4 * C * 4);
    orc_cpu_lpe(x, tokens, params[0], params[1], B, T, C, T, V);
    for (int l = 0; l < L; ++l)
    {
        const float* const* p = params + 2 + 12 * l;
        orc_cpu_layernorm(ln, NULL, NULL, x, p[0], p[1], M, C, 1, 1e-5f);
        orc_cpu_linear(qkv, ln, p[2], p[3], M, C, 3 * C);
        orc_cpu_mha(att, qkv, B, T, C, NH);
        orc_cpu_linear(proj, att, p[4], p[5], M, C, C);
        orc_cpu_residual(res1, x, proj, M * C);
        orc_cpu_layernorm(ln, NULL, NULL, res1, p[6], p[7], M, C, 1, 1e-5f);
        orc_cpu_linear(h1, ln, p[8], p[9], M, C, 4 * C);
        orc_cpu_gelu(h2, h1, M * 4 * C);
        orc_cpu_linear(proj, h2, p[10], p[11], M, 4 * C, C);
        orc_cpu_residual(x, res1, proj, M * C);
    }
    const float* const* pf = params + 2 + 12 * L;
    orc_cpu_layernorm(ln, NULL, NULL, x, pf[0], pf[1], M, C, 1, 1e-5f);
    orc_cpu_linear(logits, ln, pf[2], NULL, M, C, V);
    free(x); free(ln); free(qkv); free(att); free(proj); free(res1); free(h1); free(h2);
}

/* ===========================================================================================
 * Activations (Components/Activations/Activation/Kernels/ElementwiseActivation.h:41-75)
 * ========================================================================================= */
float orc_gelu_tanh(float x)
{
    const float cube = 0.044715f * x * x * x;
    return 0.5f * x * (1.0f + tanhf(0.7978845608f * (x + cube)));
}

float orc_silu(float x)
{
    const float s = 1.0f / (1.0f + expf(-x));
    return x * s;
}

/* OPS/Activations/Geglu/Kernels/Geglu.cu:42-61: row = [gate(0..H) | up(H..2H)], fp32 math. */
void orc_geglu(float* Y, const float* X, int64_t tokens, int64_t half)
{
    for (int64_t t = 0; t < tokens; ++t)
        for (int64_t c = 0; c < half; ++c)
            Y[t * half + c] = orc_gelu_tanh(X[t * 2 * half + c]) * X[t * 2 * half + half + c];
}

/* ===========================================================================================
 * RMSNorm (OPS/Normalizations/RmsNorm/Kernels/RmsNorm.Bf16.cu:47-72; host reference
 * Tests/Dnn/Components/Normalization/RmsNorm/RmsNorm.Cuda.cpp:52-75).  Sum of squares and the
 * normalisation in double; y = x * rstd * (w + offset) + b.  rstd is returned unrounded.
 * ========================================================================================= */
void orc_rmsnorm(float* Y, float* rstd, const float* X, const float* w, const float* b,
                 int64_t outer, int64_t dim, int64_t inner, float eps, float w_offset)
{
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t in = 0; in < inner; ++in)
        {
            const float* x = X + o * dim * inner + in;
            float* y = Y + o * dim * inner + in;
            double m2 = 0.0;
            for (int64_t i = 0; i < dim; ++i) m2 += (double)x[i * inner] * (double)x[i * inner];
            const double r = 1.0 / sqrt(m2 / (double)dim + (double)eps);
            if (rstd) rstd[o * inner + in] = (float)r;
            for (int64_t i = 0; i < dim; ++i)
            {
                const double ww = w ? ((double)w[i] + (double)w_offset) : 1.0;
                const double bb = b ? (double)b[i] : 0.0;
                y[i * inner] = (float)((double)x[i * inner] * r * ww + bb);
            }
        }
}

/* ===========================================================================================
 * RoPE (cache: OPS/Encodings/Rope/Kernels/Rope.Fp32.cu:27-59,288-321; rotation:
 * Rope.Bf16.cu:45-69).  theta_i = base^(-2i/head_dim) for i < rope_pairs, else (cos,sin)=(1,0);
 * angle = float(pos) * theta in FP32 exactly as the kernel does; cos/sin evaluated in double of
 * that FP32 angle (the reference uses cosf/sinf of the same FP32 angle).
 * ========================================================================================= */
void orc_rope_build_cache(float* cos_out, float* sin_out, int max_seq, int head_dim, float base,
                          int rotary_dim)
{
    const int half = head_dim / 2;
    const int pairs = (rotary_dim > 0 && rotary_dim < head_dim) ? rotary_dim / 2 : half;
    for (int pos = 0; pos < max_seq; ++pos)
        for (int i = 0; i < half; ++i)
        {
            const size_t idx = (size_t)pos * half + i;
            if (i < pairs)
            {
                const float theta = (float)pow((double)base, -2.0 * (double)i / (double)(half * 2));
                const float angle = (float)pos * theta;
                cos_out[idx] = (float)cos((double)angle);
                sin_out[idx] = (float)sin((double)angle);
            }
            else
            {
                cos_out[idx] = 1.0f;
                sin_out[idx] = 0.0f;
            }
        }
}

/* Half-split (NeoX) pairing (i, i + HS/2): r0 = x0 c - x1 s, r1 = x0 s + x1 c. */
void orc_rope_rotate(float* out, const float* in, const float* cos_c, const float* sin_c,
                     int64_t B, int64_t T, int64_t n_heads, int64_t head_dim, int64_t pos_offset)
{
    const int64_t half = head_dim / 2;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t t = 0; t < T; ++t)
            for (int64_t h = 0; h < n_heads; ++h)
            {
                const int64_t base = ((b * T + t) * n_heads + h) * head_dim;
                const int64_t pos = t + pos_offset;
                for (int64_t i = 0; i < half; ++i)
                {
                    const double c = cos_c[pos * half + i], s = sin_c[pos * half + i];
                    const double x0 = in[base + i], x1 = in[base + i + half];
                    out[base + i] = (float)(x0 * c - x1 * s);
                    out[base + i + half] = (float)(x0 * s + x1 * c);
                }
            }
}

/* ===========================================================================================
 * Quantize-on-load -- integer outputs, restated step by step in the reference's FP32 order
 * ========================================================================================= */

/* OPS/Linear/Kernels/Quantization/CudaFp8WeightQuantization.cu:57-121:
 * absmax over the row; scale = absmax>0 ? absmax/448 : 1; inv = 1.0f/scale;
 * q = e4m3( bf16->f32 * inv ).  All in FP32 with IEEE division. */
void orc_quantize_fp8_per_channel(uint8_t* dst, float* scales, const uint16_t* src_bf16,
                                  int64_t N, int64_t K)
{
    for (int64_t n = 0; n < N; ++n)
    {
        const uint16_t* row = src_bf16 + n * K;
        float absmax = 0.0f;
        for (int64_t i = 0; i < K; ++i) absmax = fmaxf(absmax, fabsf(orc_bf16_to_f32(row[i])));
        const float scale = (absmax > 0.0f) ? (absmax / 448.0f) : 1.0f;
        const float inv = 1.0f / scale;
        scales[n] = scale;
        for (int64_t i = 0; i < K; ++i)
        {
            volatile float v = orc_bf16_to_f32(row[i]) * inv;   /* one FP32 rounding, as on device */
            dst[n * K + i] = orc_f32_to_e4m3(v);
        }
    }
}

/* OPS/Linear/Kernels/Quantization/CudaFp4WeightQuantization.cu:82-144: per (row, group):
 * scale = absmax>0 ? absmax/6 : 1; inv = 1.0f/scale; nibble = e2m1(val*inv);
 * byte = n_even | n_odd << 4 (low nibble = even column).  scales [N, K/group]. */
void orc_quantize_fp4_per_group(uint8_t* dst_packed, float* scales, const uint16_t* src_bf16,
                                int64_t N, int64_t K, int group)
{
    const int64_t ng = K / group;
    for (int64_t n = 0; n < N; ++n)
        for (int64_t g = 0; g < ng; ++g)
        {
            const uint16_t* p = src_bf16 + n * K + g * group;
            float absmax = 0.0f;
            for (int i = 0; i < group; ++i) absmax = fmaxf(absmax, fabsf(orc_bf16_to_f32(p[i])));
            const float scale = (absmax > 0.0f) ? (absmax / 6.0f) : 1.0f;
            const float inv = 1.0f / scale;
            scales[n * ng + g] = scale;
            for (int i = 0; i < group; i += 2)
            {
                volatile float v0 = orc_bf16_to_f32(p[i]) * inv;
                volatile float v1 = orc_bf16_to_f32(p[i + 1]) * inv;
                const uint8_t n0 = orc_f32_to_e2m1(v0), n1 = orc_f32_to_e2m1(v1);
                dst_packed[n * (K / 2) + (g * group + i) / 2] = (uint8_t)(n0 | (n1 << 4));
            }
        }
}

/* OPS/Linear/Kernels/W4A16Gemm/CudaW4A16Gemm.cu:244-294: sB = max(max_group_scale,1e-12)*(6/448). */
float orc_fp8_weight_scale_from_groups(const float* group_scales, int64_t n)
{
    float m = 0.0f;
    for (int64_t i = 0; i < n; ++i) m = fmaxf(m, group_scales[i]);
    return fmaxf(m, 1e-12f) * (6.0f / 448.0f);
}

void orc_dequant_fp8(float* W, const uint8_t* q, const float* scales, int64_t N, int64_t K)
{
    for (int64_t n = 0; n < N; ++n)
        for (int64_t i = 0; i < K; ++i) W[n * K + i] = orc_e4m3_to_f32(q[n * K + i]) * scales[n];
}

void orc_dequant_fp4(float* W, const uint8_t* packed, const float* scales, int64_t N, int64_t K,
                     int group)
{
    const int64_t ng = K / group;
    for (int64_t n = 0; n < N; ++n)
        for (int64_t i = 0; i < K; ++i)
        {
            const uint8_t byte = packed[n * (K / 2) + i / 2];
            const uint8_t nib = (i & 1) ? (uint8_t)(byte >> 4) : (uint8_t)(byte & 0xf);
            W[n * K + i] = orc_e2m1_to_f32(nib) * scales[n * ng + i / group];
        }
}

/* CudaW4A16Gemm.cu:300-323 (W4A8 prefill staging): out[2b], out[2b+1] = e4m3(lut(nibble) * (group_scale * (1 / sB))), FP32 steps. */
void orc_upcast_fp4_to_fp8(uint8_t* out, const uint8_t* packed, const float* scales, float weight_fp8_scale, int64_t N, int64_t K,
                           int group)
{
    const int64_t ng = K / group;
    const float inv = 1.0f / weight_fp8_scale;
    for (int64_t n = 0; n < N; ++n)
        for (int64_t b = 0; b < K / 2; ++b)
        {
            const uint8_t byte = packed[n * (K / 2) + b];
            volatile float sc = scales[n * ng + (2 * b) / group] * inv;
            volatile float lo = orc_e2m1_to_f32((uint8_t)(byte & 0xf)) * sc, hi = orc_e2m1_to_f32((uint8_t)(byte >> 4)) * sc;
            out[n * K + 2 * b] = orc_f32_to_e4m3(lo);
            out[n * K + 2 * b + 1] = orc_f32_to_e4m3(hi);
        }
}

/* ===========================================================================================
 * Linear on bf16 activations (decode matvec + prefill GEMM share one mathematical definition)
 * ========================================================================================= */

/* OPS/Linear/Kernels/MatVec/CudaMatVecBias.Bf16.cu:134-181: y = sum x*w (+ bias), bias bf16. */
void orc_linear_bf16w(float* Y, const float* X, const uint16_t* W, const uint16_t* bias,
                      int64_t M, int64_t K, int64_t N)
{
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n)
        {
            double acc = 0.0;
            const uint16_t* w = W + n * K;
            const float* x = X + m * K;
            for (int64_t i = 0; i < K; ++i) acc += (double)x[i] * (double)orc_bf16_to_f32(w[i]);
            if (bias) acc += (double)orc_bf16_to_f32(bias[n]);
            Y[m * N + n] = (float)acc;
        }
}

/* CudaMatVecBias.Bf16.cu:198-251: y = scale[oc] * sum(x * float(w8)) + bias -- scale applied
 * once AFTER the reduction. */
void orc_linear_fp8w(float* Y, const float* X, const uint8_t* W, const float* scales,
                     const uint16_t* bias, int64_t M, int64_t K, int64_t N)
{
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = orc_e4m3_to_f32((uint8_t)i);
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n)
        {
            double acc = 0.0;
            const uint8_t* w = W + n * K;
            const float* x = X + m * K;
            for (int64_t i = 0; i < K; ++i) acc += (double)x[i] * (double)lut[w[i]];
            acc *= (double)scales[n];
            if (bias) acc += (double)orc_bf16_to_f32(bias[n]);
            Y[m * N + n] = (float)acc;
        }
}

/* CudaMatVecBias.Bf16.cu:271-343 and :376-508: y = sum_c x[c]*lut[nib]*scale[oc,c/G] + bias. */
void orc_linear_fp4w(float* Y, const float* X, const uint8_t* Wp, const float* scales,
                     const uint16_t* bias, int64_t M, int64_t K, int64_t N, int group)
{
    const int64_t ng = K / group;
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n)
        {
            double acc = 0.0;
            const float* x = X + m * K;
            for (int64_t g = 0; g < ng; ++g)
            {
                double sub = 0.0;
                for (int i = 0; i < group; ++i)
                {
                    const int64_t c = g * group + i;
                    const uint8_t byte = Wp[n * (K / 2) + c / 2];
                    const uint8_t nib = (c & 1) ? (uint8_t)(byte >> 4) : (uint8_t)(byte & 0xf);
                    sub += (double)x[c] * (double)orc_e2m1_to_f32(nib);
                }
                acc += sub * (double)scales[n * ng + g];
            }
            if (bias) acc += (double)orc_bf16_to_f32(bias[n]);
            Y[m * N + n] = (float)acc;
        }
}

/* OPS/Linear/Kernels/Fp8Prefill/CudaFp8Prefill.cu:116-163: per-token
 * scale = max(absmax,1e-12)/448; q = e4m3(x * (1/scale)). */
void orc_quantize_act_fp8_per_token(uint8_t* q, float* token_scales, const float* X, int64_t M,
                                    int64_t K)
{
    for (int64_t m = 0; m < M; ++m)
    {
        float absmax = 0.0f;
        for (int64_t i = 0; i < K; ++i) absmax = fmaxf(absmax, fabsf(X[m * K + i]));
        const float scale = fmaxf(absmax, 1e-12f) / 448.0f;
        const float inv = 1.0f / scale;
        token_scales[m] = scale;
        for (int64_t i = 0; i < K; ++i)
        {
            volatile float v = X[m * K + i] * inv;
            q[m * K + i] = orc_f32_to_e4m3(v);
        }
    }
}

/* CudaFp8Prefill.cu:191-211 epilogue: y = acc * (weight scale) * token_scale + bias, FP32. */
void orc_linear_fp8a_fp8w(float* Y, const uint8_t* Xq, const float* token_scales,
                          const uint8_t* Wq, const float* w_row_scale, float w_tensor_scale,
                          const uint16_t* bias, int64_t M, int64_t K, int64_t N)
{
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = orc_e4m3_to_f32((uint8_t)i);
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n)
        {
            double acc = 0.0;
            for (int64_t i = 0; i < K; ++i) acc += (double)lut[Xq[m * K + i]] * (double)lut[Wq[n * K + i]];
            const double ws = w_row_scale ? (double)w_row_scale[n] : (double)w_tensor_scale;
            acc = acc * ws * (double)token_scales[m];
            if (bias) acc += (double)orc_bf16_to_f32(bias[n]);
            Y[m * N + n] = (float)acc;
        }
}

/* ===========================================================================================
 * Attention
 * Mask: OPS/Attention/GQA/Kernels/Gqa.Prefill.Bf16.cu:76-81 (keys max(0,t-window+1)..t),
 * decode band Gqa.Decode.Bf16.cu:100-105 (same set); head map h -> h/(NH/NKV) (:93-98);
 * score = dot*scale before max/exp (:212).  Double math throughout.
 * ========================================================================================= */
void orc_gqa_attention(float* out, const float* q, const float* k, const float* v, int B, int Tq,
                       int Tk, int NH, int NKV, int HS, int pos_offset, int window, float scale)
{
    const int gs = NH / NKV;
    double* p = (double*)malloc((size_t)(Tk > 0 ? Tk : 1) * sizeof(double));
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < Tq; ++t)
            for (int h = 0; h < NH; ++h)
            {
                const int kvh = h / gs;
                const int pos = pos_offset + t;
                int lo = (window > 0) ? (pos - window + 1) : 0;
                if (lo < 0) lo = 0;
                int hi = pos;                      /* inclusive */
                if (hi > Tk - 1) hi = Tk - 1;
                const float* qq = q + (((size_t)b * Tq + t) * NH + h) * HS;
                float* o = out + (((size_t)b * Tq + t) * NH + h) * HS;
                double mx = -INFINITY;
                for (int j = lo; j <= hi; ++j)
                {
                    const float* kk = k + (((size_t)b * Tk + j) * NKV + kvh) * HS;
                    double s = 0.0;
                    for (int d = 0; d < HS; ++d) s += (double)qq[d] * (double)kk[d];
                    s *= (double)scale;
                    p[j] = s;
                    if (s > mx) mx = s;
                }
                double l = 0.0;
                for (int j = lo; j <= hi; ++j) { p[j] = exp(p[j] - mx); l += p[j]; }
                for (int d = 0; d < HS; ++d)
                {
                    double acc = 0.0;
                    for (int j = lo; j <= hi; ++j)
                        acc += p[j] * (double)v[(((size_t)b * Tk + j) * NKV + kvh) * HS + d];
                    o[d] = (float)(l > 0.0 ? acc / l : 0.0);
                }
            }
    free(p);
}

/* OPS/Attention/GQA/Kernels/Gqa.Cache.Bf16.cu:86-130: src [B,chunk,NKV,HS] ->
 * cache [B,NKV,capacity,HS] at row (start_pos + t) % capacity. */
void orc_kv_write(float* Kc, float* Vc, const float* k, const float* v, int B, int chunk, int NKV,
                  int HS, int start_pos, int capacity)
{
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < chunk; ++t)
            for (int n = 0; n < NKV; ++n)
            {
                const int row = (start_pos + t) % capacity;
                const size_t dst = (((size_t)b * NKV + n) * capacity + row) * HS;
                const size_t src = (((size_t)b * chunk + t) * NKV + n) * HS;
                memcpy(Kc + dst, k + src, (size_t)HS * 4);
                memcpy(Vc + dst, v + src, (size_t)HS * 4);
            }
}

void orc_kv_ring_to_linear(float* k_lin, const float* Kc, int B, int NKV, int HS, int capacity,
                           int first, int Tk)
{
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < Tk; ++j)
            for (int n = 0; n < NKV; ++n)
            {
                const int row = (first + j) % capacity;
                memcpy(k_lin + (((size_t)b * Tk + j) * NKV + n) * HS,
                       Kc + (((size_t)b * NKV + n) * capacity + row) * HS, (size_t)HS * 4);
            }
}

/* ===========================================================================================
 * Glue
 * ========================================================================================= */

/* OPS/Embeddings/Kernels/TokenEmbedding.Bf16.cu:23-39 (row gather) followed by the component's
 * scale: static_cast<T>(float(x) * s) (Components/Embeddings/TokenEmbedding.ixx:179-181,
 * Compute/Devices/Cuda/Tensors/Operations/Kernels/Math.Elementwise.cu:106-113).
 * scale_then_round_bf16 == 0 -> plain gather. */
int orc_embedding_gather(float* Y, const int32_t* tokens, const float* table, int64_t n_tok,
                         int64_t C, int64_t vocab, float scale_then_round_bf16)
{
    for (int64_t t = 0; t < n_tok; ++t)
    {
        const int32_t tok = tokens[t];
        if (tok < 0 || tok >= vocab) return -1;
        for (int64_t c = 0; c < C; ++c)
        {
            float v = table[(int64_t)tok * C + c];
            if (scale_then_round_bf16 != 0.0f)
                v = orc_bf16_to_f32(orc_f32_to_bf16(v * scale_then_round_bf16));
            Y[t * C + c] = v;
        }
    }
    return 0;
}

/* Compute/Devices/Cuda/Tensors/Operations/Kernels/Structural.cu:106 (split3): last-dim split. */
void orc_split3(float* a, float* b, float* c, const float* X, int64_t rows, int64_t na, int64_t nb,
                int64_t nc)
{
    const int64_t w = na + nb + nc;
    for (int64_t r = 0; r < rows; ++r)
    {
        memcpy(a + r * na, X + r * w, (size_t)na * 4);
        memcpy(b + r * nb, X + r * w + na, (size_t)nb * 4);
        if (nc) memcpy(c + r * nc, X + r * w + na + nb, (size_t)nc * 4);
    }
}

/* OPS/Sampling/Kernels/Sampling.cuh:46-57: cap * tanh(x / cap). */
float orc_softcap(float x, float cap) { return cap * tanhf(x / cap); }

/* ---- stochastic sampler ------------------------------------------------------------------------
 * Restates the truncation semantics of the reference's single-block multinomial kernel
 * (OPS/Sampling/Kernels/Sampling.cu:760-905) in closed form instead of its 40-step bisections:
 *   x_i   = (softcap > 0 ? softcap * tanhf(l_i / softcap) : l_i) / temperature            (:773-781, fp32)
 *   top-k : the bisection converges to hi = the smallest float with count(x >= hi) <= k, i.e. the survivors are
 *           the values STRICTLY greater than the (k+1)-th largest value (exactly k of them when the k-th and
 *           (k+1)-th differ; a tie across the boundary drops the whole tie)                (:803-829)
 *   e_i   = survivors: expf(x_i - max), others 0; total = sum                              (:831-841)
 *   top-p : lo converges to the largest probability p with mass(e >= p) > top_p * total: the nucleus is
 *           the smallest prefix of the probability-sorted survivors (ties grouped) whose mass exceeds the target
 *           (:843-880); total is recomputed over the nucleus
 *   draw  : the first index i, in token-index order, with e_i > 0 and cumulative >= r * total; vocab-1 otherwise (:882-903)
 * Sums are accumulated in double.  margins[0..2] (optional) report how far the three decisions are from flipping:
 *   [0] (a_k - a_{k+1}) / max(|a_k|, 1e-30)   [1] distance of the nucleus prefix masses to the target / total
 *   [2] distance of r * total to the nearest cumulative boundary of the chosen token / total.      Returns the token. */
typedef struct { float v; int i; } orc_vi;
static int orc_cmp_desc(const void* a, const void* b)
{
    const float x = ((const orc_vi*)a)->v, y = ((const orc_vi*)b)->v;
    if (x > y) return -1;
    if (x < y) return 1;
    return ((const orc_vi*)a)->i - ((const orc_vi*)b)->i;
}
int orc_sample_stochastic(const float* logits, int vocab, float softcap, float temperature, int top_k, float top_p, float r,
                          double* margins)
{
    float* x = (float*)malloc(sizeof(float) * (size_t)vocab);
    float* e = (float*)malloc(sizeof(float) * (size_t)vocab);
    orc_vi* srt = (orc_vi*)malloc(sizeof(orc_vi) * (size_t)vocab);
    double m0 = 1e30, m1 = 1e30, m2 = 1e30;
    float mx = -FLT_MAX;
    for (int i = 0; i < vocab; ++i)
    {
        float v = logits[i];
        if (softcap > 0.0f) v = softcap * tanhf(v / softcap);
        v = v / temperature;
        x[i] = v;
        if (v > mx) mx = v;
        srt[i].v = v; srt[i].i = i;
    }
    float kthr = -FLT_MAX;    /* survivors: x > kthr (or all) */
    int use_k = top_k > 0 && top_k < vocab;
    if (use_k)
    {
        qsort(srt, (size_t)vocab, sizeof(orc_vi), orc_cmp_desc);
        kthr = srt[top_k].v;                       /* (k+1)-th largest */
        const double ak = srt[top_k - 1].v;
        m0 = (ak - (double)kthr) / fmax(fabs(ak), 1e-30);
    }
    double total = 0.0;
    for (int i = 0; i < vocab; ++i)
    {
        e[i] = (!use_k || x[i] > kthr) ? expf(x[i] - mx) : 0.0f;
        total += e[i];
    }
    if (top_p < 1.0f)
    {
        for (int i = 0; i < vocab; ++i) { srt[i].v = e[i]; srt[i].i = i; }
        qsort(srt, (size_t)vocab, sizeof(orc_vi), orc_cmp_desc);
        const double target = (double)top_p * total;
        double mass = 0.0;
        float pthr = 0.0f;
        int j = 0;
        while (j < vocab && srt[j].v > 0.0f)
        {
            int g = j;
            const double before = mass;
            while (g < vocab && srt[g].v == srt[j].v) { mass += srt[g].v; ++g; }   /* a tie group enters as a whole */
            if (mass > target)
            {
                pthr = srt[j].v;
                m1 = fmin(fabs(mass - target), fabs(target - before)) / total;
                break;
            }
            j = g;
        }
        double t2 = 0.0;
        for (int i = 0; i < vocab; ++i)
        {
            if (e[i] < pthr) e[i] = 0.0f;
            t2 += e[i];
        }
        total = t2;
    }
    const double target = (double)r * total;
    double cum = 0.0;
    int result = vocab - 1;
    for (int i = 0; i < vocab; ++i)
    {
        const double before = cum;
        cum += e[i];
        if (e[i] > 0.0f && cum >= target)
        {
            result = i;
            m2 = fmin(fabs(cum - target), fabs(target - before)) / total;
            break;
        }
    }
    if (margins) { margins[0] = m0; margins[1] = m1; margins[2] = m2; }
    free(x); free(e); free(srt);
    return result;
}

