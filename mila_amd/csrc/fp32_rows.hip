// FP32 rows of the ops whose reference keeps an FP32 path "for validation and reference" (OPS/OperationTraits.Cuda.ixx:50-54, :108-126, :274-282):
// Linear (decode matvec + prefill GEMM), multi-head attention on packed QKV (forward + KV-cache prefill / decode), learned positional encoding, RoPE.
// With the FP32 rows that already existed (LayerNorm, Softmax, GELU, Residual, RMSNorm) BASELINE config 1's model -- GPT-2 124M in FP32 -- runs on the device,
// and the device can be compared with the reference's CPU backend at FP32 tolerance instead of bf16's.
//
// These are VALIDATION kernels, not performance kernels (as in the reference, whose FP32 GEMM is a plain tiled kernel beside cuBLASLt): plain fused multiply-adds
// in ascending K order, fp32 accumulate -- no matrix cores (v_mfma_f32_32x32x2_f32 would buy speed nobody measures and a second summation order to explain).
//
// replaces Linear/Kernels/MatVec/CudaMatVecBias.Fp32.cu:40, Linear/Kernels/MatMul/CudaMatMulFp32.cu:31-183, Attention/MHA/CudaMhaOp.ixx:145-380 (FP32 is the
// reference's only CUDA MHA row), Encodings/Lpe/Kernels/Lpe.Fp32.cu:33-124, Encodings/Rope/Kernels/Rope.Fp32.cu:288-321.
#include "common.h"
#include "attention_generic.h"

namespace mila {

// ---- Linear, M == 1: one wave per output row, lanes stride K in float4 ----
__global__ __launch_bounds__(256) void matvec_fp32_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                          int K, int N)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const float* w = W + (size_t)n * K;
    float acc = 0.0f;
    if ((K & 3) == 0)
    {
        for (int k = lane * 4; k < K; k += 256)
        {
            const f32x4 a = *reinterpret_cast<const f32x4*>(w + k), b = *reinterpret_cast<const f32x4*>(x + k);
            acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc); acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
        }
    }
    else
    {
        for (int k = lane; k < K; k += 64) acc = fmaf(w[k], x[k], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) y[n] = acc + (bias ? bias[n] : 0.0f);
}

// ---- Linear, M > 1: Y[M, N] = X[M, K] W[N, K]^T (+ bias) (+ tanh-GELU), 64 x 64 tile, 256 threads x (4 x 4) outputs, K in steps of 16 through LDS ----
constexpr int kF32Tile = 64, kF32K = 16;
__global__ __launch_bounds__(256) void gemm_fp32_kernel(float* __restrict__ Y, const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                        int M, int K, int N, int act)
{
    __shared__ float xs[kF32K][kF32Tile + 1], ws[kF32K][kF32Tile + 1];      // [k][row]: the inner product walks k, a thread's 4 rows / 4 columns are contiguous
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * kF32Tile, n0 = blockIdx.x * kF32Tile;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += kF32K)
    {
        // 64 rows x 16 k of each operand: 1024 elements, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int e = tid + 256 * i, row = e >> 4, k = e & 15;
            xs[k][row] = (m0 + row < M && k0 + k < K) ? X[(size_t)(m0 + row) * K + k0 + k] : 0.0f;
            ws[k][row] = (n0 + row < N && k0 + k < K) ? W[(size_t)(n0 + row) * K + k0 + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kF32K; ++k)
        {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = xs[k][ty * 4 + i]; b[i] = ws[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j] + (bias ? bias[n] : 0.0f);
            if (act) v = gelu_tanh_precise(v);
            Y[(size_t)m * N + n] = v;
        }
    }
}

// ---- learned positional encoding: Y[b, t, :] = wte[tok[b, t], :] + wpe[t, :] ----
__global__ __launch_bounds__(256) void lpe_fp32_kernel(float* __restrict__ Y, const int32_t* __restrict__ tokens, const float* __restrict__ wte, const float* __restrict__ wpe,
                                                       int T, int C, int out_stride_T, int vocab, int32_t* error_flag)
{
    const int bt = blockIdx.x, b = bt / T, t = bt % T;
    const int tok = tokens[bt];
    if (tok < 0 || tok >= vocab)
    {
        if (threadIdx.x == 0 && error_flag) atomicExch(error_flag, 1 + bt);
        return;
    }
    const float* we = wte + (size_t)tok * C;
    const float* wp = wpe + (size_t)t * C;
    float* dst = Y + ((size_t)b * out_stride_T + t) * C;
    for (int i = threadIdx.x; i < C; i += 256) dst[i] = we[i] + wp[i];
}

// ---- RoPE: half-split pairs (i, i + HS / 2), fp32 cos / sin tables; may run in place ----
__global__ __launch_bounds__(256) void rope_rotate_fp32_kernel(float* out, const float* in, const float* __restrict__ cos_c, const float* __restrict__ sin_c, int64_t total_pairs,
                                                               int half, int T, int n_heads, int pos_offset)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total_pairs; v += stride)
    {
        const int64_t bth = v / half;
        const int i = (int)(v % half);
        const int pos = (int)((bth / n_heads) % T) + pos_offset;
        const size_t base = (size_t)bth * half * 2;
        const float c = cos_c[(size_t)pos * half + i], s = sin_c[(size_t)pos * half + i];
        const float x0 = in[base + i], x1 = in[base + i + half];
        out[base + i] = x0 * c - x1 * s;
        out[base + i + half] = x0 * s + x1 * c;
    }
}

// ---- K / V rows of packed [B, T, 3C] projections into [B, NH, capacity, HS] caches ----
__global__ __launch_bounds__(256) void mha_kv_write_fp32_kernel(float* __restrict__ Kc, float* __restrict__ Vc, const float* __restrict__ QKV, int64_t total, int T, int C, int HS,
                                                                int start_pos, int capacity)
{
    const int NH = C / HS;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride)
    {
        const int c = (int)(i % C);
        const int64_t bt = i / C;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int h = c / HS, d = c - h * HS;
        const size_t dst = (((size_t)b * NH + h) * capacity + (start_pos + t)) * HS + d;
        const float* row = QKV + bt * 3 * (int64_t)C;
        Kc[dst] = row[C + c];
        Vc[dst] = row[2 * C + c];
    }
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_matvec_fp32(float* y, const float* x, const float* W, const float* bias, int K, int N, mila_stream_t stream)
{
    MILA_REQUIRE(y && x && W, "matvec_fp32: null pointer");
    MILA_REQUIRE(K > 0 && N > 0, "matvec_fp32: K and N must be positive (K=%d N=%d)", K, N);
    hipLaunchKernelGGL(matvec_fp32_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), y, x, W, bias, K, N);
    MILA_LAUNCH_CHECK("matvec_fp32");
}

int mila_cdna4_gemm_fp32(float* Y, const float* X, const float* W, const float* bias, int M, int K, int N, int act, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && W, "gemm_fp32: null pointer");
    MILA_REQUIRE(M > 0 && K > 0 && N > 0, "gemm_fp32: M, K, N must be positive (%d,%d,%d)", M, K, N);
    MILA_REQUIRE(act == 0 || act == 1, "gemm_fp32: act must be 0 (none) or 1 (tanh-GELU), got %d", act);
    MILA_REQUIRE((M + kF32Tile - 1) / kF32Tile <= 65535, "gemm_fp32: M=%d exceeds the grid", M);
    hipLaunchKernelGGL(gemm_fp32_kernel, dim3((N + kF32Tile - 1) / kF32Tile, (M + kF32Tile - 1) / kF32Tile), dim3(256), 0, as_stream(stream), Y, X, W, bias, M, K, N, act);
    MILA_LAUNCH_CHECK("gemm_fp32");
}

int mila_cdna4_lpe_fp32(float* Y, const int32_t* tokens, const float* wte, const float* wpe, int B, int T, int C, int out_stride_T, int vocab, int32_t* error_flag,
                        mila_stream_t stream)
{
    MILA_REQUIRE(Y && tokens && wte && wpe, "lpe_fp32: null pointer");
    MILA_REQUIRE(B > 0 && T > 0 && C > 0 && vocab > 0, "lpe_fp32: bad sizes");
    MILA_REQUIRE(out_stride_T >= T, "lpe_fp32: output row stride %d is shorter than T=%d", out_stride_T, T);
    hipLaunchKernelGGL(lpe_fp32_kernel, dim3(B * T), dim3(256), 0, as_stream(stream), Y, tokens, wte, wpe, T, C, out_stride_T, vocab, error_flag);
    MILA_LAUNCH_CHECK("lpe_fp32");
}

int mila_cdna4_rope_forward_fp32(float* Qout, float* Kout, const float* Qin, const float* Kin, const float* cos_cache, const float* sin_cache, int B, int T, int NH, int NKV,
                                 int HS, int pos_offset, int max_seq, mila_stream_t stream)
{
    MILA_REQUIRE(cos_cache && sin_cache, "rope_forward_fp32: null cache");
    MILA_REQUIRE((Qout && Qin) || (Kout && Kin), "rope_forward_fp32: nothing to rotate");
    MILA_REQUIRE(B > 0 && T > 0 && HS > 0 && HS % 2 == 0, "rope_forward_fp32: bad sizes (B=%d T=%d HS=%d)", B, T, HS);
    MILA_REQUIRE(pos_offset >= 0 && pos_offset + T <= max_seq, "rope_forward_fp32: positions [%d,%d) exceed the cache length %d", pos_offset, pos_offset + T, max_seq);
    const int half = HS / 2;
    for (int which = 0; which < 2; ++which)
    {
        float* o = which ? Kout : Qout;
        const float* in = which ? Kin : Qin;
        if (!o || !in) continue;
        const int heads = which ? NKV : NH;
        const int64_t tp = (int64_t)B * T * heads * half;
        int blocks = ceil_div(tp, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(rope_rotate_fp32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), o, in, cos_cache, sin_cache, tp, half, T, heads, pos_offset);
    }
    MILA_LAUNCH_CHECK("rope_forward_fp32");
}

int mila_cdna4_mha_fp32(float* Y, const float* QKV, int B, int T, int C, int NH, mila_stream_t stream)
{
    MILA_REQUIRE(Y && QKV, "mha_fp32: null pointer");
    MILA_REQUIRE(B > 0 && T > 0 && C > 0 && NH > 0 && C % NH == 0, "mha_fp32: bad sizes (B=%d T=%d C=%d NH=%d)", B, T, C, NH);
    const int HS = C / NH;
    GenericAttnParamsT<float> g{Y, QKV, QKV + C, QKV + 2 * C, (int64_t)T * 3 * C, (int64_t)3 * C, (int64_t)T * 3 * C, (int64_t)HS, (int64_t)3 * C,
                                B, T, NH, NH, HS, T, 0, 0, 1.0f / sqrtf((float)HS)};
    return launch_attn_generic_f32(g, as_stream(stream));
}

int mila_cdna4_mha_kv_write_fp32(float* Kc, float* Vc, const float* QKV, int B, int T, int C, int NH, int start_pos, int capacity, mila_stream_t stream)
{
    MILA_REQUIRE(Kc && Vc && QKV, "mha_kv_write_fp32: null pointer");
    MILA_REQUIRE(B > 0 && T > 0 && C > 0 && NH > 0 && C % NH == 0 && capacity > 0, "mha_kv_write_fp32: bad sizes (B=%d T=%d C=%d NH=%d capacity=%d)", B, T, C, NH, capacity);
    MILA_REQUIRE(start_pos >= 0 && start_pos + T <= capacity, "mha_kv_write_fp32: positions [%d, %d) do not fit the cache capacity %d", start_pos, start_pos + T, capacity);
    const int64_t total = (int64_t)B * T * C;
    int blocks = ceil_div(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mha_kv_write_fp32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), Kc, Vc, QKV, total, T, C, C / NH, start_pos, capacity);
    MILA_LAUNCH_CHECK("mha_kv_write_fp32");
}

int mila_cdna4_mha_decode_fp32(float* Y, const float* QKV, float* Kc, float* Vc, int B, int C, int NH, int capacity, int position, mila_stream_t stream)
{
    MILA_REQUIRE(Y && QKV && Kc && Vc, "mha_decode_fp32: null pointer");
    MILA_REQUIRE(B > 0 && C > 0 && NH > 0 && C % NH == 0 && capacity > 0, "mha_decode_fp32: bad sizes (B=%d C=%d NH=%d capacity=%d)", B, C, NH, capacity);
    MILA_REQUIRE(position >= 0 && position < capacity, "mha_decode_fp32: position %d out of range [0, %d)", position, capacity);     // CudaMhaOp.ixx:262-265
    const int HS = C / NH;
    int rc = mila_cdna4_mha_kv_write_fp32(Kc, Vc, QKV, B, 1, C, NH, position, capacity, stream);
    if (rc) return rc;
    GenericAttnParamsT<float> g{Y, QKV, Kc, Vc, (int64_t)3 * C, (int64_t)3 * C, (int64_t)NH * capacity * HS, (int64_t)capacity * HS, (int64_t)HS,
                                B, 1, NH, NH, HS, capacity, position, 0, 1.0f / sqrtf((float)HS)};
    return launch_attn_generic_f32(g, as_stream(stream));
}

}  // extern "C"
