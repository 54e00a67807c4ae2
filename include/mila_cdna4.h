/*
 * mila_cdna4.h -- C ABI of the MI355X (gfx950 / CDNA4) device backend for Mila's forward() hot path.
 *
 * This is the drop-in seam.  In the reference every device op ends in a free function of the form
 *     void cuda_<op>_<dtype>(T* out, const T* in, ..., int dims..., cudaStream_t)
 * declared in a .cuh header (SURVEY.md section 0 finding 3).  Each entry point below replaces one of
 * those launchers one-for-one: raw device pointers, plain ints, an opaque stream handle, nothing
 * retained after the call, no hidden allocation (scratch is passed in), all work enqueued on the
 * given stream.  The "replaces" notes cite the reference declaration (paths relative to
 * /root/reference/Mila/Src/Dnn/Compute/Devices/Cuda/Operations unless noted).
 *
 * Conventions
 *   - return value: MILA_OK (0) or a negative MILA_E_* code; mila_cdna4_last_error() returns the
 *     text of the last failure on the calling thread.  The reference throws std::invalid_argument /
 *     std::runtime_error / CudaException from the same checks; the C++ op classes in
 *     mila_amd/host rethrow these codes as the same exception types.
 *   - bf16 tensors are `uint16_t` bit patterns, fp8 (OCP E4M3FN) and packed fp4 (E2M1, low nibble =
 *     even column) are `uint8_t`, scales are `float`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - pointers are DEVICE pointers unless the parameter name starts with host_.
 */
#ifndef MILA_CDNA4_H
#define MILA_CDNA4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MILA_API __attribute__((visibility("default")))
#else
#define MILA_API
#endif

typedef void* mila_stream_t;

enum {
    MILA_OK = 0,
    MILA_E_INVALID_ARGUMENT = -1, /* reference: std::invalid_argument                     */
    MILA_E_UNSUPPORTED = -2,      /* reference: std::runtime_error("unsupported ...")     */
    MILA_E_RUNTIME = -3,          /* reference: CudaException from cudaCheck              */
    MILA_E_SCRATCH_TOO_SMALL = -4
};

/* ---------------------------------------------------------------------------------------------
 * Runtime (counterpart of ExecutionContext<Cuda>: ../CudaExecutionContext.ixx:106-368 and the
 * device memory resources).  Thin wrappers so that a host that does not include HIP headers (the
 * reference's C++23 module units) can own streams and device memory.
 * ------------------------------------------------------------------------------------------- */
MILA_API const char* mila_cdna4_last_error(void);
MILA_API int mila_cdna4_abi_version(void);
MILA_API int mila_cdna4_device_count(int* count);
MILA_API int mila_cdna4_set_device(int device);
/* name: at least 64 bytes */
MILA_API int mila_cdna4_device_info(int device, char* name, int* compute_units, size_t* hbm_bytes);
MILA_API int mila_cdna4_stream_create(mila_stream_t* stream);
MILA_API int mila_cdna4_stream_destroy(mila_stream_t stream);
MILA_API int mila_cdna4_stream_synchronize(mila_stream_t stream);
MILA_API int mila_cdna4_malloc(void** ptr, size_t bytes);
MILA_API int mila_cdna4_free(void* ptr);
MILA_API int mila_cdna4_host_alloc_pinned(void** host_ptr, size_t bytes);
MILA_API int mila_cdna4_host_free_pinned(void* host_ptr);
MILA_API int mila_cdna4_memcpy_h2d(void* dst, const void* host_src, size_t bytes, mila_stream_t stream);
MILA_API int mila_cdna4_memcpy_d2h(void* host_dst, const void* src, size_t bytes, mila_stream_t stream);
MILA_API int mila_cdna4_memcpy_d2d(void* dst, const void* src, size_t bytes, mila_stream_t stream);
MILA_API int mila_cdna4_memset_zero(void* dst, size_t bytes, mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Linear -- decode (M == 1) matvec.  y[n] = sum_k x[k] * W[n,k] (+ bias[n]); fp32 accumulate,
 * bf16 out.  replaces Linear/Kernels/Linear.cuh: cuda_matvec_decode_bf16 / _qfp8 / _qfp4
 * (kernels Linear/Kernels/MatVec/CudaMatVecBias.Bf16.cu:134-181, :198-251, :271-508).
 *   W layouts: bf16 [N,K]; fp8 [N,K] + scales[N] (scale applied once after the reduction);
 *   fp4 packed [N,K/2] + scales[N,K/group], group in {64,128}.
 *   K % 8 == 0 (bf16), K % 16 == 0 (fp8), K % 32 == 0 and K % group == 0 (fp4).
 * ------------------------------------------------------------------------------------------- */
MILA_API int mila_cdna4_matvec_bf16(uint16_t* y, const uint16_t* x, const uint16_t* W,
                                    const uint16_t* bias, int K, int N, mila_stream_t stream);
MILA_API int mila_cdna4_matvec_bf16_qfp8(uint16_t* y, const uint16_t* x, const uint8_t* W,
                                         const float* scales, const uint16_t* bias, int K, int N,
                                         mila_stream_t stream);
MILA_API int mila_cdna4_matvec_bf16_qfp4(uint16_t* y, const uint16_t* x, const uint8_t* W_packed,
                                         const float* scales, const uint16_t* bias, int K, int N,
                                         int group, mila_stream_t stream);
/* fp32 logits variant used for the lm_head so the 1e-3 relative bar is asserted on fp32
 * (weight format selected by `fmt`: 0 bf16, 1 fp8 per-channel, 2 fp4 per-group). */
MILA_API int mila_cdna4_matvec_f32out(float* y, const uint16_t* x, const void* W, const float* scales,
                                      int fmt, int K, int N, int group, mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Linear -- prefill (M > 1) GEMM  Y[M,N] = X[M,K] * W[N,K]^T (+ bias), bf16 in/out, fp32 MFMA
 * accumulate.  replaces the cuBLASLt NT plans (Linear/CudaLinearOp.ixx:798-824,
 * Common/CublasLtLinearPlan.ixx:307-381) + cuda_add_bias (Fp8Prefill/CudaFp8Prefill.cu:239-256),
 * and for quantized weights the 2-phase dequantize-to-scratch + GEMM path
 * (Linear/CudaLinearOp.ixx:597-644, :716-764) by dequantizing in registers inside the GEMM (same
 * arithmetic: weights rounded to bf16, fp32 accumulate).  K % 8 == 0 (bf16 weights), K % 16 == 0 (fp8), K % 32 == 0 (fp4).
 * ------------------------------------------------------------------------------------------- */
MILA_API int mila_cdna4_gemm_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W,
                                  const uint16_t* bias, int M, int K, int N, mila_stream_t stream);
/* Linear + tanh-GELU in one kernel (the GPT-2 MLP's fc_1 -> gelu, Components/FFN/MLP/MLP.ixx:148-161): Y = bf16(gelu(bf16(X W^T (+ bias)))), every rounding of the two
 * launches kept, so the result is bit-identical to gemm_bf16 followed by gelu_bf16 -- the [M, N] intermediate is neither written nor re-read. */
MILA_API int mila_cdna4_gemm_gelu_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K,
                                       int N, mila_stream_t stream);
/* The same GEMM (act 0) / GEMM + tanh-GELU (act 1) with a caller workspace -- the counterpart of the cuBLASLt workspace CudaLinearOp hands every plan
 * (Linear/CudaLinearOp.ixx:637-638, :817-818; CudaExecutionContext.ixx:337-383, 4 MiB there).  With it, short prompts and the remainders of long ones -- tile lists that
 * cover a fraction of the CUs -- split K over the idle ones (fp32 partials in the workspace, summed in a fixed order: results do not depend on timing; they may differ
 * from gemm_bf16's in the last bf16 bit, the fp32 sum being taken in another order).  gemm_workspace_bytes() is what this (M, K, N) needs (0: the call is gemm_bf16 /
 * gemm_gelu_bf16; never more than 32 MiB); a smaller or unaligned (16 bytes) workspace is MILA_E_SCRATCH_TOO_SMALL / MILA_E_INVALID_ARGUMENT. */
MILA_API size_t mila_cdna4_gemm_workspace_bytes(int M, int K, int N);
MILA_API int mila_cdna4_gemm_bf16_ws(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, int act, void* workspace,
                                     size_t workspace_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_gemm_bf16_w8a16(uint16_t* Y, const uint16_t* X, const uint8_t* W,
                                        const float* scales, const uint16_t* bias, int M, int K,
                                        int N, mila_stream_t stream);
MILA_API int mila_cdna4_gemm_bf16_w4a16(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed,
                                        const float* scales, const uint16_t* bias, int M, int K,
                                        int N, int group, mila_stream_t stream);

/* Linear + GeGLU in ONE kernel for the prefill fc_gate_up (Components/Transformers/Gemma/Gemma.Block.ixx:343-348: the
 * Linear writes [M, 2F] = [gate | up], the GeGLU kernel reads it back): Y[M, F] = bf16(gelu_tanh(bf16(gate)) * bf16(up)),
 * gate = X W[0:F]^T, up = X W[F:2F]^T.  A tile pairs 128 gate rows with the matching 128 up rows; results are
 * bit-identical to gemm_bf16 + geglu_bf16.  gemm_geglu_applicable() != 0 says the fused kernels serve (M, K, F) -- every shape they can run; the
 * _staged forms dequantize the whole [2F, K] weight to `scratch` (2 * F * K * 2 bytes) first.
 * gemm_geglu_preferred() != 0: for a caller that holds the gemm_bf16_ws workspace the fused form is also the FASTER choice; 0 where that caller's
 * gemm_bf16_ws (split-K) + geglu_bf16 pair wins (few-row prompts, short tile lists) -- RocmLinearOp / GemmaBlock ask this one. */
MILA_API int mila_cdna4_gemm_geglu_applicable(int M, int K, int F);
MILA_API int mila_cdna4_gemm_geglu_preferred(int M, int K, int F);
MILA_API int mila_cdna4_gemm_geglu_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F,
                                        mila_stream_t stream);
MILA_API int mila_cdna4_gemm_geglu_bf16_w8a16_staged(uint16_t* Y, const uint16_t* X, const uint8_t* W,
                                                     const float* scales, int M, int K, int F, void* scratch,
                                                     size_t scratch_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_gemm_geglu_bf16_w4a16_staged(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed,
                                                     const float* scales, int M, int K, int F, int group,
                                                     void* scratch, size_t scratch_bytes, mila_stream_t stream);

/* W4A8 prefill: the reference's DEFAULT prefill for the fp4 policy (Linear/CudaLinearOp.ixx:646-715, kUseFp8ActivationPrefillPath):
 *   sB   = max(max(group scales), 1e-12) * 6 / 448                       (fp4_weight_fp8_scale; once, at load: CudaW4A16Gemm.cu:244-294)
 *   W8   = e4m3( lut(nibble) * (group scale * (1 / sB)) )                 (upcast_fp4_to_fp8: CudaW4A16Gemm.cu:300-323)
 *   s_m  = max(absmax(x_m), 1e-12) / 448,  X8 = e4m3( x * (1 / s_m) )     (quantize_fp8_per_token: Fp8Prefill/CudaFp8Prefill.cu:108-160)
 *   y    = bf16( float(bf16(sB * sum_k X8 W8)) * s_m + bias )             (fp8 x fp8 GEMM + cuda_fp8_apply_per_token_scales, :191-211)
 * The contraction runs on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit block scales, fp32 accumulate): the two LDS-DMA
 * GEMM kernels on the leading multiple of 256 rows where one serves the shape, a masked 128-row kernel on every other row -- the
 * same instruction chain per output element, so a row's bits do not depend on M.  Integer outputs (W8, X8) are bit-exact, y is
 * within 1 bf16 ulp of the restated reference.  As in the reference the path serves EVERY M > 1: gemm_fp8_applicable() is true for
 * any M, N > 0 with K % 16 == 0 (ABI 3; it used to require M % 256 == 0 and a full grid).  gemm_bf16_w4a8 runs the three steps with
 * scratch = [W8 | X8 | s_m] of gemm_w4a8_scratch_bytes(M, K, N) bytes. */
MILA_API int mila_cdna4_fp4_weight_fp8_scale(float* out_scale, const float* group_scales, int64_t num_scales,
                                             mila_stream_t stream);
MILA_API int mila_cdna4_upcast_fp4_to_fp8(uint8_t* out, const uint8_t* W_packed, const float* scales,
                                          const float* weight_fp8_scale, int N, int K, int group, mila_stream_t stream);
MILA_API int mila_cdna4_quantize_fp8_per_token(uint8_t* dst, float* token_scales, const uint16_t* X, int M, int K,
                                               mila_stream_t stream);
MILA_API int mila_cdna4_gemm_fp8_applicable(int M, int K, int N);
MILA_API int mila_cdna4_gemm_fp8_scaled(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* token_scales,
                                        const float* weight_scale, const uint16_t* bias, int M, int K, int N,
                                        mila_stream_t stream);
MILA_API size_t mila_cdna4_gemm_w4a8_scratch_bytes(int M, int K, int N);
MILA_API int mila_cdna4_gemm_bf16_w4a8(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed, const float* scales,
                                       const float* weight_fp8_scale, const uint16_t* bias, int M, int K, int N,
                                       int group, void* scratch, size_t scratch_bytes, mila_stream_t stream);
/* the same with the GeGLU epilogue (W_packed = [gate | up] rows, Y[M, F]): bit-identical to gemm_bf16_w4a8 + geglu_bf16;
 * scratch = gemm_w4a8_scratch_bytes(M, K, 2 F) */
MILA_API int mila_cdna4_gemm_geglu_w4a8_applicable(int M, int K, int F);
MILA_API int mila_cdna4_gemm_geglu_bf16_w4a8(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed, const float* scales,
                                             const float* weight_fp8_scale, int M, int K, int F, int group,
                                             void* scratch, size_t scratch_bytes, mila_stream_t stream);

/* Resident prefill weights (288 GB of HBM: the staging passes need not be repeated every forward as on the reference's 12 GB card).
 * dequantize_to_bf16 is the staging pass of the 2-phase prefill as an entry point of its own (cuda_fp8_dequantize_to_bf16,
 * Fp8Prefill/CudaFp8Prefill.cu:64-84; cuda_fp4_dequantize_to_bf16, W4A16Gemm/CudaW4A16Gemm.cu:210-235): out [N, K] bf16, the values
 * gemm_bf16_w8a16_staged / _w4a16_staged put in their scratch, so gemm_bf16 on `out` equals the staged call bit for bit.
 * gemm_geglu_fp8_scaled is gemm_geglu_bf16_w4a8 on operands staged by the caller (upcast_fp4_to_fp8 once at load,
 * quantize_fp8_per_token per forward). */
MILA_API int mila_cdna4_dequantize_to_bf16(uint16_t* out, const void* W, const float* scales, int fmt, int N, int K, int group,
                                           mila_stream_t stream);
/* gemm_fp8_scaled with a caller workspace (see gemm_bf16_ws: the cuBLASLt workspace of CudaLinearOp.ixx:637-638, :706-707): short prompts and long-prompt remainders
 * split K through it.  gemm_fp8_workspace_bytes() = what (M, K, N) needs (0: the call is gemm_fp8_scaled); gemm_w4a8_scratch_bytes() includes it for gemm_bf16_w4a8,
 * which therefore gives the same bits. */
MILA_API size_t mila_cdna4_gemm_fp8_workspace_bytes(int M, int K, int N);
MILA_API int mila_cdna4_gemm_fp8_scaled_ws(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* token_scales, const float* weight_scale, const uint16_t* bias,
                                           int M, int K, int N, void* workspace, size_t workspace_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_gemm_geglu_fp8_scaled(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* token_scales,
                                              const float* weight_scale, int M, int K, int F, mila_stream_t stream);

/* W8A8 prefill for PerChannelFp8<> weights (ABI 4; opt-in beside the W8A16 forms): the policy's own e4m3 [N, K] weights and fp32 scale[N] are the fp8 matrix cores'
 * operand as they lie in HBM -- the intent the reference states for this policy (../../../Quantization/Weight/Policies.ixx:39-40: "FP8 matmul consumes weights and scales
 * natively -- no dequantization on the forward hot path") and BASELINE config 4 names; no staging pass, no bf16 copy of the weights.  Activations per token exactly as on
 * the W4A8 path (quantize_fp8_per_token; Fp8Prefill/CudaFp8Prefill.cu:108-160):
 *   y = bf16( (sum_k X8 W8)[m, n] * channel_scales[n] * s_m + bias[n] )      fp32 throughout, ONE rounding (no intermediate bf16 tensor exists here)
 * The kernels, row-count rules, split-K workspace (gemm_fp8_workspace_bytes) and fused-GeGLU applicability (gemm_geglu_w4a8_applicable) are those of the W4A8 forms; a row's
 * bits do not depend on which tile form served it.  Not the reference's arithmetic for this policy (that is W8A16, CudaLinearOp.ixx:597-644, and stays RocmLinearOp's default):
 * the activation quantization puts the result within the reference's own W4A8 bar (1e-1 row_absmax, Tests/.../Linear.Cuda.cpp:760-774) of the W8A16 one.
 *   gemm_bf16_w8a8 / gemm_geglu_bf16_w8a8: one call from bf16 activations; scratch = [X8 | s_m | workspace] of gemm_w8a8_scratch_bytes(M, K, N) (GeGLU: N = 2 F) bytes.
 *   channel_scales: 16-byte aligned; the GeGLU forms take [2 F] = [gate rows | up rows]. */
MILA_API int mila_cdna4_gemm_fp8_w8a8_ws(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* token_scales, const float* channel_scales, const uint16_t* bias,
                                         int M, int K, int N, void* workspace, size_t workspace_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_gemm_geglu_fp8_w8a8(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* token_scales, const float* channel_scales, int M, int K, int F,
                                            mila_stream_t stream);
MILA_API size_t mila_cdna4_gemm_w8a8_scratch_bytes(int M, int K, int N);
MILA_API int mila_cdna4_gemm_bf16_w8a8(uint16_t* Y, const uint16_t* X, const uint8_t* W8, const float* channel_scales, const uint16_t* bias, int M, int K, int N,
                                       void* scratch, size_t scratch_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_gemm_geglu_bf16_w8a8(uint16_t* Y, const uint16_t* X, const uint8_t* W8, const float* channel_scales, int M, int K, int F, void* scratch,
                                             size_t scratch_bytes, mila_stream_t stream);

/* 2-phase forms for quantized weights (the reference's own structure, Linear/CudaLinearOp.ixx:597-644, :716-764:
 * dequantize to a bf16 scratch, then the bf16 GEMM).  Chosen automatically when the 256 x 256 LDS-DMA GEMM
 * applies to (M,K,N) -- gemm_staging_bytes() says how much scratch that needs (0 = the register-dequantizing kernel is
 * used and no scratch is touched).  Same arithmetic as the fused forms: w = bf16(decode(q) * scale), fp32 accumulate.
 * The scratch (16-byte aligned) holds the N K bf16 weights and, behind them, gemm_workspace_bytes(M, K, N) of split-K workspace: a staged call gives the bits of
 * gemm_bf16_ws on weights dequantized ahead of time (dequantize_to_bf16). */
MILA_API size_t mila_cdna4_gemm_staging_bytes(int M, int K, int N);
MILA_API int mila_cdna4_gemm_bf16_w8a16_staged(uint16_t* Y, const uint16_t* X, const uint8_t* W, const float* scales,
                                               const uint16_t* bias, int M, int K, int N, void* scratch,
                                               size_t scratch_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_gemm_bf16_w4a16_staged(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed,
                                               const float* scales, const uint16_t* bias, int M, int K, int N,
                                               int group, void* scratch, size_t scratch_bytes, mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Linear -- quantize-on-load.  Integer outputs are bit-exact with the reference kernels.
 * replaces cuda_quantize_fp8_per_channel (Linear/Kernels/Quantization/CudaFp8WeightQuantization.cu:
 * 57-121,209-249) and cuda_quantize_fp4_per_group (CudaFp4WeightQuantization.cu:54-144,184-224).
 * ------------------------------------------------------------------------------------------- */
MILA_API int mila_cdna4_quantize_fp8_per_channel(uint8_t* dst, float* scales, const uint16_t* src_bf16,
                                                 int N, int K, mila_stream_t stream);
MILA_API int mila_cdna4_quantize_fp4_per_group(uint8_t* dst_packed, float* scales,
                                               const uint16_t* src_bf16, int N, int K, int group,
                                               mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Attention over a KV cache [B, NKV, capacity, HS] (bf16), row = abs_pos % capacity.
 * replaces Attention/GQA/Kernels/CudaGqa.cuh: cuda_kvcache_write_kv_bf16 (Gqa.Cache.Bf16.cu:86),
 * cuda_gqa_decode_attention_bf16 (+ fixup; Gqa.Decode.Bf16.cu:64-428) and the flash-prefill
 * launchers (Gqa.Flash.Wmma.cu:246, Gqa.Flash.Fa2.cu:177).
 *   decode: q [B, NH*HS], len = position + 1, keys [max(0,len-window), len) (window 0 = all).
 *   prefill: q [B, chunk, NH*HS]; query t has absolute position pos_offset + t and sees keys
 *            max(0,pos-window+1) .. pos.  The cache must already contain the chunk (kv_write first).
 *   scale multiplies q.k before max/exp (Gemma passes 1.0).  HS in {64,128,256,512} on the MFMA / split-K kernels; any other HS <= 512 on a generic
 *   kernel (the unfused entry points only; the fused and device-position forms take the four sizes).
 * ------------------------------------------------------------------------------------------- */
MILA_API int mila_cdna4_kv_write_bf16(uint16_t* Kc, uint16_t* Vc, const uint16_t* k, const uint16_t* v,
                                      int B, int chunk, int NKV, int HS, int start_pos, int capacity,
                                      mila_stream_t stream);
MILA_API size_t mila_cdna4_attn_decode_scratch_bytes(int B, int NH, int HS);
MILA_API int mila_cdna4_attn_decode_bf16(uint16_t* Y, const uint16_t* Q, const uint16_t* Kc,
                                         const uint16_t* Vc, void* scratch, size_t scratch_bytes,
                                         int B, int NH, int NKV, int HS, int capacity, int len,
                                         int window, float scale, mila_stream_t stream);
MILA_API int mila_cdna4_attn_prefill_bf16(uint16_t* Y, const uint16_t* Q, const uint16_t* Kc,
                                          const uint16_t* Vc, int B, int chunk, int NH, int NKV, int HS,
                                          int capacity, int pos_offset, int window, float scale,
                                          mila_stream_t stream);
/* GPT-2 multi-head attention on packed QKV [B,T,3C] -> [B,T,C], causal, scale 1/sqrt(HS).
 * replaces Attention/MHA/CudaMhaOp.ixx:456-553 (permute + 2 batched GEMMs + softmax + unpermute). */
MILA_API int mila_cdna4_mha_bf16(uint16_t* Y, const uint16_t* QKV, int B, int T, int C, int NH,
                                 mila_stream_t stream);
/* The same op over a KV cache (CudaMhaOp.ixx:137-380: IPositionalUnaryOp::prefill / decode + IKvCacheLifecycle).  Cache [B, NH, capacity, HS] bf16.
 *   mha_kv_write: K / V of the packed rows [B, T, 3C] into cache rows start_pos .. start_pos + T - 1 (the K / V half of permute_qkv[_decode]);
 *     prefill = mha_bf16 + mha_kv_write(start_pos 0).
 *   mha_decode: QKV [B, 1, 3C] of the token at `position`: appends its K / V, then Y[B, C] = attention of its q over keys 0 .. position
 *     (replaces permute_qkv_decode + 2 cuBLASLt GEMMs + softmax_decode + unpermute, :276-380).  scratch: mha_decode_scratch_bytes(B, C, NH).
 *   Any head size: 64 / 128 / 256 / 512 run the MFMA flash and the split-K flash-decode kernels, every other HS <= 512 a generic kernel. */
MILA_API int mila_cdna4_mha_kv_write_bf16(uint16_t* Kc, uint16_t* Vc, const uint16_t* QKV, int B, int T, int C, int NH,
                                          int start_pos, int capacity, mila_stream_t stream);
MILA_API size_t mila_cdna4_mha_decode_scratch_bytes(int B, int C, int NH);
MILA_API int mila_cdna4_mha_decode_bf16(uint16_t* Y, const uint16_t* QKV, uint16_t* Kc, uint16_t* Vc, void* scratch,
                                        size_t scratch_bytes, int B, int C, int NH, int capacity, int position,
                                        mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Normalisation / activations.
 * rmsnorm: y = x * rsqrt(mean(x^2) + eps) * (w + w_offset) + b over `dim` strided by `inner`;
 *   rstd (bf16, may be NULL) gets one value per slice.  replaces
 *   Normalizations/RmsNorm/Kernels/RmsNorm.cuh:84-149 (RmsNorm.Bf16.cu:20-73).
 * layernorm: replaces Normalizations/LayerNorm/Kernels (LayerNorm.Fp32.cu:20,206), bf16 row added.
 * softmax:   replaces Normalizations/Softmax/Kernels/Softmax.Fp32.cu:56-88.
 * gelu/geglu: replaces Activations/Gelu/Kernels/Gelu.Fp32.cu:29-40, Activations/Geglu/Kernels/
 *   Geglu.cu:25-87 (row = [gate | up], y = gelu_tanh(gate) * up).
 * residual:  replaces Residual/Kernels/Residual.Bf16.cu:15-40.
 * ------------------------------------------------------------------------------------------- */
MILA_API int mila_cdna4_rmsnorm_bf16(uint16_t* Y, uint16_t* rstd, const uint16_t* X, const uint16_t* w,
                                     const uint16_t* b, int outer, int inner, int dim, float eps,
                                     float w_offset, mila_stream_t stream);
/* fp32 row (RmsNorm.cuh:84-123, cuda_rmsnorm_forward_fp32): same arguments, rstd fp32 */
MILA_API int mila_cdna4_rmsnorm_fp32(float* Y, float* rstd, const float* X, const float* w, const float* b,
                                     int outer, int inner, int dim, float eps, float w_offset,
                                     mila_stream_t stream);
MILA_API int mila_cdna4_layernorm_bf16(uint16_t* Y, float* mean, float* rstd, const uint16_t* X,
                                       const uint16_t* w, const uint16_t* b, int outer, int dim,
                                       float eps, mila_stream_t stream);
MILA_API int mila_cdna4_layernorm_fp32(float* Y, float* mean, float* rstd, const float* X,
                                       const float* w, const float* b, int outer, int dim, float eps,
                                       mila_stream_t stream);
MILA_API int mila_cdna4_softmax_fp32(float* Y, const float* X, int outer, int dim, int inner,
                                     mila_stream_t stream);
MILA_API int mila_cdna4_softmax_bf16(uint16_t* Y, const uint16_t* X, int outer, int dim, int inner,
                                     mila_stream_t stream);
MILA_API int mila_cdna4_gelu_bf16(uint16_t* Y, const uint16_t* X, int64_t n, mila_stream_t stream);
MILA_API int mila_cdna4_gelu_fp32(float* Y, const float* X, int64_t n, mila_stream_t stream);
MILA_API int mila_cdna4_geglu_bf16(uint16_t* Y, const uint16_t* X, int tokens, int half,
                                   mila_stream_t stream);
MILA_API int mila_cdna4_residual_bf16(uint16_t* Y, const uint16_t* A, const uint16_t* B, int64_t n,
                                      mila_stream_t stream);
MILA_API int mila_cdna4_residual_fp32(float* Y, const float* A, const float* B, int64_t n,
                                      mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * RoPE.  replaces Encodings/Rope/Kernels/Rope.cuh:29-182 (cache: Rope.Fp32.cu:27-59,288-321;
 * rotation: Rope.Bf16.cu:28-118).  Half-split pairing (i, i+HS/2); cos/sin fp32 [max_seq, HS/2];
 * pairs >= rotary_dim/2 (when 0 < rotary_dim < HS) are the identity.  In-place allowed
 * (Qout == Qin).  decode == forward with T = 1 and pos_offset = position.
 * ------------------------------------------------------------------------------------------- */
MILA_API int mila_cdna4_rope_build_cache(float* cos_cache, float* sin_cache, int max_seq, int HS,
                                         float base, int rotary_dim, mila_stream_t stream);
MILA_API int mila_cdna4_rope_forward_bf16(uint16_t* Qout, uint16_t* Kout, const uint16_t* Qin,
                                          const uint16_t* Kin, const float* cos_cache,
                                          const float* sin_cache, int B, int T, int NH, int NKV, int HS,
                                          int pos_offset, int max_seq, mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Glue.  Token ids are read from DEVICE memory (Embeddings/Kernels/TokenEmbedding.Bf16.cu:23-97);
 * an id outside [0,vocab) raises a sticky device flag reported by mila_cdna4_check_index_error.
 * embedding scale: y = bf16(float(x) * scale) when scale != 0 (Components/Embeddings/
 * TokenEmbedding.ixx:179-181).  lpe: Encodings/Lpe/Kernels/Lpe.Fp32.cu:33-124 (bf16 row added).
 * split3: ../Tensors/Operations/Kernels/Structural.cu:106.  scale: Math.Elementwise.cu:106-113.
 * ------------------------------------------------------------------------------------------- */
MILA_API int mila_cdna4_embedding_gather_bf16(uint16_t* Y, const int32_t* tokens, const uint16_t* table,
                                              int n_tok, int C, int vocab, float scale,
                                              int32_t* error_flag, mila_stream_t stream);
/* FP8 tied embedding/lm_head table with one fp32 scale per vocab row
 * (Embeddings/Kernels/TokenEmbedding.Fp8.cu:33-69): y = bf16(float(e4m3) * row_scale[tok]) [* scale] */
MILA_API int mila_cdna4_embedding_gather_bf16_qfp8(uint16_t* Y, const int32_t* tokens, const uint8_t* table,
                                                   const float* row_scales, int n_tok, int C, int vocab,
                                                   float scale, int32_t* error_flag, mila_stream_t stream);
MILA_API int mila_cdna4_lpe_bf16(uint16_t* Y, const int32_t* tokens, const uint16_t* wte,
                                 const uint16_t* wpe, int B, int T, int C, int out_stride_T, int vocab,
                                 int32_t* error_flag, mila_stream_t stream);
MILA_API int mila_cdna4_split3_bf16(uint16_t* a, uint16_t* b, uint16_t* c, const uint16_t* X, int rows,
                                    int na, int nb, int nc, mila_stream_t stream);
MILA_API int mila_cdna4_scale_bf16(uint16_t* Y, const uint16_t* X, int64_t n, float s,
                                   mila_stream_t stream);
/* Synthetic parameters, generated where they live: dst[i] = bf16(offset + amp * (2u - 1)), u in [0, 1) from the counter-based
 * splitmix64(seed, i) -- the same bits for the same (seed, i) on any grid, so the CPU oracle regenerates them without a file
 * (SURVEY.md section 8d).  Stands where the reference's initializeParameters kernels do (Linear.ixx:1029-1054); no checkpoint exists offline. */
MILA_API int mila_cdna4_fill_uniform_bf16(uint16_t* dst, int64_t n, uint64_t seed, float amp, float offset,
                                          mila_stream_t stream);
MILA_API int mila_cdna4_convert_f32_to_bf16(uint16_t* Y, const float* X, int64_t n, mila_stream_t stream);
MILA_API int mila_cdna4_convert_bf16_to_f32(float* Y, const uint16_t* X, int64_t n, mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * FP32 rows (ABI 4).  The reference keeps FP32 paths of these ops "for validation and reference" (OperationTraits.Cuda.ixx:50-54, :108-126, :274-282): with them -- and
 * layernorm_fp32 / softmax_fp32 / gelu_fp32 / residual_fp32 / rmsnorm_fp32 above -- BASELINE config 1's model (GPT-2 124M, FP32) runs on the device and is compared with
 * the reference's CPU backend at FP32 tolerance.  fp32 in, fp32 accumulate (ascending K), fp32 out; validation kernels, not performance kernels.
 *   matvec_fp32 / gemm_fp32: y = x W^T (+ bias) [act 1: + tanh-GELU of the result], W [N, K] row-major -- replaces Linear/Kernels/MatVec/CudaMatVecBias.Fp32.cu:40,
 *     Linear/Kernels/MatMul/CudaMatMulFp32.cu:31-183;
 *   mha_fp32 / mha_kv_write_fp32 / mha_decode_fp32: the FP32 twins of mha_bf16 / mha_kv_write_bf16 / mha_decode_bf16 (Attention/MHA/CudaMhaOp.ixx:145-380: FP32 is the
 *     reference's only CUDA MHA row); caches [B, NH, capacity, HS] fp32; any head size <= 512;
 *   lpe_fp32: Encodings/Lpe/Kernels/Lpe.Fp32.cu:33-124;  rope_forward_fp32: Encodings/Rope/Kernels/Rope.Fp32.cu:288-321 (tables from rope_build_cache).
 * ------------------------------------------------------------------------------------------- */
MILA_API int mila_cdna4_matvec_fp32(float* y, const float* x, const float* W, const float* bias, int K, int N, mila_stream_t stream);
MILA_API int mila_cdna4_gemm_fp32(float* Y, const float* X, const float* W, const float* bias, int M, int K, int N, int act, mila_stream_t stream);
MILA_API int mila_cdna4_mha_fp32(float* Y, const float* QKV, int B, int T, int C, int NH, mila_stream_t stream);
MILA_API int mila_cdna4_mha_kv_write_fp32(float* Kc, float* Vc, const float* QKV, int B, int T, int C, int NH, int start_pos, int capacity, mila_stream_t stream);
MILA_API int mila_cdna4_mha_decode_fp32(float* Y, const float* QKV, float* Kc, float* Vc, int B, int C, int NH, int capacity, int position, mila_stream_t stream);
MILA_API int mila_cdna4_lpe_fp32(float* Y, const int32_t* tokens, const float* wte, const float* wpe, int B, int T, int C, int out_stride_T, int vocab,
                                 int32_t* error_flag, mila_stream_t stream);
MILA_API int mila_cdna4_rope_forward_fp32(float* Qout, float* Kout, const float* Qin, const float* Kin, const float* cos_cache, const float* sin_cache, int B, int T,
                                          int NH, int NKV, int HS, int pos_offset, int max_seq, mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Greedy device sampler: token_out[0] = argmax(logits), ties to the lowest index.
 * replaces Sampling/Kernels/Sampling.cuh: cuda_sample_argmax_fp32 / _bf16 (Sampling.cu:23-75).  The token stays on
 * the device and feeds the next step's embedding gather (the reference's "decode-ahead", Models/GemmaModel.ixx:497-537).
 * ------------------------------------------------------------------------------------------- */
MILA_API size_t mila_cdna4_sample_scratch_bytes(void);
MILA_API int mila_cdna4_sample_argmax_fp32(const float* logits, int32_t* token_out, int vocab, void* scratch,
                                           size_t scratch_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_sample_argmax_bf16(const uint16_t* logits, int32_t* token_out, int vocab, void* scratch,
                                           size_t scratch_bytes, mila_stream_t stream);

/* Stochastic sampler: x = (softcap > 0 ? softcap * tanh(l / softcap) : l) / temperature;
 * top-k (0 = off) keeps the values strictly above the (k+1)-th largest; top-p (>= 1 = off) keeps the smallest set of
 * highest-probability survivors whose mass exceeds top_p * total; token = first index, in token order, with
 * probability > 0 and cumulative >= r * total (vocab - 1 otherwise); r in [0, 1] is drawn by the caller.
 * replaces Sampling/Kernels/Sampling.cuh: cuda_sample_stochastic_fp32 / _bf16 (Sampling.cu:655-714; semantics of the
 * single-block kernel :760-905).  temperature <= 0 is the greedy sampler above.  Deterministic for fixed arguments. */
MILA_API size_t mila_cdna4_sample_stochastic_scratch_bytes(int vocab);
MILA_API int mila_cdna4_sample_stochastic_fp32(const float* logits, int32_t* token_out, int vocab, float softcap,
                                               float temperature, int top_k, float top_p, float r, void* scratch,
                                               size_t scratch_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_sample_stochastic_bf16(const uint16_t* logits, int32_t* token_out, int vocab, float softcap,
                                               float temperature, int top_k, float top_p, float r, void* scratch,
                                               size_t scratch_bytes, mila_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused decode-step kernels (GemmaBlock::decode,
 * Components/Transformers/Gemma/Gemma.Block.ixx:287-356, as a fused schedule).  Each computes
 * exactly what the listed chain of unfused ops computes, including every intermediate bf16 rounding,
 * so results are bit-identical to calling the unfused entry points in sequence.
 * ------------------------------------------------------------------------------------------- */
/* y = Linear( rmsnorm(x; nw, eps) ), weight format `fmt` (0 bf16, 1 fp8, 2 fp4).  With
 * `geglu` != 0, W has 2*N rows [gate | up] and y[n] = bf16( gelu_tanh(bf16(gate_n)) * bf16(up_n) ).
 * With res != NULL the prologue is the sandwich tail of the previous sub-block:
 *   r = bf16( res + bf16(rmsnorm(x; pw)) ) [* post_scale]; x' = rmsnorm(r; nw); r is written to
 *   res_out by block 0. */
typedef struct mila_fused_matvec_args {
    uint16_t* y;              /* [N] bf16 output (or [N] after GeGLU)                         */
    const uint16_t* x;        /* [K] bf16 input                                               */
    const void* W;            /* weights in `fmt` layout                                      */
    const float* scales;      /* fp8: [rows]; fp4: [rows, K/group]; bf16: NULL                */
    const uint16_t* norm_w;   /* [K] rmsnorm weight applied to the matvec input, or NULL      */
    const uint16_t* post_w;   /* [K] rmsnorm weight of the sandwich tail, or NULL             */
    const uint16_t* res;      /* [K] residual input of the sandwich tail, or NULL             */
    uint16_t* res_out;        /* [K] where r is stored (required when res != NULL)            */
    float post_scale;         /* layer scalar applied to r (1.0f = none)                      */
    float eps;
    int fmt, K, N, group, geglu;
    int f32_out;              /* != 0: y is float[N] (lm_head logits); not combinable with geglu */
    /* ABI 3, optional (NULL = off), f32_out only: the greedy sampler's first stage in this launch's epilogue -- every workgroup leaves the largest of its logits and its
     * index (ties to the lowest index, Sampling.cu:23-75) in argmax_scratch (>= sample_scratch_bytes()), *argmax_blocks (host) receives their number;
     * sample_argmax_final_advance then picks the token: the same token as sample_argmax_fp32 on y, one launch fewer. */
    void* argmax_scratch;
    size_t argmax_scratch_bytes;
    int* argmax_blocks;
} mila_fused_matvec_args;
MILA_API int mila_cdna4_fused_norm_matvec(const mila_fused_matvec_args* host_args, mila_stream_t stream);

/* q/k(/v) per-head RMSNorm + RoPE + KV-cache append for one decode token, in one launch:
 *   q <- rope(rmsnorm(q; qw)), k' = rope(rmsnorm(k; kw)), v' = rmsnorm(v_src; vw or ones),
 *   cache[pos % capacity] <- (k', v').  v_src == k (raw) on Gemma global layers. */
MILA_API int mila_cdna4_fused_qkv_post(uint16_t* q_out, uint16_t* Kc, uint16_t* Vc, const uint16_t* q,
                                       const uint16_t* k, const uint16_t* v_src, const uint16_t* qw,
                                       const uint16_t* kw, const uint16_t* vw, const float* cos_cache,
                                       const float* sin_cache, int NH, int NKV, int HS, int position,
                                       int capacity, float eps, mila_stream_t stream);


/* Prefill forms of the two glue fusions above, T rows per launch (B == 1), bit-identical to the unfused launches:
 *   fused_qkv_post_prefill: for token t < T at position pos_offset + t, q/k/v_src rows at + t * src_row_stride elements
 *     (the packed qkv_proj output: no split3 copy), q_out[t] <- rope(rmsnorm(q)), cache row <- (rope(rmsnorm(k)), rmsnorm(v_src))
 *     -- replaces split3 + q_norm + k_norm + v_norm + rope.prefill + kv_write (Gemma.Block.ixx:215-262);
 *   fused_tail_norm_bf16: R = bf16(bf16(RES + bf16(rmsnorm(A; post_w))) * post_scale), XN = rmsnorm(R; next_w) (XN / next_w
 *     may both be NULL) -- replaces RmsNorm + Residual (+ scale) + RmsNorm (Gemma.Block.ixx:339-356, :287-289); 1024 < dim <= 8192. */
MILA_API int mila_cdna4_fused_qkv_post_prefill(uint16_t* q_out, uint16_t* Kc, uint16_t* Vc, const uint16_t* q,
                                               const uint16_t* k, const uint16_t* v_src, int64_t src_row_stride,
                                               const uint16_t* qw, const uint16_t* kw, const uint16_t* vw,
                                               const float* cos_cache, const float* sin_cache, int T, int NH, int NKV,
                                               int HS, int pos_offset, int capacity, float eps, mila_stream_t stream);
MILA_API int mila_cdna4_fused_tail_norm_bf16(uint16_t* R, uint16_t* XN, const uint16_t* A, const uint16_t* RES,
                                             const uint16_t* post_w, const uint16_t* next_w, int rows, int dim,
                                             float post_scale, float eps, mila_stream_t stream);
/* fused_tail_norm_bf16 whose XN row is ALSO written as the W4A8 Linear's activation operand: XQ[row] = e4m3(XN[row] / XS[row]),
 * XS[row] = max(absmax(XN[row]), 1e-12) / 448 -- bit for bit what quantize_fp8_per_token(XQ, XS, XN) would produce (the reference's
 * cuda_fp8_quantize_per_token, Fp8Prefill/CudaFp8Prefill.cu:108-160, folded into the producer of its input). */
MILA_API int mila_cdna4_fused_tail_norm_quant_bf16(uint16_t* R, uint16_t* XN, uint8_t* XQ, float* XS, const uint16_t* A,
                                                   const uint16_t* RES, const uint16_t* post_w, const uint16_t* next_w, int rows,
                                                   int dim, float post_scale, float eps, mila_stream_t stream);

/* Graph-replay forms: the decode position lives in DEVICE memory so that one captured hipGraph
 * serves every step (the reference re-launches ~1100 kernels per token from the host,
 * SPEC/Gemma4InferenceReview.md:71-84).  `max_len` fixes the split count at capture time; splits past
 * the live band contribute (m=-inf, l=0) partials.  advance_position: *position_dev += 1. */
MILA_API int mila_cdna4_attn_decode_bf16_devpos(uint16_t* Y, const uint16_t* Q, const uint16_t* Kc,
                                                const uint16_t* Vc, void* scratch, size_t scratch_bytes,
                                                int B, int NH, int NKV, int HS, int capacity,
                                                const int32_t* position_dev, int max_len, int window,
                                                float scale, mila_stream_t stream);
MILA_API int mila_cdna4_fused_qkv_post_devpos(uint16_t* q_out, uint16_t* Kc, uint16_t* Vc, const uint16_t* q,
                                              const uint16_t* k, const uint16_t* v_src, const uint16_t* qw,
                                              const uint16_t* kw, const uint16_t* vw, const float* cos_cache,
                                              const float* sin_cache, int NH, int NKV, int HS,
                                              const int32_t* position_dev, int capacity, float eps,
                                              mila_stream_t stream);
MILA_API int mila_cdna4_advance_position(int32_t* position_dev, mila_stream_t stream);
/* The host side of the reference's decode-ahead loop (Models/GemmaModel.ixx:496-568: enqueueSampleNext / awaitSampledToken) learns each
 * sampled token while the NEXT step already runs.  `ring` is ring_size 64-bit words of host-visible memory (pinned + mapped), `seq_dev` a device
 * counter: both kernels do  seq = ++*seq_dev;  ring[seq % ring_size] = seq << 32 | (uint32)*token  with ONE system-scope release store, so the host
 * polls the slot until it carries the sequence number it expects -- no event, no copy, no stream wait on the decode path.
 * advance_position_snapshot additionally does *position_dev += 1 (the last node of a captured step); snapshot_token is the eager sampler's counterpart. */
MILA_API int mila_cdna4_advance_position_snapshot(int32_t* position_dev, const int32_t* token, unsigned long long* seq_dev, unsigned long long* ring, int ring_size,
                                                  mila_stream_t stream);
MILA_API int mila_cdna4_snapshot_token(const int32_t* token, unsigned long long* seq_dev, unsigned long long* ring, int ring_size, mila_stream_t stream);
/* the tail of a captured greedy step in two launches instead of three: sample_argmax_fp32 whose final reduction also does *position_dev += 1 and, when ring != NULL,
 * the publication of advance_position_snapshot (seq_dev / ring / ring_size as there; NULL / NULL / 0 = no publication).  Same token as sample_argmax_fp32. */
MILA_API int mila_cdna4_sample_argmax_advance_fp32(const float* logits, int32_t* token_out, int vocab, void* scratch, size_t scratch_bytes,
                                                   int32_t* position_dev, unsigned long long* seq_dev, unsigned long long* ring, int ring_size,
                                                   mila_stream_t stream);

/* the final stage of the greedy sampler alone, over the `blocks` partials a fused_norm_matvec launch with argmax_scratch left in `scratch`, with the tail of
 * sample_argmax_advance_fp32 (*position_dev += 1, publication when ring != NULL) */
MILA_API int mila_cdna4_sample_argmax_final_advance(int32_t* token_out, const void* scratch, size_t scratch_bytes, int blocks, int32_t* position_dev,
                                                    unsigned long long* seq_dev, unsigned long long* ring, int ring_size, mila_stream_t stream);

/* One-launch decode attention for one token (B == 1): q/k/v per-head RMSNorm + RoPE + KV append (the
 * work of fused_qkv_post) folded into the flash-decode kernel's prologue, where it overlaps the first
 * K/V round trip.  q_raw [NH*HS], k_raw / v_raw [NKV*HS] are the raw projections (v_raw == k_raw on Gemma
 * global layers).  position_dev != NULL selects the graph-replay form; `position` is then an upper bound on the live length (position + 1) the captured
 * launch will be replayed at, 0 = the capacity: an unwindowed layer's launch geometry is chosen per BUCKET of the live length (4096, 8192, 16384, ... keys, clipped to
 * the capacity; the eager form uses position + 1), so a caller whose positions leave the bucket re-captures (GemmaTransformer::ensureGraph).  Bit-identical to
 * fused_qkv_post + attn_decode_bf16. */
MILA_API int mila_cdna4_fused_attn_decode_bf16(uint16_t* Y, uint16_t* Kc, uint16_t* Vc, const uint16_t* q_raw,
                                               const uint16_t* k_raw, const uint16_t* v_raw, const uint16_t* qw,
                                               const uint16_t* kw, const uint16_t* vw, const float* cos_cache,
                                               const float* sin_cache, void* scratch, size_t scratch_bytes, int NH,
                                               int NKV, int HS, int capacity, int position,
                                               const int32_t* position_dev, int window, float scale, float eps,
                                               mila_stream_t stream);

/* fused_attn_decode_bf16 for B rows decoded at ONE position (the reference's decode kernels take the batch in their grid, Gqa.Decode.Bf16.cu:379-387): batch row b reads
 * q_raw / k_raw / v_raw + b * raw_b_stride elements and its own caches [b]; Y [B, NH*HS]; scratch from attn_decode_scratch_bytes(B, NH, HS).  Bit-identical, row by row,
 * to fused_qkv_post on that row's caches + attn_decode_bf16 with the same B. */
MILA_API int mila_cdna4_fused_attn_decode_batch_bf16(uint16_t* Y, uint16_t* Kc, uint16_t* Vc, const uint16_t* q_raw, const uint16_t* k_raw,
                                                     const uint16_t* v_raw, int64_t raw_b_stride, const uint16_t* qw, const uint16_t* kw,
                                                     const uint16_t* vw, const float* cos_cache, const float* sin_cache, void* scratch,
                                                     size_t scratch_bytes, int B, int NH, int NKV, int HS, int capacity, int position,
                                                     const int32_t* position_dev, int window, float scale, float eps, mila_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MILA_CDNA4_H */
