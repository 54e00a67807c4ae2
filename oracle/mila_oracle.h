/*
 * mila_oracle.h -- CPU ORACLE for the Mila forward() hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithms, used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the CHECKER.  Nothing in the
 * product path (mila_amd/, include/) may include, link or call it.
 *
 * Two families (paths relative to /root/reference/Mila/Src/Dnn):
 *   orc_cpu_*   line-faithful restatements of the reference's own CPU backend ops (FP32),
 *               Compute/Devices/Cpu/Operations/Cpu*.ixx -- the parity oracle for GPT-2 configs.
 *   orc_*       FP32/FP64 restatements of the arithmetic of the reference's CUDA kernels for the
 *               ops that have NO reference CPU op (RMSNorm, RoPE, GQA, GeGLU, fp8/fp4 Linear);
 *               integer outputs (fp8 bytes, fp4 nibbles, packing, gathers) are bit-exact
 *               restatements, floating outputs are computed in double so that every association
 *               order the reference's kernels use falls inside the stated tolerance.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - GELU-tanh/SiLU/...: pinned bit-for-bit against the reference's own functor header compiled
 *     from /root/reference (oracle/_ref, built by oracle/Makefile when the reference is present).
 *   - every other op: pinned by the reference tests' closed-form generators + in-test host
 *     references restated in tests/test_oracle_kats.py (the reference ships no
 *     golden-vector files), and by fixtures under tests/golden/.
 *   - net-level wiring (GPT-2 forward here; the Gemma-4 composition tests/ref_gemma.py over these ops): the reference's own
 *     tree asserts only shapes/finiteness there, so the wiring is pinned against the implementation the reference validates
 *     its checkpoints against -- the transformers GPT2LMHeadModel / Gemma4ForCausalLM FP32 logits on synthetic weights,
 *     committed as tests/golden/gpt2_hf_logits.npz and gemma4_hf_logits.npz with their generating scripts.
 */
#ifndef MILA_ORACLE_H
#define MILA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar format helpers -------------------------------------------------------------- */
uint16_t orc_f32_to_bf16(float x);            /* RNE, NaN kept quiet                         */
float    orc_bf16_to_f32(uint16_t h);
uint8_t  orc_f32_to_e4m3(float x);            /* OCP E4M3FN, RNE, saturate-to-finite (448)   */
float    orc_e4m3_to_f32(uint8_t b);
uint8_t  orc_f32_to_e2m1(float x);            /* threshold encoder, bit3 = sign              */
float    orc_e2m1_to_f32(uint8_t nibble);
void     orc_round_bf16_inplace(float* x, int64_t n);   /* x <- bf16(x) as f32              */
void     orc_f32_to_bf16_array(uint16_t* dst, const float* src, int64_t n);
void     orc_bf16_to_f32_array(float* dst, const uint16_t* src, int64_t n);

/* ---- reference CPU backend ops (FP32) ---------------------------------------------------- */
void orc_cpu_linear_naive(float* Y, const float* X, const float* W, const float* B,
                          int64_t batch, int64_t in_features, int64_t out_features);
void orc_cpu_linear_unrolled(float* Y, const float* X, const float* W, const float* B,
                             int64_t batch, int64_t in_features, int64_t out_features);
/* dispatcher: unrolled iff batch % 8 == 0 (CpuLinearOp::build) */
void orc_cpu_linear(float* Y, const float* X, const float* W, const float* B,
                    int64_t batch, int64_t in_features, int64_t out_features);
void orc_cpu_gelu(float* Y, const float* X, int64_t n);
void orc_cpu_softmax(float* Y, const float* X, int64_t outer, int64_t dim, int64_t inner);
void orc_cpu_layernorm(float* Y, float* mean, float* rstd, const float* X, const float* w,
                       const float* b, int64_t outer, int64_t dim, int64_t inner, float eps);
void orc_cpu_residual(float* Y, const float* A, const float* B, int64_t n);
/* returns 0, or -1 when a token id is out of [0, vocab) (reference throws std::out_of_range) */
int  orc_cpu_lpe(float* Y, const int32_t* tokens, const float* wte, const float* wpe,
                 int64_t B, int64_t T, int64_t C, int64_t out_stride_T, int64_t vocab);
/* packed QKV [B,T,3C] -> [B,T,C]; scratch-free interface (allocates internally) */
void orc_cpu_mha(float* Y, const float* X, int B, int T, int C, int NH);
/* GPT-2 forward (GptTransformer::forward): params in the order documented in mila_oracle.c */
void orc_cpu_gpt2_forward(float* logits, const int32_t* tokens, const float* const* params,
                          int B, int T, int C, int L, int NH, int V, int maxT);

/* ---- activations (shared functor header semantics) --------------------------------------- */
float orc_gelu_tanh(float x);
float orc_silu(float x);
void  orc_geglu(float* Y, const float* X, int64_t tokens, int64_t half);   /* fp32 math */

/* ---- RMSNorm / RoPE ---------------------------------------------------------------------- */
void orc_rmsnorm(float* Y, float* rstd, const float* X, const float* w, const float* b,
                 int64_t outer, int64_t dim, int64_t inner, float eps, float w_offset);
void orc_rope_build_cache(float* cos_out, float* sin_out, int max_seq, int head_dim,
                          float base, int rotary_dim);
void orc_rope_rotate(float* out, const float* in, const float* cos_c, const float* sin_c,
                     int64_t B, int64_t T, int64_t n_heads, int64_t head_dim, int64_t pos_offset);

/* ---- quantize-on-load (bit-exact integer outputs) ---------------------------------------- */
void orc_quantize_fp8_per_channel(uint8_t* dst, float* scales, const uint16_t* src_bf16,
                                  int64_t N, int64_t K);
void orc_quantize_fp4_per_group(uint8_t* dst_packed, float* scales, const uint16_t* src_bf16,
                                int64_t N, int64_t K, int group);
float orc_fp8_weight_scale_from_groups(const float* group_scales, int64_t n);  /* sB */
void orc_upcast_fp4_to_fp8(uint8_t* out, const uint8_t* packed, const float* scales, float weight_fp8_scale, int64_t N, int64_t K,
                           int group);
void orc_dequant_fp8(float* W, const uint8_t* q, const float* scales, int64_t N, int64_t K);
void orc_dequant_fp4(float* W, const uint8_t* packed, const float* scales, int64_t N,
                     int64_t K, int group);

/* ---- Linear on bf16 activations, three weight formats (double accumulate) ---------------- */
/* X: [M,K] f32 holding bf16-representable values; Y: [M,N] f32 (NOT rounded to bf16).       */
void orc_linear_bf16w(float* Y, const float* X, const uint16_t* W, const uint16_t* bias,
                      int64_t M, int64_t K, int64_t N);
void orc_linear_fp8w(float* Y, const float* X, const uint8_t* W, const float* scales,
                     const uint16_t* bias, int64_t M, int64_t K, int64_t N);
void orc_linear_fp4w(float* Y, const float* X, const uint8_t* Wp, const float* scales,
                     const uint16_t* bias, int64_t M, int64_t K, int64_t N, int group);
/* W8A8 / W4A8 prefill path (per-token fp8 activations, fp8 weights): restates the reference's
 * FP8-activation prefill arithmetic; weights given already as e4m3 bytes + one scale per row
 * (w_row_scale != NULL) or one per-tensor scale (w_tensor_scale).                             */
void orc_quantize_act_fp8_per_token(uint8_t* q, float* token_scales, const float* X,
                                    int64_t M, int64_t K);
void orc_linear_fp8a_fp8w(float* Y, const uint8_t* Xq, const float* token_scales,
                          const uint8_t* Wq, const float* w_row_scale, float w_tensor_scale,
                          const uint16_t* bias, int64_t M, int64_t K, int64_t N);

/* ---- attention --------------------------------------------------------------------------- */
/* Windowed causal GQA over a LINEAR history (position p at row p).
 * q [B,Tq,NH,HS]; k,v [B,Tk,NKV,HS]; query t has absolute position pos_offset+t and sees keys
 * max(0,pos-window+1)..pos when window>0, else 0..pos.  out [B,Tq,NH*HS].  double math.      */
void orc_gqa_attention(float* out, const float* q, const float* k, const float* v,
                       int B, int Tq, int Tk, int NH, int NKV, int HS,
                       int pos_offset, int window, float scale);
/* KV append into [B,NKV,capacity,HS], row = abs_pos % capacity */
void orc_kv_write(float* Kc, float* Vc, const float* k, const float* v, int B, int chunk,
                  int NKV, int HS, int start_pos, int capacity);
/* gather the linear history [B,Tk,NKV,HS] of positions [first, first+Tk) back out of a ring */
void orc_kv_ring_to_linear(float* k_lin, const float* Kc, int B, int NKV, int HS, int capacity,
                           int first, int Tk);

/* ---- glue -------------------------------------------------------------------------------- */
int  orc_embedding_gather(float* Y, const int32_t* tokens, const float* table, int64_t n_tok,
                          int64_t C, int64_t vocab, float scale_then_round_bf16);
void orc_split3(float* a, float* b, float* c, const float* X, int64_t rows, int64_t na,
                int64_t nb, int64_t nc);
float orc_softcap(float x, float cap);
int orc_sample_stochastic(const float* logits, int vocab, float softcap, float temperature, int top_k, float top_p, float r,
                          double* margins);

#ifdef __cplusplus
}
#endif
#endif
