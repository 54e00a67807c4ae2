// Test / tuning hooks exported by the shared object but NOT part of the drop-in ABI
// (include/mila_cdna4.h).  Used by tests/ and the micro-benchmarks only; nothing under mila_amd/host includes this file.
// The mila_cdna4_tune_* hooks change process-wide launch heuristics, so they are INERT (MILA_E_UNSUPPORTED, nothing stored)
// unless the process set MILA_CDNA4_TUNING=1 before the library was loaded: a product process cannot reach that state.
#pragma once
#include "../../include/mila_cdna4.h"

#ifdef __cplusplus
extern "C" {
#endif
/* NAMED tuning variables (round 4): every launch heuristic a test or a tool may override is an int with a dotted name, registered next to the rule it steers
 * (MILA_TUNE in csrc/common.h).  tune(name, value) sets one, tune_get reads it, tune_reset restores every default, tune_list prints "name=value (default d)" lines.
 *   gemm.*        bf16 / staged GEMM dispatch (csrc/gemm.hip, gemm256.hip)      gemm_fp8.*   fp8 x fp8 (W4A8 / W8A8) dispatch (gemm256.hip, gemm_fp8_tail.hip)
 *   attn.*        decode attention (csrc/attention.hip)                          flash.*      flash prefill (attention_prefill.hip)
 * last_form: the kernel forms the calling thread's Linear / attention entry points ran since the previous call ("gemm256+skinny_bf16", "fp8_ldsdma_256x128+fp8_skinny", ...),
 * so a test can assert WHICH form a row count was routed to. */
MILA_API int mila_cdna4_tune(const char* name, int value);
MILA_API int mila_cdna4_tune_get(const char* name, int* value);
MILA_API int mila_cdna4_tune_reset(void);
MILA_API size_t mila_cdna4_tune_list(char* buf, size_t cap);
MILA_API size_t mila_cdna4_last_form(char* buf, size_t cap);
/* The names (tune_list prints them with their values and defaults; each is documented where it is registered):
 *   matvec.rows_per_wave, matvec.chunks_in_flight, matvec.max_workgroups                 0 = the default rule                                         (csrc/matvec.hip)
 *   gemm.force128          1 = always the 128 x 128 register-staged GEMM (A/B against the LDS-DMA kernels)                                           (csrc/gemm.hip)
 *   gemm.bf16_skinny       the bf16 skinny weight-streaming kernel for <= 64-row prompts and remainders (default 1)
 *   gemm.fewrow            the few-row (<= 32 rows) form of gemm_bf16_ws (default 1)
 *   gemm.skinny_ahead_rows up to this many rows the skinny kernels go ahead of an applicable tile grid (default 1)
 *   gemm.splitk, gemm.splitk_min_rows      the split-K forms of gemm_bf16_ws / gemm_fp8_scaled_ws; row counts below the minimum stay off them (1, 2)
 *   gemm.ldsdma_loose_tiles  the 256 x 128 ring from this many tiles on whatever its last round's fill (30; 0 = the fill rule only)                   (csrc/gemm256.hip)
 *   gemm.rowwise_epilogue  0 = an output whose row pitch is no multiple of 128 bytes keeps the direct epilogue stores (default 1: row-wise through LDS)
 *   gemm.schedule          0 = all eight waves in lockstep; 1 = staggered (ping-pong), four phases per K-tile; 2 = 1, preferring the 256 x 128 ring; 3 = staggered, two
 *                          phases per K-tile; 4 = 3 + the fp8 shapes on the 256 x 256 kernel wherever it applies; 5 (default) = 4 with a static priority for waves 4-7.  Same bits.
 *   gemm.persistent        0 = one workgroup per tile instead of the persistent tile walk (default 1).  Same bits.
 *   gemm.tile256_min_fill  the 256 x 256 grid applies when its tiles fill at least this many percent of their rounds of CUs (default 80; profiles/r04_tile256_fill_rule.txt:
 *                          at 75 % -- GPT-2's fc_1, 384 tiles -- the 256 x 128 ring is 56 us against 74).  Same bits.
 *   gemm.walk_min_tiles    the 256 x 128 ring walks its tiles from this many on (default 3 x 256 + 1: up to three rounds one workgroup per tile is as fast or faster).  Same bits.
 *   gemm.colsplit          the column split of a tile list whose last round is nearly empty (default 1)
 *   gemm_fp8.tail_form     0 (default) = LDS-DMA kernels on the leading multiple of 256 rows, the tail kernels of gemm_fp8_tail.hip on the rest; 1 = EVERY row on the masked
 *                          128-row tiles (bit-identical to the LDS-DMA kernels: the test of that statement); 2 = every row as skinny pieces (fp32-rounding-level differences)
 *   gemm_fp8.skinny_whole_x, gemm_fp8.big_rule, gemm_fp8.splitk_min_rows      the skinny kernel's barrier-free <= 4-row form (1); which row counts below 512 take the LDS-DMA
 *                          kernels (rules 0 .. 3 of gemm256.hip: fp8_big_rows; 3); row counts below this stay off the fp8 split-K form (17)
 *   attn.positions_per_split, attn.max_workgroups, attn.heads_per_group_512, attn.xcd_local, attn.mfma_decode, attn.mfma_min_band      decode attention (csrc/attention.hip)
 *   flash.form             8 (default) = the LDS-DMA forms; 9 = lockstep 8-wave workgroups at HS 256 too; 10 = the ping-pong 8-wave form; 11 = the software-pipelined loop; 2 = HS 512 as 4-wave d-split
 *                          workgroups; 1 = the register-staged kernels.  Same bits.                                                       (csrc/attention_prefill.hip) */
/* engine diagnostics: the next decode_engine launches write wall-clock stamps (100 MHz) of the first 8 workgroups' 8 waves, 16 slots each */
MILA_API int mila_cdna4_decode_engine_debug(unsigned long long* buf);
/* decode all 256 byte values with the hardware converts used by the kernels:
 * out_fp8[256] floats; out_fp4[512] floats (byte b -> [2b] low nibble, [2b+1] high nibble),
 * each for the 4 byte positions of a dword: out_fp4 has 4*512 floats, out_fp8 4*256. */
MILA_API int mila_cdna4_selftest_decode(float* out_fp8, float* out_fp4, mila_stream_t stream);
/* out[0..63] = wave_sum(in[lane]), out[64..127] = wave_max, out[128..191] = the ds_bpermute butterfly */
MILA_API int mila_cdna4_selftest_wave_reduce(float* out, const float* in, mila_stream_t stream);
MILA_API int mila_cdna4_selftest_mfma_fp8(float* C, const uint8_t* A, const uint8_t* B, int mode, mila_stream_t stream);
/* streaming-copy ceiling: dst <- src with 16-byte accesses; used to report a measured HBM roof */
MILA_API int mila_cdna4_stream_copy(void* dst, const void* src, size_t bytes, mila_stream_t stream);
MILA_API int mila_cdna4_stream_read(float* sink, const void* src, size_t bytes, mila_stream_t stream);

/* =============================================================================================================================
 * EXPERIMENTS: in-launch alternatives to the six-launch decode layer, each bit-identical to it and each MEASURED SLOWER on MI355X
 * (DESIGN.md section 5).  Not part of the drop-in ABI and not called by anything under mila_amd/host.  The attention / matvec / prefetch
 * variants are template flags of product kernels and live in libmila_cdna4.so; the decode chain and the persistent engine are whole files
 * and live in libmila_cdna4_experiments.so (csrc/experiments/), which only tests/ and tools/ load.
 * ============================================================================================================================= */
/* Flash-decode with the split combine moved into the consumer: fused_attn_decode_partials_bf16 is fused_attn_decode_bf16
 * without its second launch -- it leaves the per-split partials (float [NH, splits, HS + 4]: O | m | l | pad) in `scratch`;
 * matvec_attn_combine is the o_proj Linear (Gemma.Block.ixx: o_proj after the attention op) whose x is combined from those
 * partials in its prologue, element for element the arithmetic of the combine launch, so
 *   fused_attn_decode_partials + matvec_attn_combine  ==  fused_attn_decode + matvec_bf16[_qfp8|_qfp4]   bit for bit,
 * one launch fewer.  Every workgroup re-reads all partials from L2, so callers use it while NH * splits * (HS + 4) * 4 bytes
 * stays small (Gemma sliding-window layers: 266 KB); attn_decode_split_count() gives `splits` for a (window, capacity). */
MILA_API int mila_cdna4_attn_decode_split_count(int B, int NH, int NKV, int HS, int capacity, int window);
MILA_API int mila_cdna4_fused_attn_decode_partials_bf16(uint16_t* Kc, uint16_t* Vc, const uint16_t* q_raw,
                                                        const uint16_t* k_raw, const uint16_t* v_raw,
                                                        const uint16_t* qw, const uint16_t* kw, const uint16_t* vw,
                                                        const float* cos_cache, const float* sin_cache, void* scratch,
                                                        size_t scratch_bytes, int NH, int NKV, int HS, int capacity,
                                                        int position, const int32_t* position_dev, int window,
                                                        float scale, float eps, mila_stream_t stream);
MILA_API int mila_cdna4_matvec_attn_combine(uint16_t* y, const void* partials, int splits, int NH, int HS, const void* W,
                                            const float* scales, int fmt, int N, int group, mila_stream_t stream);

/* Decode chain: the four Linears between two attention calls of a Gemma decode step in ONE launch
 * (Gemma.Block.ixx:287-356 from o_proj to the end of the block, plus the next block's input norm + qkv_proj,
 * Gemma.Block.ixx:287-300, or the final norm + tied lm_head, Gemma.ixx forward tail):
 *   phase 0  a  = o_proj(attn)
 *   phase 1  r1 = bf16(res + bf16(rmsnorm(a; post_attn_w)));  h = GeGLU(fc_gate_up(rmsnorm(r1; pre_ffn_w)))
 *   phase 2  d  = fc_down(h)
 *   phase 3  r2 = bf16(bf16(r1 + bf16(rmsnorm(d; post_ffn_w))) * layer_scalar);  y = next(rmsnorm(r2; next_norm_w))
 * `next` is a qkv_proj in the layer's weight format (y: bf16[N_next]) or, with f32_out, the lm_head table in
 * `next_fmt` (y: float[N_next]).  r2 is written to res_out.  Results are bit-identical to the sequence
 * matvec + fused_norm_matvec(geglu) + matvec + fused_norm_matvec.  One workgroup per CU; the phases hand their
 * vectors over through `scratch` (decode_chain_scratch_bytes(D, F), zero its header ONCE with decode_chain_init;
 * consecutive launches that share a scratch must be stream-ordered).  A launch whose workgroups cannot all be
 * resident gives up after a bounded wait and sets the error word read by decode_chain_status (0 = healthy). */
typedef struct mila_decode_chain_args {
    const uint16_t* attn;         /* [K_attn] attention output                                        */
    const uint16_t* res;          /* [D] residual stream entering the post-attention tail             */
    uint16_t* res_out;            /* [D] r2                                                           */
    void* y;                      /* [N_next] bf16, or float when f32_out                             */
    const void* W_o;       const float* s_o;         /* [D, K_attn]                                   */
    const void* W_gate_up; const float* s_gate_up;   /* [2F, D] rows [gate | up]                      */
    const void* W_down;    const float* s_down;      /* [D, F]                                        */
    const void* W_next;    const float* s_next;      /* [N_next, D]                                   */
    const uint16_t* post_attn_w;  /* [D] */
    const uint16_t* pre_ffn_w;    /* [D] */
    const uint16_t* post_ffn_w;   /* [D] */
    const uint16_t* next_norm_w;  /* [D] next layer's input norm, or the final norm                   */
    float layer_scalar, eps;
    int fmt, group;               /* format of o_proj / fc_gate_up / fc_down (0 bf16, 1 fp8, 2 fp4)   */
    int next_fmt, next_group;     /* format of W_next                                                 */
    int f32_out;
    int D, F, K_attn, N_next;
    void* scratch; size_t scratch_bytes;
} mila_decode_chain_args;
MILA_API size_t mila_cdna4_decode_chain_scratch_bytes(int D, int F);
MILA_API int mila_cdna4_decode_chain_init(void* scratch, size_t scratch_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_decode_chain_status(const void* scratch, int32_t* error_out, mila_stream_t stream);
MILA_API int mila_cdna4_decode_chain(const mila_decode_chain_args* host_args, mila_stream_t stream);

/* The same four phases, same arguments, same bits, as a persistent ENGINE (csrc/experiments/engine.hip): one 512-thread workgroup per CU, EIGHT
 * self-loading waves -- each streams its own share of the four weight matrices HBM -> LDS by LDS-DMA into a private 12-KiB ring, keeps requests
 * of phase p + 1 in flight across the hand-off of phase p, and accumulates out of its ring in matvec's canonical order; a phase's outputs reach
 * every CU as 4-byte data-tagged granules (no flag, no fence).  MEASURED SLOWER than the four launches it replaces, 1.4-2x per layer
 * (profiles/r02_engine_measurements.txt: launches 75.6 / 47.0 / 34.5 us, engine 107 / 82 / 69 us for bf16 / fp8 / fp4); opt-in for the
 * micro-benchmarks only, never called by GemmaTransformer.  decode_engine_applicable() says whether the geometry fits (256 CUs, the LDS budget,
 * fp4 scale alignment); scratch as for the chain (decode_engine_scratch_bytes, zeroed ONCE by decode_engine_init); status as for the chain. */
MILA_API size_t mila_cdna4_decode_engine_scratch_bytes(int D, int F);
MILA_API int mila_cdna4_decode_engine_init(void* scratch, size_t scratch_bytes, mila_stream_t stream);
MILA_API int mila_cdna4_decode_engine_status(const void* scratch, int32_t* error_out, mila_stream_t stream);
MILA_API int mila_cdna4_decode_engine_applicable(int fmt, int group, int D, int F, int K_attn, int N_next, int next_fmt);
MILA_API int mila_cdna4_decode_engine(const mila_decode_chain_args* host_args, mila_stream_t stream);


/* The same in ONE launch on layers whose live band is split over workgroups: the combine launch of the reference's split-K decode
 * (the fixup kernel, Gqa.Decode.Bf16.cu:297-351) becomes the tail of the workgroup whose partials arrive last.
 * tickets: uint32 [mila_cdna4_attn_decode_ticket_count(1, NH)], zeroed once by the caller (memset_zero) and owned by ONE
 * stream at a time; every call leaves it zero again.  Bit-identical to fused_attn_decode_bf16. */
MILA_API size_t mila_cdna4_attn_decode_ticket_count(int B, int NH);
MILA_API int mila_cdna4_fused_attn_decode_onepass_bf16(uint16_t* Y, uint16_t* Kc, uint16_t* Vc, const uint16_t* q_raw,
                                                       const uint16_t* k_raw, const uint16_t* v_raw, const uint16_t* qw,
                                                       const uint16_t* kw, const uint16_t* vw, const float* cos_cache,
                                                       const float* sin_cache, void* scratch, size_t scratch_bytes,
                                                       uint32_t* tickets, size_t ticket_count, int NH, int NKV, int HS,
                                                       int capacity, int position, const int32_t* position_dev, int window,
                                                       float scale, float eps, mila_stream_t stream);

/* Warm the 256 MiB Infinity Cache with a byte range a later kernel will stream (the next Linear's weights), from a side stream
 * while the current kernel runs; reads one dword per 128-byte line in address order, writes nothing (sink: any 4 writable bytes,
 * never written in practice).  No reference counterpart (a 12 GB card has no memory-side cache to warm); results are unaffected. */
MILA_API int mila_cdna4_prefetch_l3(const void* src, size_t bytes, int workgroups, float* sink, mila_stream_t stream);

/* fused_attn_decode_bf16 from an argument block (tickets != NULL: the one-pass form), with optional WARM RANGES: warm_a_blocks
 * extra workgroups per grid row of the attention launch and warm_b_blocks extra block planes of the combine launch touch one dword
 * per 128-byte line of warm_a / warm_b (weights a later Linear of the step will stream), so the 256 MiB Infinity Cache fills
 * while these latency-bound launches leave HBM idle.  The warm blocks only read; the attention result is bit-identical. */
typedef struct mila_fused_attn_args {
    uint16_t* Y; uint16_t* Kc; uint16_t* Vc;
    const uint16_t* q_raw; const uint16_t* k_raw; const uint16_t* v_raw;
    const uint16_t* qw; const uint16_t* kw; const uint16_t* vw;
    const float* cos_cache; const float* sin_cache;
    void* scratch; size_t scratch_bytes;
    uint32_t* tickets; size_t ticket_count;         /* NULL / 0: attention + combine launches */
    const void* warm_a; size_t warm_a_bytes; int warm_a_blocks;
    const void* warm_b; size_t warm_b_bytes; int warm_b_blocks;
    size_t warm_b_pair_offset;                      /* != 0: warm_b is the head of TWO streams this many bytes apart (gate | up), warm_b_bytes in all */
    int NH, NKV, HS, capacity, position;
    const int32_t* position_dev;
    int window;
    float scale, eps;
} mila_fused_attn_args;
MILA_API int mila_cdna4_fused_attn_decode_ex(const mila_fused_attn_args* host_args, mila_stream_t stream);

/* the 256 x 256 bf16 GEMM tile on FOUR waves (one per SIMD, 128 x 128 each; csrc/experiments/gemm4w.hip): bit-identical to gemm_bf16 / gemm_geglu_bf16 on the
 * shapes both take, measured 15-25 % slower (profiles/r03_gemm4w.txt).  libmila_cdna4_experiments.so. */
MILA_API int mila_cdna4_exp_gemm4w_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, mila_stream_t stream);
MILA_API int mila_cdna4_exp_gemm4w_geglu_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, mila_stream_t stream);
#ifdef __cplusplus
}
#endif
