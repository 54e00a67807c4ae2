// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mila_cdna4.h"

namespace mila {

// ---- host-side error plumbing ----------------------------------------------------------------
// (default visibility: libmila_cdna4_experiments.so -- csrc/experiments/, never loaded by the product -- links these three from libmila_cdna4.so)
#define MILA_SHARED_HELPER __attribute__((visibility("default")))
MILA_SHARED_HELPER int set_error(int code, const char* fmt, ...);   // runtime.hip; returns `code`
/// the mila_cdna4_tune_* hooks (csrc/internal.h) mutate process-wide launch heuristics: they act only in a process that set
/// MILA_CDNA4_TUNING=1 before the library was loaded (tests, tools/); in any other process they return MILA_E_UNSUPPORTED and
/// the library keeps no state between calls
MILA_SHARED_HELPER bool tuning_hooks_enabled();
MILA_SHARED_HELPER int check_hip(hipError_t e, const char* what);   // MILA_OK or MILA_E_RUNTIME (+ message)

// Named tuning variables (csrc/internal.h: mila_cdna4_tune / tune_get / tune_reset / tune_list): a launch heuristic that tests and tools may override is an int with a
// dotted name, registered where it lives -- MILA_TUNE("attn.positions_per_split", g_positions_per_split);  The registry is inert unless tuning_hooks_enabled().
struct TuneVar { const char* name; int* var; int def; TuneVar* next; };
MILA_SHARED_HELPER void register_tune_var(TuneVar* v);
struct TuneReg { TuneVar v; TuneReg(const char* n, int* p) : v{n, p, *p, nullptr} { register_tune_var(&v); } };
#define MILA_TUNE_CAT2(a, b) a##b
#define MILA_TUNE_CAT(a, b) MILA_TUNE_CAT2(a, b)
#define MILA_TUNE(NAME, VAR) static ::mila::TuneReg MILA_TUNE_CAT(mila_tune_reg_, __LINE__)(NAME, &(VAR))
// which kernel form served the calling thread's most recent Linear / attention entry (tests assert that a dispatch threshold routes a shape where they think it does)
MILA_SHARED_HELPER void note_form(const char* form);

#define MILA_REQUIRE(cond, ...)                                        \
    do {                                                               \
        if (!(cond)) return ::mila::set_error(MILA_E_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)

#define MILA_LAUNCH_CHECK(name) return ::mila::check_hip(hipGetLastError(), name)

static inline hipStream_t as_stream(mila_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

constexpr int kWave = 64;
constexpr int kNumCU = 256;     // MI355X
constexpr int kArgmaxPartials = 2 * kNumCU;      // per-workgroup (value, index) partials of the greedy sampler's first stage (own kernel, or the lm_head matvec's epilogue)

// ---- vector types ----------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- bf16 <-> f32 ----------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }

// RNE, NaN-preserving (hipcc lowers the cast to v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f)
{
    __bf16 h = static_cast<__bf16>(f);
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    bf16x2 v;
    v[0] = static_cast<__bf16>(lo);
    v[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(uint32_t, v);
}
// round an f32 value to the nearest bf16 and return it as f32 (for fused kernels that must
// reproduce the intermediate bf16 store of the unfused chain)
__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

// ---- wave / block reductions (64 lanes) -------------------------------------------------------
// All-lanes butterfly without the LDS crossbar: four DPP steps inside a 16-lane row (quad_perm xor 1,
// quad_perm xor 2, row_half_mirror, row_mirror -- every lane of a finished sub-group holds the same
// partial, so a mirror is as good as an xor), then the gfx950 v_permlane16_swap / v_permlane32_swap
// pair for the cross-row steps.  Addition is commutative, so every lane ends with identical bits.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);    // row_half_mirror
    v += dpp_f32<0x140>(v);    // row_mirror
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}
// reductions over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} only (the last two butterfly steps above): the MFMA layouts put
// one matrix row on those four lanes
__device__ __forceinline__ float quad_rows_max(float v)
{
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}
__device__ __forceinline__ float quad_rows_sum(float v)
{
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}
// integer 64-lane sum (all lanes get the result)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}
// reference implementation through the LDS crossbar (tests compare the two)
__device__ __forceinline__ float wave_sum_shfl(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide sum over `NW` waves; `red` is an LDS array of at least NW floats.  All threads get
// the result.  Contains two barriers.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red)
{
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- 16-byte global accesses -------------------------------------------------------------------
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
// streamed-once data (weights in decode): non-temporal so it does not evict x / KV from L2
__device__ __forceinline__ u32x4 ld16_nt(const void* p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
// the same 16-byte store to an address that is only 2-byte aligned (a bf16 row whose pitch is odd: GPT-2's N = 50257).  Global memory takes unaligned vector
// accesses on this target (amdhsa runs with unaligned-access mode on; the compiler itself emits global_store_dwordx4 for an align-2 16-byte object)
struct __attribute__((packed, aligned(2))) u32x4_a2 { uint32_t x, y, z, w; };
__device__ __forceinline__ void st16_a2(void* p, u32x4 v) { *reinterpret_cast<u32x4_a2*>(p) = u32x4_a2{v[0], v[1], v[2], v[3]}; }

// ---- W4A8 epilogue (CudaFp8Prefill.cu:162-211): y = float(bf16(acc * sB)) * s_m (+ bias) -------------------------------------------------------------------------
// With a bias the second step is ONE fused multiply-add in every kernel that can serve a row (LDS-DMA tiles, masked tiles, skinny): left to the compiler's contraction
// each code shape decided for itself, and the 256 x 128 tiles differed from the masked tiles in 2 of 2.4 M outputs (found when short prompts moved between them)
__device__ __forceinline__ float w4a8_scale_bias(float acc, float ws, float ts, bool has_bias, float b)
{
    const float a = round_bf16(acc * ws);
    return has_bias ? __builtin_fmaf(a, ts, b) : a * ts;
}

// ---- W8A8 epilogue (PerChannelFp8<> weights consumed natively by the fp8 matrix cores, Quantization/Weight/Policies.ixx:39-40; opt-in beside the reference's W8A16 default):
// y = (acc * s_c[n]) * s_m (+ bias) in fp32, rounded ONCE at the store -- no intermediate bf16 tensor exists on this path, so none is simulated.  The last step is one explicit
// FMA when there is a bias, for the reason given above: every kernel form that can serve a row must give the row the same bits.
__device__ __forceinline__ float w8a8_scale_bias(float acc, float wc, float ts, bool has_bias, float b)
{
    const float a = acc * wc;
    return has_bias ? __builtin_fmaf(a, ts, b) : a * ts;
}
// the fp8 x fp8 kernels' epilogue: `pc` (wave-uniform) = the weight scale is a per-channel vector (W8A8), else the per-tensor scalar of W4A8
__device__ __forceinline__ float fp8_scale_bias(bool pc, float acc, float ws, float ts, bool has_bias, float b)
{
    return pc ? w8a8_scale_bias(acc, ws, ts, has_bias, b) : w4a8_scale_bias(acc, ws, ts, has_bias, b);
}
// the Linear's stored bf16 output without a bias, as a float (the operand of a fused GeGLU epilogue)
__device__ __forceinline__ float fp8_linear_out(bool pc, float acc, float ws, float ts) { return round_bf16(fp8_scale_bias(pc, acc, ws, ts, false, 0.0f)); }
// how the host passes the weight scale through the launchers: p = device scalar (per_channel 0) or a vector over the W rows (per_channel 1; GeGLU forms: [gate rows | up rows])
struct Fp8WScale { const float* p; int per_channel; };

// ---- GELU (tanh) -------------------------------------------------------------------------------
// Components/Activations/Activation/Kernels/ElementwiseActivation.h:41-50: 0.5 x (1 + tanh(u)), u = sqrt(2 / pi) (x + 0.044715 x^3), as the reference functor writes it
__device__ __forceinline__ float gelu_tanh_precise(float x)
{
    const float cube = 0.044715f * x * x * x;
    return 0.5f * x * (1.0f + tanhf(0.7978845608f * (x + cube)));
}
// The same function for every consumer that rounds the result to bf16 (all but the FP32 elementwise kernel): 0.5 (1 + tanh(u)) == 1 / (1 + exp(-2 u)), evaluated
// with v_exp_f32 and v_rcp_f32 -- 7 vector instructions where the library tanhf costs ~30 (a 256 x 128 tile's epilogue evaluates 64 per lane: 7.7 us per tile with
// tanhf).  Relative error <= 6e-8 (1 + |2 u|) + 2 ulp: four orders below a bf16 half-ulp, so the bf16 result differs from the tanhf form's only where the exact value
// sits within ~1e-6 of a rounding boundary.  No cancellation at either end: x -> +inf gives x / 1, x -> -large gives x / exp(-2 u) -> -0.
__device__ __forceinline__ float gelu_tanh(float x)
{
    const float cube = 0.044715f * x * x * x;
    const float u = 0.7978845608f * (x + cube);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}

// ---- FP8 E4M3FN / FP4 E2M1 decode ---------------------------------------------------------------
// Hardware converts (gfx950): one instruction per 2 elements, scale operand 1.0f => exact values.
__device__ __forceinline__ bf16x2 fp8x2_to_bf16x2(uint32_t word, bool hi_half)
{
    return hi_half ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(word, 1.0f, true)
                   : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(word, 1.0f, false);
}
__device__ __forceinline__ f32x2 fp8x2_to_f32x2(uint32_t word, bool hi_half)
{
    return hi_half ? __builtin_amdgcn_cvt_pk_f32_fp8(word, true) : __builtin_amdgcn_cvt_pk_f32_fp8(word, false);
}
// byte `B` (0..3) of `word` holds two E2M1 nibbles: low nibble = even column.
template <int B>
__device__ __forceinline__ bf16x2 fp4x2_to_bf16x2(uint32_t word)
{
    return __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(word, 1.0f, B);
}
// Software decode (bit arithmetic, no table): used by the self-test to validate the hardware
// converts and by the cold paths.  magnitude index m (0..7) -> {0,.5,1,1.5,2,3,4,6}.
__device__ __forceinline__ float fp4_decode_sw(uint32_t nib)
{
    const uint32_t m = nib & 7u;
    // m<2: m*0.5 ; else (1 + (m&1)*0.5) * 2^((m>>1)-1)
    const float mag = (m < 2u) ? 0.5f * (float)m : __uint_as_float(((126u + (m >> 1)) << 23) | ((m & 1u) << 22));
    return (nib & 8u) ? -mag : mag;
}
__device__ __forceinline__ float fp8_decode_sw(uint32_t b)
{
    const uint32_t ex = (b >> 3) & 0xfu, man = b & 7u;
    float mag;
    if (ex == 0u) mag = (float)man * 0.001953125f;                       // man * 2^-9
    else mag = __uint_as_float(((ex + 120u) << 23) | (man << 20));       // (1+man/8) * 2^(ex-7)
    if (ex == 0xfu && man == 7u) mag = __uint_as_float(0x7fc00000u);
    return (b & 0x80u) ? -mag : mag;
}

__device__ __forceinline__ float dot2_bf16(bf16x2 a, bf16x2 b, float c)
{
    return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
}
__device__ __forceinline__ bf16x2 as_bf16x2(uint32_t u) { return __builtin_bit_cast(bf16x2, u); }

}  // namespace mila
