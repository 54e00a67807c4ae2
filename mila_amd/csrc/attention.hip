// Attention over a (ring) KV cache: append and split-K flash-decode (+ combine).  The MFMA
// flash-prefill / MHA kernels live in attention_prefill.hip.
//
// Semantics restated from the reference:
//   * cache [B, NKV, capacity, HS] bf16, row = abs_pos % capacity   (Gqa.Cache.Bf16.cu:86-130)
//   * Q head h reads KV head h / (NH/NKV)                            (Gqa.Decode.Bf16.cu:93-98)
//   * a query at absolute position t sees keys max(0, t-window+1)..t when window > 0, else 0..t
//     (Gqa.Prefill.Bf16.cu:76-81; decode band Gqa.Decode.Bf16.cu:100-105 -- the same set)
//   * score = dot(q,k) * scale BEFORE max/exp                        (Gqa.Decode.Bf16.cu:212)
//   * fp32 scores / probabilities / accumulators, bf16 only at the final store.
//
// CDNA4 design of the decode kernel (HBM/latency bound: 4-8 MB of K/V per layer):
//   grid (splits, NKV, B*Tq), 256 threads = 4 waves.  A wave owns every 4th position of its
//   split; a lane owns HS/64 contiguous elements of every row (16-byte loads at HS = 512), so one
//   wave-instruction fetches one whole K (or V) row; q is kept packed (bf16 pairs) in registers
//   and multiplied with v_dot2_f32_bf16; the 64-lane score reduction is a xor butterfly; online
//   softmax state (m, l, O) lives in registers per wave and is merged across the 4 waves through
//   LDS four heads at a time; split partials (m, l, O) go to caller-provided scratch and are
//   merged by a second tiny kernel (a kernel boundary is cheaper than an in-kernel agent-scope
//   acquire on this chip, MI355X_MICROARCH "boundary" vs "barrier-xcd").
#include "common.h"
#include "rms_common.h"
#include "rope_common.h"
#include "internal.h"
#include "attention_generic.h"
#include "attention_tiles.h"

namespace mila {

constexpr int kMaxSplits = 64;
constexpr int kMaxSplitsMfma = 256;      // the long-context MFMA decode (attn_decode_mfma_kernel): one workgroup per CU
constexpr int kMfmaMinBand = 8192;        // band bucket (keys) from which a 16-head group on one KV head takes the MFMA decode (see band_bucket; profiles/r04_attn_band.txt)
static int g_mfma_min_band = kMfmaMinBand;      // tuning "attn.mfma_min_band" (keys)
MILA_TUNE("attn.mfma_min_band", g_mfma_min_band);

// ---- KV append ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv_write_bf16_kernel(uint16_t* __restrict__ Kc, uint16_t* __restrict__ Vc,
                                                            const uint16_t* __restrict__ k,
                                                            const uint16_t* __restrict__ v, int64_t total_vec,
                                                            int chunk, int NKV, int HS, int start_pos, int capacity)
{
    const int hv = HS / 8;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += stride)
    {
        // source order [b, t, nkv, hs]
        const int e = (int)(i % hv);
        int64_t r = i / hv;
        const int n = (int)(r % NKV);
        r /= NKV;
        const int t = (int)(r % chunk);
        const int b = (int)(r / chunk);
        const int row = (start_pos + t) % capacity;
        const size_t dst = ((((size_t)b * NKV + n) * capacity + row) * hv + e) * 8;
        st16(Kc + dst, ld16(k + i * 8));
        st16(Vc + dst, ld16(v + i * 8));
    }
}

// ---- decode attention ---------------------------------------------------------------------------------
struct AttnParams
{
    uint16_t* Y;              // [B, NH*HS]
    const uint16_t* Q;        // [B, NH*HS] post-norm/rope queries (unfused form), or NULL in the fused form
    int64_t q_b_stride;       // elements between two batches' query rows (0 = NH*HS; a packed [B, 1, 3C] MHA row: 3C)
    uint16_t* K;              // cache [B, NKV, capacity, HS]
    uint16_t* V;
    float* scratch;           // [B, NH, splits, HS+4] partials when splits > 1: O (HS) | m | l | pad (rows stay 16-byte aligned)
    int no_combine;           // leave the partials for the consumer (matvec_attn_combine) instead of launching the combine
    // warm ranges: extra workgroups of the attention launch (warm_a) and of the combine launch (warm_b) touch one dword per 128-byte
    // line of weights a LATER kernel streams, so the Infinity Cache fills while these latency-bound launches leave HBM idle
    const uint8_t* warm_a; int64_t warm_a_lines; int warm_a_blocks;   // blocks per grid row (blockIdx.x >= splits)
    const uint8_t* warm_b; int64_t warm_b_lines; int warm_b_blocks; int64_t warm_b_pair;
    uint32_t* tickets;        // one-pass form: arrival counters [B, NKV * head-groups], zero between launches; the workgroup whose
                              // partials arrive last combines its head-group's splits itself, so there is no combine launch
    int NH, NKV, capacity, position, window, splits;
    float scale;
    const int32_t* pos_dev;   // when set: the position is read from device memory (graph replay)
    int flat;                 // XCD-local grid (attn.xcd_local): blockIdx.x = group + ngroups * split, so that every split of a head group -- and the combine
                              // workgroups of its heads -- carry the same blockIdx.x % 8 and land on one XCD (workgroups are dealt round-robin over the 8 XCDs)
    // fused prologue (GemmaBlock::decode lines 315-337 folded in): raw projections + norm weights + RoPE cache
    int64_t raw_b_stride;     // elements between two batch rows of q_raw / k_raw / v_raw (a packed [B, 1, q | k | v] projection row; unused at B == 1)
    const uint16_t* q_raw;    // [NH*HS] per batch row
    const uint16_t* k_raw;    // [NKV*HS]
    const uint16_t* v_raw;    // [NKV*HS] (== k_raw on Gemma global layers)
    const uint16_t* qw;
    const uint16_t* kw;
    const uint16_t* vw;       // may be NULL (unit weight)
    const float* cos_cache;
    const float* sin_cache;
    float eps;
};

template <int EPL>
__device__ __forceinline__ void load_row(uint32_t (&dst)[EPL / 2], const uint16_t* p)
{
    if constexpr (EPL == 8)
    {
        const u32x4 v = ld16(p);
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
    else if constexpr (EPL == 4)
    {
        const u32x2 v = *reinterpret_cast<const u32x2*>(p);
        dst[0] = v[0]; dst[1] = v[1];
    }
    else
    {
        dst[0] = *reinterpret_cast<const uint32_t*>(p);
    }
}

constexpr int kDecodeWaves = 8;     // 512 threads per workgroup

// norm (+ RoPE) of one head row by one wave, canonical helpers => bit-identical to the standalone
// rmsnorm / rope kernels and to qkv_post_kernel.  dst_lds / dst_glb may be NULL.
template <int HS>
__device__ __forceinline__ void head_row_post(const uint16_t* __restrict__ src, const uint16_t* __restrict__ w, bool rotate,
                                              const float* cos_row, const float* sin_row, float eps,
                                              uint16_t* dst_lds, uint16_t* dst_glb)
{
    constexpr int hv = HS / 16;
    const int lane = threadIdx.x & 63;
    const bool act = lane < hv;
    const int l2 = act ? lane : 0;
    const u32x4 xlo = ld16(src + (size_t)l2 * 8), xhi = ld16(src + (size_t)(l2 + hv) * 8);
    u32x4 wlo = u32x4{0u, 0u, 0u, 0u}, whi = wlo;
    if (w) { wlo = ld16(w + (size_t)l2 * 8); whi = ld16(w + (size_t)(l2 + hv) * 8); }
    const float rstd = rms_rstd_wave(src, HS, eps);
    if (act)
    {
        u32x4 lo, hi;
        if (w) { lo = rms_apply8(xlo, wlo, rstd, 0.0f); hi = rms_apply8(xhi, whi, rstd, 0.0f); }
        else { lo = rms_apply8_now(xlo, rstd); hi = rms_apply8_now(xhi, rstd); }
        if (rotate) rope_rotate8_vals(lo, hi, cos_row, sin_row, lane * 8);
        if (dst_lds) { st16(dst_lds + (size_t)lane * 8, lo); st16(dst_lds + (size_t)(lane + hv) * 8, hi); }
        if (dst_glb) { st16(dst_glb + (size_t)lane * 8, lo); st16(dst_glb + (size_t)(lane + hv) * 8, hi); }
    }
}

// Combine of one head's split partials for output dim d by the lane that owns it: the arithmetic of attn_combine_kernel, shared by
// the standalone kernel (plain loads behind a kernel boundary) and the one-pass tail of attn_decode_kernel (SC1 = agent-coherent
// loads of partials other workgroups wrote through inside the same launch).  Lane s of the wave holds split s's (m, l).
template <bool SC1>
__device__ __forceinline__ float partial_ld(const float* p)
{
    if constexpr (SC1)
        return __hip_atomic_load((const __attribute__((address_space(1))) float*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        return *p;
}
template <bool SC1>
__device__ __forceinline__ uint16_t combine_dim(const float* __restrict__ base, int HS, int splits, int d)
{
    const int STR = HS + 4;
    const int s = threadIdx.x & 63;
    // every load of the first 16 splits is requested before anything is reduced: ONE memory round trip for splits <= 16
    float vals[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) vals[u] = (u < splits) ? partial_ld<SC1>(base + (size_t)u * STR + d) : 0.0f;
    const float ms = (s < splits) ? partial_ld<SC1>(base + (size_t)s * STR + HS) : -INFINITY;
    const float ls = (s < splits) ? partial_ld<SC1>(base + (size_t)s * STR + HS + 1) : 0.0f;
    float acc = 0.0f;
    const float M = wave_max(ms);
    const float fs = (ms == -INFINITY) ? 0.0f : __expf(ms - M);
    const float L = wave_sum(ls * fs);
    const int n8 = (splits + 7) & ~7;                      // the fma chain runs over whole batches of 8 (zeros past the last split)
    for (int i0 = 0; i0 < n8; i0 += 16)
    {
        if (i0 > 0)
        {
#pragma unroll
            for (int u = 0; u < 16; ++u) vals[u] = (i0 + u < splits) ? partial_ld<SC1>(base + (size_t)(i0 + u) * STR + d) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
        {
            const float f = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fs), min(i0 + u, 63)));
            if (i0 + u < n8) acc = fmaf(vals[u], f, acc);
        }
    }
    return f32_to_bf16_bits(L > 0.0f ? acc / L : 0.0f);
}
__device__ __forceinline__ void partial_st_sc1(float* p, float v)
{
    __hip_atomic_store((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one dword per 128-byte line of [base, base + 128 nlines), lines strided over `nthreads` threads; nothing is written
// pair != 0: the range is the head of TWO streams `pair` bytes apart (the gate and up halves the fused gate_up kernel reads in
// lockstep): line i is line i / 2 of stream i % 2
__device__ __forceinline__ void warm_lines(const uint8_t* __restrict__ base, int64_t nlines, int64_t tid, int64_t nthreads, float* never,
                                           int64_t pair = 0)
{
    uint32_t acc = 0;
    int64_t i = tid;
    auto at = [&](int64_t l) { return pair ? base + (l & 1) * pair + (l >> 1) * 128 : base + l * 128; };
    for (; i + 7 * nthreads < nlines; i += 8 * nthreads)
    {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint32_t*>(at(i + u * nthreads));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    for (; i < nlines; i += nthreads) acc ^= *reinterpret_cast<const uint32_t*>(at(i));
    if (acc == 0x12345679u && never != nullptr) never[0] = 1.0f;   // keeps the loads alive; never true in practice
}

// HS = 64 * EPL (EPL in {2,4,8}) or HS = 64 handled as EPL = 2 on 32 active lanes.
// GH = query heads per workgroup; grid = (splits, NKV * GS/GH, B); 8 waves, wave w owns positions
// begin + w + 8 j of the split.  FUSED: the q/k/v post-processing of the token is done in the prologue,
// overlapped with the first K/V round trip; the new K/V row is appended to the cache by the split that owns it.
template <int HS, int GH, bool FUSED>
__global__ __launch_bounds__(kDecodeWaves * 64) void attn_decode_kernel(const AttnParams p)
{
    constexpr int NW = kDecodeWaves;
    constexpr int EPL = (HS >= 128) ? HS / 64 : 2;
    constexpr int NPAIR = EPL / 2;
    constexpr int ACTIVE = HS / EPL;                       // lanes that own data (64, or 32 for HS = 64)
    constexpr int STR = HS + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);        // [NW][GH][HS + 2]
    uint16_t* qs = reinterpret_cast<uint16_t*>(smem_raw + (size_t)NW * GH * STR * sizeof(float));   // [GH][HS], then k_new, v_new

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool owner = lane < ACTIVE;
    const int GS = p.NH / p.NKV, hgroups = GS / GH;
    if (!p.flat && (int)blockIdx.x >= p.splits)             // a warm block (block-uniform): no barrier is shared with the others
    {
        const int64_t wb = ((int64_t)(blockIdx.x - p.splits) * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z;
        warm_lines(p.warm_a, p.warm_a_lines, wb * (NW * 64) + threadIdx.x, (int64_t)p.warm_a_blocks * gridDim.y * gridDim.z * (NW * 64),
                   p.scratch);
        return;
    }
    const int ngrp = p.NKV * hgroups;
    const int split = p.flat ? (int)blockIdx.x / ngrp : (int)blockIdx.x, grp = p.flat ? (int)blockIdx.x % ngrp : (int)blockIdx.y;
    const int kvh = grp / hgroups, hg = grp % hgroups;
    const int h0 = kvh * GS + hg * GH;                     // first query head of this workgroup
    const int b = blockIdx.z;
    const int pos = p.pos_dev ? *p.pos_dev : p.position;
    const int len = pos + 1;
    const int band_begin = (p.window > 0) ? max(0, len - p.window) : 0;
    const int band = len - band_begin;
    const int chunk = (band + p.splits - 1) / p.splits;
    const int begin = band_begin + split * chunk;
    const int end = min(begin + chunk, len);
    const bool owns_new = FUSED && pos >= begin && pos < end;   // this split appends (and consumes) the new K/V row

    uint16_t* kbase = p.K + ((size_t)b * p.NKV + kvh) * p.capacity * HS;
    uint16_t* vbase = p.V + ((size_t)b * p.NKV + kvh) * p.capacity * HS;

    constexpr int PG = (HS >= 512) ? 4 : 8;
    struct KVG { uint32_t k[PG][NPAIR], v[PG][NPAIR]; };
    auto load_group = [&](KVG& gbuf, int base) {
#pragma unroll
        for (int j = 0; j < PG; ++j)
        {
            const int pp = base + NW * j;
            // in the fused form row `pos` is not in the cache yet: it is patched in from LDS below
            if (owner && pp < end && !(FUSED && pp == pos))
            {
                const size_t r = (size_t)(pp % p.capacity) * HS + lane * EPL;
                load_row<EPL>(gbuf.k[j], kbase + r);
                load_row<EPL>(gbuf.v[j], vbase + r);
            }
            else
            {
#pragma unroll
                for (int e = 0; e < NPAIR; ++e) { gbuf.k[j][e] = 0u; gbuf.v[j][e] = 0u; }
            }
        }
    };

    int base = begin + wave;
    KVG ga, gb;
    if (base < end) load_group(ga, base);                  // in flight during the prologue

    // ---- q (and the new K/V row) ----
    uint32_t q[GH][NPAIR];
    if constexpr (FUSED)
    {
        const int half = HS / 2;
        const float* cos_row = p.cos_cache + (size_t)pos * half;
        const float* sin_row = p.sin_cache + (size_t)pos * half;
        const int nrows = GH + (owns_new ? 2 : 0);
        for (int r = wave; r < nrows; r += NW)
        {
            if (r < GH)
                head_row_post<HS>(p.q_raw + (size_t)b * p.raw_b_stride + (size_t)(h0 + r) * HS, p.qw, true, cos_row, sin_row, p.eps, qs + (size_t)r * HS, nullptr);
            else if (r == GH)
                head_row_post<HS>(p.k_raw + (size_t)b * p.raw_b_stride + (size_t)kvh * HS, p.kw, true, cos_row, sin_row, p.eps, qs + (size_t)GH * HS,
                                  (hg == 0) ? kbase + (size_t)(pos % p.capacity) * HS : nullptr);
            else
                head_row_post<HS>(p.v_raw + (size_t)b * p.raw_b_stride + (size_t)kvh * HS, p.vw, false, cos_row, sin_row, p.eps, qs + (size_t)(GH + 1) * HS,
                                  (hg == 0) ? vbase + (size_t)(pos % p.capacity) * HS : nullptr);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < GH; ++g)
        {
            if (owner) load_row<EPL>(q[g], qs + (size_t)g * HS + lane * EPL);
            else
            {
#pragma unroll
                for (int e = 0; e < NPAIR; ++e) q[g][e] = 0u;
            }
        }
    }
    else
    {
#pragma unroll
        for (int g = 0; g < GH; ++g)
        {
            const uint16_t* qp = p.Q + (size_t)b * (p.q_b_stride ? (size_t)p.q_b_stride : (size_t)p.NH * HS) + (size_t)(h0 + g) * HS + lane * EPL;
            if (owner) load_row<EPL>(q[g], qp);
            else
            {
#pragma unroll
                for (int e = 0; e < NPAIR; ++e) q[g][e] = 0u;
            }
        }
    }

    float m[GH], l[GH], o[GH][EPL];
#pragma unroll
    for (int g = 0; g < GH; ++g)
    {
        m[g] = -INFINITY;
        l[g] = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[g][e] = 0.0f;
    }

    auto patch_new = [&](KVG& gbuf, int base_) {           // put the LDS copy of row `pos` into its slot
        if constexpr (FUSED)
        {
            if (owns_new)
            {
#pragma unroll
                for (int j = 0; j < PG; ++j)
                    if (base_ + NW * j == pos && owner)
                    {
                        load_row<EPL>(gbuf.k[j], qs + (size_t)GH * HS + lane * EPL);
                        load_row<EPL>(gbuf.v[j], qs + (size_t)(GH + 1) * HS + lane * EPL);
                    }
            }
        }
    };
    auto compute_group = [&](const KVG& gbuf, int base_) {
        float sc[PG][GH];
#pragma unroll
        for (int j = 0; j < PG; ++j)
#pragma unroll
            for (int g = 0; g < GH; ++g)
            {
                float a = 0.0f;
#pragma unroll
                for (int e = 0; e < NPAIR; ++e) a = dot2_bf16(as_bf16x2(q[g][e]), as_bf16x2(gbuf.k[j][e]), a);
                sc[j][g] = a;
            }
#pragma unroll
        for (int j = 0; j < PG; ++j)
#pragma unroll
            for (int g = 0; g < GH; ++g) sc[j][g] = wave_sum(sc[j][g]);
#pragma unroll
        for (int g = 0; g < GH; ++g)
        {
            float a[PG], mt = -INFINITY;
#pragma unroll
            for (int j = 0; j < PG; ++j)
            {
                a[j] = (base_ + NW * j < end) ? sc[j][g] * p.scale : -INFINITY;
                mt = fmaxf(mt, a[j]);
            }
            const float mn = fmaxf(m[g], mt);
            const float msafe = (mn == -INFINITY) ? 0.0f : mn;
            const float alpha = __expf(m[g] - msafe);        // m = -inf first time: exp(-inf) = 0
            float ex[PG], rs = 0.0f;
#pragma unroll
            for (int j = 0; j < PG; ++j) { ex[j] = __expf(a[j] - msafe); rs += ex[j]; }
            l[g] = l[g] * alpha + rs;
            m[g] = mn;
#pragma unroll
            for (int e = 0; e < NPAIR; ++e)
            {
                float lo = o[g][2 * e] * alpha, hi = o[g][2 * e + 1] * alpha;
#pragma unroll
                for (int j = 0; j < PG; ++j)
                {
                    lo = fmaf(ex[j], bf16_lo(gbuf.v[j][e]), lo);
                    hi = fmaf(ex[j], bf16_hi(gbuf.v[j][e]), hi);
                }
                o[g][2 * e] = lo;
                o[g][2 * e + 1] = hi;
            }
        }
    };
    for (;;)
    {
        if (base >= end) break;
        int nb = base + NW * PG;
        if (nb < end) load_group(gb, nb);
        patch_new(ga, base);
        compute_group(ga, base);
        base = nb;
        if (base >= end) break;
        nb = base + NW * PG;
        if (nb < end) load_group(ga, nb);
        patch_new(gb, base);
        compute_group(gb, base);
        base = nb;
    }

    // ---- merge the NW waves through LDS; wave w < GH finalises head w ----
#pragma unroll
    for (int g = 0; g < GH; ++g)
    {
        float* dst = sm + ((size_t)wave * GH + g) * STR;
        if (owner)
        {
#pragma unroll
            for (int e = 0; e < EPL; ++e) dst[lane * EPL + e] = o[g][e];
        }
        if (lane == 0) { dst[HS] = m[g]; dst[HS + 1] = l[g]; }
    }
    __syncthreads();
    if (wave < GH)
    {
        const int g = wave;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, sm[((size_t)w * GH + g) * STR + HS]);
        float L = 0.0f, acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
        {
            const float* src = sm + ((size_t)w * GH + g) * STR;
            const float mw = src[HS];
            const float f = (mw == -INFINITY) ? 0.0f : __expf(mw - M);
            L += src[HS + 1] * f;
            if (owner)
            {
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] += src[lane * EPL + e] * f;
            }
        }
        const int h = h0 + g;
        if (p.splits == 1)
        {
            const float inv = (L > 0.0f) ? 1.0f / L : 0.0f;
            uint16_t* y = p.Y + ((size_t)b * p.NH + h) * HS + lane * EPL;
            if (owner)
            {
#pragma unroll
                for (int e = 0; e < EPL; e += 2)
                    *reinterpret_cast<uint32_t*>(y + e) = pack_bf16x2(acc[e] * inv, acc[e + 1] * inv);
            }
        }
        else if (p.tickets == nullptr)
        {
            float* dst = p.scratch + (((size_t)b * p.NH + h) * p.splits + split) * (HS + 4);
            if (owner)
            {
#pragma unroll
                for (int e = 0; e < EPL; ++e) dst[lane * EPL + e] = acc[e];
            }
            if (lane == 0) { dst[HS] = M; dst[HS + 1] = L; }
        }
        else
        {
            // one-pass form: write-through (sc1) stores, drained by this wave before the workgroup's arrival is counted
            float* dst = p.scratch + (((size_t)b * p.NH + h) * p.splits + split) * (HS + 4);
            if (owner)
            {
#pragma unroll
                for (int e = 0; e < EPL; ++e) partial_st_sc1(dst + lane * EPL + e, acc[e]);
            }
            if (lane == 0) { partial_st_sc1(dst + HS, M); partial_st_sc1(dst + HS + 1, L); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (p.splits > 1 && p.tickets != nullptr)
    {
        // Arrival ticket per (batch row, head-group): MI355X_MICROARCH.md hand-off table, row 1 (sc1 payload stores drained by
        // every storing wave, a workgroup barrier, ONE lane's agent-scope atomic add; the workgroup whose add came last reads
        // with sc1 loads, its other waves behind a barrier the adding wave joins).  The last arriver re-arms the counter.
        __shared__ int s_last;
        __syncthreads();
        uint32_t* tk = p.tickets + (size_t)b * gridDim.y + blockIdx.y;
        if (threadIdx.x == 0)
        {
            // acquire-release at agent scope on top of the drained write-through stores and the sc1 loads: the hand-off then
            // holds by the memory model too, not only by the measured behaviour of the sc1 forms (this path is not the fast one)
            const uint32_t t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == (uint32_t)(p.splits - 1));
            if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (s_last)
        {
            for (int idx = threadIdx.x; idx < GH * HS; idx += NW * 64)
            {
                const int g = idx / HS, d = idx % HS;       // a wave's 64 dims lie in one head (HS % 64 == 0)
                const int h = h0 + g;
                const float* base = p.scratch + ((size_t)b * p.NH + h) * p.splits * (HS + 4);
                p.Y[((size_t)b * p.NH + h) * HS + d] = combine_dim<true>(base, HS, p.splits, d);
            }
        }
    }
}

// combine split partials: grid (NH, B, HS/64), 64 threads -> 64 dims each; splits <= 64.
// Every thread reads all (m, l) pairs itself (broadcast loads) and all partial values of its dim in
// unrolled batches, so the kernel is two dependent memory round trips long.
// flat_gh > 0 (attn.xcd_local): grid (NH * HS / 64, B): blockIdx.x = group + ngroups * (head in group + flat_gh * dim chunk), groups of flat_gh heads -- the
// workgroups of a head group share blockIdx.x % 8 with the decode workgroups that wrote its partials
__global__ __launch_bounds__(64) void attn_combine_kernel(uint16_t* __restrict__ Y, const float* __restrict__ scratch, int NH,
                                                          int HS, int splits, const uint8_t* __restrict__ warm, int64_t warm_lines_n, int64_t warm_pair, int flat_gh)
{
    int h = blockIdx.x, zc = blockIdx.z;
    const int b = blockIdx.y;
    if (flat_gh > 0)
    {
        const int ngroups = NH / flat_gh, grp = (int)blockIdx.x % ngroups, r = (int)blockIdx.x / ngroups;
        h = grp * flat_gh + r % flat_gh;
        zc = r / flat_gh;
    }
    else if ((int)blockIdx.z >= HS / 64)                    // warm blocks (see AttnParams::warm_b)
    {
        const int64_t wb = ((int64_t)(blockIdx.z - HS / 64) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        warm_lines(warm, warm_lines_n, wb * 64 + threadIdx.x, (int64_t)(gridDim.z - HS / 64) * gridDim.y * gridDim.x * 64, nullptr, warm_pair);
        return;
    }
    const int d = zc * 64 + threadIdx.x;
    const float* base = scratch + ((size_t)b * NH + h) * splits * (HS + 4);
    Y[((size_t)b * NH + h) * HS + d] = combine_dim<false>(base, HS, splits, d);
}


// ---- long-context decode on the matrix cores (round 3) ------------------------------------------------------------------------------------------
// Gemma's global layers put 16 query heads on ONE KV head: at one decode position those 16 heads ARE a 16-row MFMA tile, and K / V rows are shared by all of them.
// The wave-per-position kernel above costs a 64-lane reduction per (head, key): at 32 768 keys it is vector-ALU-bound, 63.7 us for 67 MB of K / V (1.05 TB/s,
// profiles/r03_long_context.txt) -- and its 8 head-groups of 2 read every row 8 times from L2.  Here a workgroup takes one split of the keys for all 16 heads:
// K / V tiles of 32 keys go global -> registers -> LDS (double-buffered images of attention_tiles.h, the next tile's loads in flight during this tile's products),
// S^T = K Q^T and O^T += V^T P^T run as in the flash prefill (transposed products: a head's keys sit in its lane's registers, the row statistics are in-lane plus two
// permlane steps, the exponentiated scores ARE the second product's B operand).  The four waves split the OUTPUT dimensions (each computes the whole S^T and softmax
// of the tile -- identical in all four -- and a quarter of O^T: 32 accumulator registers).  Partials in the layout of the scalar kernel: O (unnormalised, relative
// to M) | M | L per (head, split); up to 256 splits, merged by attn_combine_many_kernel.  fp32 scores and statistics, P rounded to bf16 for the PV product (as the
// flash kernels do): within 1 bf16 ulp of the double-precision oracle like them, NOT bit-identical to the scalar kernel -- the choice depends on (window, capacity)
// only, so every path of a model (reference order, fused, graph replay) takes the same kernel.
template <int HS>
__global__ __launch_bounds__(256) void attn_decode_mfma_kernel(const AttnParams p)
{
    constexpr int KSTEPS = HS / 32, DT = HS / 16, DTW = DT / 4;
    constexpr int ROWB = HS * 2, TILE_BYTES = kKeysPerTile * ROWB;
    constexpr int CH = (kKeysPerTile * (ROWB / 16)) / 256;          // 16-byte chunks each thread stages per tile and operand
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_mfma[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int GS = p.NH / p.NKV, n16 = GS / 16;
    const int split = blockIdx.x, kvh = blockIdx.y / n16, h0 = kvh * GS + (blockIdx.y % n16) * 16, b = blockIdx.z;
    const int pos = p.pos_dev ? *p.pos_dev : p.position;
    const int len = pos + 1;
    const int band_begin = (p.window > 0) ? max(0, len - p.window) : 0;
    const int band = len - band_begin;
    const int chunk = (band + p.splits - 1) / p.splits;
    const int begin = band_begin + split * chunk;
    const int end = min(begin + chunk, len);
    float* part = p.scratch + (((size_t)b * p.NH + h0 + l15) * p.splits + split) * (HS + 4);      // this lane's head row

    f32x4 o[DTW];
#pragma unroll
    for (int d = 0; d < DTW; ++d) o[d] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float m_run = -INFINITY, l_run = 0.0f;

    if (begin < end)      // workgroup-uniform
    {
        // Q: the 16 heads' rows as a 16-row image in LDS (the layout of a K tile's first 16 rows); the fragments of a k-step are read per tile -- held in registers they
        // cost 64 VGPRs, which buy the second staging set (two tiles of K / V in flight per workgroup instead of one: the kernel is a latency chain of 4-8 tiles)
        unsigned char* ldsQ = smem_mfma + 4 * TILE_BYTES;
        {
            const uint16_t* qb = p.Q + (size_t)b * (p.q_b_stride ? (size_t)p.q_b_stride : (size_t)p.NH * HS) + (size_t)h0 * HS;
#pragma unroll
            for (int i = 0; i < (16 * (ROWB / 16)) / 256; ++i)
            {
                const int c = tid + 256 * i;
                const int row = c / (ROWB / 16), ch = c % (ROWB / 16);
                *reinterpret_cast<u32x4*>(ldsQ + k_off<HS>(row, ch)) = ld16(qb + (size_t)row * HS + ch * 8);
            }
        }
        const uint16_t* kbase = p.K + ((size_t)b * p.NKV + kvh) * p.capacity * HS;
        const uint16_t* vbase = p.V + ((size_t)b * p.NKV + kvh) * p.capacity * HS;
        const bool wraps = end > p.capacity;                          // uniform: an unbounded cache (the global layers) needs no modulo per row
        struct StageRegs { u32x4 k[CH], v[CH]; };
        const int ntiles = (end - begin + kKeysPerTile - 1) / kKeysPerTile;
        const int kt_last = begin + (ntiles - 1) * kKeysPerTile;
        auto stage_load = [&](StageRegs& r, int kt) {
            kt = min(kt, kt_last);                                    // a load past the last tile re-reads it (never stored): branch-free, the waits stay counted
#pragma unroll
            for (int i = 0; i < CH; ++i)
            {
                const int c = tid + 256 * i;
                const int row = c / (ROWB / 16), ch = c % (ROWB / 16);
                const int key = min(kt + row, end - 1);                 // rows past the split re-read its last key (masked below; a real, finite V row)
                const size_t off = (size_t)(wraps ? key % p.capacity : key) * HS + ch * 8;
                r.k[i] = ld16(kbase + off);
                r.v[i] = ld16(vbase + off);
            }
        };
        auto stage_store = [&](const StageRegs& r, unsigned char* ldsK, unsigned char* ldsV) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
            {
                const int c = tid + 256 * i;
                const int row = c / (ROWB / 16), ch = c % (ROWB / 16);
                *reinterpret_cast<u32x4*>(ldsK + k_off<HS>(row, ch)) = r.k[i];
                *reinterpret_cast<u32x4*>(ldsV + v_off<HS>(row, ch)) = r.v[i];
            }
        };
        StageRegs ra, rb;
        stage_load(ra, begin);
        stage_load(rb, begin + kKeysPerTile);
        auto tile = [&](int t, StageRegs& regs) {
            const int kt = begin + t * kKeysPerTile;
            // two [K | V] buffers: tile t is stored while slower waves may still read tile t - 1 from the other one (the store follows the barrier of tile t - 1,
            // which every wave reaches only after its reads of tile t - 2): one barrier per tile, which also publishes the Q image before its first read
            unsigned char* ldsK = smem_mfma + (t & 1) * 2 * TILE_BYTES;
            unsigned char* ldsV = ldsK + TILE_BYTES;
            stage_store(regs, ldsK, ldsV);
            __syncthreads();
            stage_load(regs, kt + 2 * kKeysPerTile);                  // two tiles ahead, in flight during this tile's and the next one's products

            f32x4 s0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
            {
                const s16x8 qf = *reinterpret_cast<const s16x8*>(ldsQ + k_off<HS>(l15, 4 * s + g));
                const s16x8 ka = *reinterpret_cast<const s16x8*>(ldsK + k_off<HS>(l15, 4 * s + g));
                const s16x8 kb = *reinterpret_cast<const s16x8*>(ldsK + k_off<HS>(16 + l15, 4 * s + g));
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka), __builtin_bit_cast(bf16x8, qf), s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb), __builtin_bit_cast(bf16x8, qf), s1, 0, 0, 0);
            }
            // lane holds keys kt + 4 g + r (s0) and kt + 16 + 4 g + r (s1) of head row l15
            float sv[8], mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                const int key = kt + ((r < 4) ? (4 * g + r) : (16 + 4 * g + (r - 4)));
                const float raw = (r < 4) ? s0[r] : s1[r - 4];
                sv[r] = (key < end) ? raw * p.scale : -INFINITY;
                mt = fmaxf(mt, sv[r]);
            }
            mt = quad_rows_max(mt);
            const float mn = fmaxf(m_run, mt);                      // finite: every tile holds at least one key of the split
            const float alpha = __expf(m_run - mn);                 // m_run = -inf -> 0
            float pe[8], rs = 0.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                pe[r] = __expf(sv[r] - mn);
                rs += pe[r];
            }
            rs = quad_rows_sum(rs);
            l_run = l_run * alpha + rs;
            m_run = mn;
            u32x4 pb;
            pb[0] = pack_bf16x2(pe[0], pe[1]);
            pb[1] = pack_bf16x2(pe[2], pe[3]);
            pb[2] = pack_bf16x2(pe[4], pe[5]);
            pb[3] = pack_bf16x2(pe[6], pe[7]);
            const bf16x8 pfrag = __builtin_bit_cast(bf16x8, pb);
            const bool rescale = __any(alpha != 1.0f);
#pragma unroll
            for (int dd = 0; dd < DTW; ++dd)
            {
                const int d = wave * DTW + dd;
                const int q4 = l15 >> 2, pp = l15 & 3;
                const int col = 16 * d + 4 * pp;
                const int r_lo = 4 * g + q4, r_hi = 16 + 4 * g + q4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(ldsV + v_off<HS>(r_lo, col >> 3) + ((col & 7) << 1)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(ldsV + v_off<HS>(r_hi, col >> 3) + ((col & 7) << 1)));
                s16x8 va;
                va[0] = lo[0]; va[1] = lo[1]; va[2] = lo[2]; va[3] = lo[3];
                va[4] = hi[0]; va[5] = hi[1]; va[6] = hi[2]; va[7] = hi[3];
                if (rescale) { o[dd][0] *= alpha; o[dd][1] *= alpha; o[dd][2] *= alpha; o[dd][3] *= alpha; }
                o[dd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va), pfrag, o[dd], 0, 0, 0);
            }
        };
        for (int t = 0; t < ntiles; t += 2)
        {
            tile(t, ra);
            if (t + 1 < ntiles) tile(t + 1, rb);
        }
    }
    // O^T[dim 16 d + 4 g + r][head l15] -> this head's partial row; (M, L) once per head (an empty split leaves O = 0, M = -inf, L = 0: the merge ignores it)
#pragma unroll
    for (int dd = 0; dd < DTW; ++dd) *reinterpret_cast<f32x4*>(part + 16 * (wave * DTW + dd) + 4 * g) = o[dd];
    if (wave == 0 && g == 0) { part[HS] = m_run; part[HS + 1] = l_run; }
}

// merge of up to kMaxSplitsMfma partials per head: grid (NH, B, HS / 64), 4 waves; wave w merges splits 64 w .. 64 w + 63 as attn_combine_kernel does (lane s holds
// split s's (M, L); every load of a batch of 16 splits requested before anything is reduced), the four (acc, M, L) triples meet in LDS.  Fixed order: a replay
// reproduces an eager launch.  (A first form -- 64 threads walking all 256 splits with broadcast loads -- took 51 us for 8.4 MB of partials.)
__global__ __launch_bounds__(256) void attn_combine_many_kernel(uint16_t* __restrict__ Y, const float* __restrict__ scratch, int NH, int HS, int splits)
{
    __shared__ float s_acc[4][64], s_m[4], s_l[4];
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = blockIdx.z * 64 + lane;
    const int STR = HS + 4;
    const int s0 = wave * 64, n = min(64, splits - s0);          // this wave's splits (n <= 0: none)
    const float* base = scratch + (((size_t)b * NH + h) * splits + s0) * STR;
    float acc = 0.0f, M = -INFINITY, L = 0.0f;
    if (n > 0)
    {
        float vals[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) vals[u] = (u < n) ? base[(size_t)u * STR + d] : 0.0f;
        const float ms = (lane < n) ? base[(size_t)lane * STR + HS] : -INFINITY;
        const float ls = (lane < n) ? base[(size_t)lane * STR + HS + 1] : 0.0f;
        M = wave_max(ms);
        const float fs = (ms == -INFINITY) ? 0.0f : __expf(ms - M);
        L = wave_sum(ls * fs);
        const int n16 = (n + 15) & ~15;
        for (int i0 = 0; i0 < n16; i0 += 16)
        {
            if (i0 > 0)
            {
#pragma unroll
                for (int u = 0; u < 16; ++u) vals[u] = (i0 + u < n) ? base[(size_t)(i0 + u) * STR + d] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
            {
                const float f = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fs), min(i0 + u, 63)));
                acc = fmaf(vals[u], f, acc);
            }
        }
    }
    s_acc[wave][lane] = acc;
    if (lane == 0) { s_m[wave] = M; s_l[wave] = L; }
    __syncthreads();
    if (wave == 0)
    {
        float Mt = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float Lt = 0.0f, out = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
        {
            const float f = (s_m[w] == -INFINITY) ? 0.0f : __expf(s_m[w] - Mt);
            Lt = fmaf(s_l[w], f, Lt);
            out = fmaf(s_acc[w][lane], f, out);
        }
        Y[((size_t)b * NH + h) * HS + d] = f32_to_bf16_bits(Lt > 0.0f ? out / Lt : 0.0f);
    }
}

static int g_tune_decode_mfma = 1;      // tuning "attn.mfma_decode": 0 = never take the MFMA decode
MILA_TUNE("attn.mfma_decode", g_tune_decode_mfma);

static bool mfma_decode_applies(int NH, int NKV, int HS, int band_max)
{
    return g_tune_decode_mfma && HS == 512 && NKV > 0 && (NH / NKV) % 16 == 0 && band_max >= g_mfma_min_band;
}
static int mfma_decode_splits(int B, int NH, int NKV, int band_max)
{
    const int groups = NKV * ((NH / NKV) / 16) * B;
    int s = (band_max + 127) / 128;                      // at least four key tiles per split at the full band
    const int cap = max(1, kNumCU / groups);
    if (s > cap) s = cap;
    if (s > kMaxSplitsMfma) s = kMaxSplitsMfma;
    return max(s, 1);
}
static int launch_decode_mfma(const AttnParams& p, int B, hipStream_t s)
{
    constexpr int HS = 512;
    note_form("attn_decode_mfma");
    static bool attr_set = false;
    const size_t lds = 4 * (size_t)kKeysPerTile * HS * 2 + 16 * (size_t)HS * 2;      // two [K | V] tile pairs + the 16 heads' Q rows
    if (!attr_set)
    {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_mfma_kernel<HS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(attn_decode_mfma)");
        if (rc) return rc;
        attr_set = true;
    }
    const int n16 = (p.NH / p.NKV) / 16;
    hipLaunchKernelGGL((attn_decode_mfma_kernel<HS>), dim3(p.splits, p.NKV * n16, B), dim3(256), lds, s, p);
    int rc = check_hip(hipGetLastError(), "attn_decode_mfma");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_combine_many_kernel, dim3(p.NH, B, HS / 64), dim3(256), 0, s, p.Y, p.scratch, p.NH, HS, p.splits);
    return check_hip(hipGetLastError(), "attn_combine_many");
}

static int g_attn_gh512 = 2;      // query heads per workgroup at HS 512 (experiment: 4 = half the head groups re-reading each K / V row, twice the accumulators per wave)
MILA_TUNE("attn.heads_per_group_512", g_attn_gh512);
static int heads_per_group(int GS, int HS) { return HS >= 512 ? (GS >= g_attn_gh512 && g_attn_gh512 == 4 ? 4 : (GS >= 2 ? 2 : 1)) : (GS >= 4 ? 4 : GS); }

template <int HS, int GH, bool FUSED>
static int launch_decode(const AttnParams& p, int B, hipStream_t s)
{
    const int hgroups = (p.NH / p.NKV) / GH;
    note_form("attn_decode");
    const size_t lds = (size_t)kDecodeWaves * GH * (HS + 2) * sizeof(float) + (size_t)(GH + 2) * HS * 2;
    const int wa = (p.warm_a && p.warm_a_lines > 0) ? p.warm_a_blocks : 0;
    if (p.flat)
    {
        // (only taken without warm blocks, tickets or a caller-side combine: run_decode)
        hipLaunchKernelGGL((attn_decode_kernel<HS, GH, FUSED>), dim3(p.NKV * hgroups * p.splits, 1, B), dim3(kDecodeWaves * 64), lds, s, p);
        int rc = check_hip(hipGetLastError(), "attn_decode (xcd-local)");
        if (rc || p.splits <= 1) return rc;
        hipLaunchKernelGGL(attn_combine_kernel, dim3(p.NH * (HS / 64), B, 1), dim3(64), 0, s, p.Y, p.scratch, p.NH, HS, p.splits, nullptr, 0, 0, GH);
        return check_hip(hipGetLastError(), "attn_combine (xcd-local)");
    }
    hipLaunchKernelGGL((attn_decode_kernel<HS, GH, FUSED>), dim3(p.splits + wa, p.NKV * hgroups, B), dim3(kDecodeWaves * 64), lds, s, p);
    int rc = check_hip(hipGetLastError(), "attn_decode");
    if (rc) return rc;
    if (p.splits > 1 && !p.no_combine && !p.tickets)
    {
        const int wbk = (p.warm_b && p.warm_b_lines > 0) ? p.warm_b_blocks : 0;
        hipLaunchKernelGGL(attn_combine_kernel, dim3(p.NH, B, HS / 64 + wbk), dim3(64), 0, s, p.Y, p.scratch, p.NH, HS, p.splits, p.warm_b,
                           p.warm_b_lines, p.warm_b_pair, 0);
        rc = check_hip(hipGetLastError(), "attn_combine");
    }
    return rc;
}

template <int HS, bool FUSED>
static int dispatch_gs(const AttnParams& p, int B, hipStream_t s)
{
    const int GS = p.NH / p.NKV;
    if (GS != 1 && GS != 2 && GS != 4 && GS != 8 && GS != 16 && GS != 32)
        return set_error(MILA_E_UNSUPPORTED, "attention: group size %d (NH/NKV) must be 1,2,4,8,16 or 32", GS);
    switch (heads_per_group(GS, HS))
    {
        case 1: return launch_decode<HS, 1, FUSED>(p, B, s);
        case 2: return launch_decode<HS, 2, FUSED>(p, B, s);
        default: return launch_decode<HS, 4, FUSED>(p, B, s);
    }
}

template <bool FUSED>
static int dispatch_hs(int HS, const AttnParams& p, int B, hipStream_t s)
{
    switch (HS)
    {
        case 64: return dispatch_gs<64, FUSED>(p, B, s);
        case 128: return dispatch_gs<128, FUSED>(p, B, s);
        case 256: return dispatch_gs<256, FUSED>(p, B, s);
        case 512: return dispatch_gs<512, FUSED>(p, B, s);
        default:
            if constexpr (!FUSED)
            {
                // any other head size: one wave per (batch, head) row on the generic kernel (attention_generic.hip); the position may live on the device only in the
                // captured forms, which the benchmarked head sizes alone use
                if (p.pos_dev) return set_error(MILA_E_UNSUPPORTED, "attention: the device-position form needs a head size of 64, 128, 256 or 512 (got %d)", HS);
                const int64_t qbs = p.q_b_stride ? p.q_b_stride : (int64_t)p.NH * HS;
                GenericAttnParams g{p.Y, p.Q, p.K, p.V, qbs, qbs, (int64_t)p.NKV * p.capacity * HS, (int64_t)p.capacity * HS, HS,
                                    B, 1, p.NH, p.NKV, HS, p.capacity, p.position, p.window, p.scale};
                return launch_attn_generic(g, s);
            }
            return set_error(MILA_E_UNSUPPORTED, "attention: head size %d must be 64, 128, 256 or 512", HS);
    }
}

static int g_tune_positions_per_split = 64;     // tuning "attn.positions_per_split": positions one split covers
MILA_TUNE("attn.positions_per_split", g_tune_positions_per_split);
// Experiment (round 4): every split of a head group and the combine workgroups of its heads on ONE XCD (blockIdx.x % 8 equal), so that the merge
// reads the partials from that XCD's L2 instead of across the fabric.  Same arithmetic per workgroup: bit-identical.  Result: profiles/r04_attn_pair_experiments.txt
static int g_attn_xcd_local = 0;
MILA_TUNE("attn.xcd_local", g_attn_xcd_local);
static int g_attn_max_wgs = 256;                // decode_splits: the split count is capped so that a launch has about this many workgroups
MILA_TUNE("attn.max_workgroups", g_attn_max_wgs);

static int decode_splits(int B, int NH, int NKV, int HS, int band)
{
    // ~256 workgroups of 8 waves, 64 positions (8 per wave) per split
    const int GS = NH / NKV;
    const int hgroups = GS / heads_per_group(GS, HS);
    int cap = g_attn_max_wgs / (NKV * hgroups * B);
    if (cap < 1) cap = 1;
    const int pps = g_tune_positions_per_split >= 8 ? g_tune_positions_per_split : 64;      // (a tuning value below 8 means the default)
    int s = (band + pps - 1) / pps;
    if (s > cap) s = cap;
    if (s > kMaxSplits) s = kMaxSplits;
    if (s < 1) s = 1;
    return s;
}

// The launch geometry of an unwindowed layer (split count, and whether the matrix-core decode serves it) is chosen from a BUCKET of the live length, not from the cache
// capacity: 4096, 8192, 16384, ... keys, clipped to the capacity.  (chosen from the capacity alone, a 32K-capacity cache decoding at position 2K ran the
// matrix-core kernel with 256 splits of 8 keys each -- 28.5 us against the scalar kernel's 23.9; the crossover is at ~8K keys, profiles/r04_attn_band.txt.)  A bucket is a
// function of the position only, so eager launches and a captured graph agree as long as the graph is re-captured when the position leaves its bucket
// (GemmaTransformer::ensureGraph does); inside a bucket the geometry is static, which is what a graph needs.  Caches of <= 4096 rows (the benchmark's) have one bucket.
static int band_bucket(int len, int capacity)
{
    int b = 4096;
    while (b < len && b < capacity) b <<= 1;
    return min(b, capacity);
}

// len_hint: an upper bound on the live length this launch is for -- position + 1 in the eager forms; in the device-position forms what the caller captured the graph
// for (0 = the capacity)
static int run_decode(uint16_t* Y, const uint16_t* Q, uint16_t* Kc, uint16_t* Vc, void* scratch, size_t scratch_bytes, int B,
                      int NH, int NKV, int HS, int capacity, int position, const int32_t* pos_dev, int window, float scale,
                      const AttnParams* fused, const char* who, hipStream_t stream, int len_hint = 0)
{
    AttnParams p{};
    if (fused) p = *fused;
    p.Y = Y; p.Q = Q; p.K = Kc; p.V = Vc; p.scratch = reinterpret_cast<float*>(scratch);
    p.NH = NH; p.NKV = NKV; p.capacity = capacity; p.position = position; p.window = window;
    // the split count depends only on (window, band bucket), never on the current length inside a bucket, so that eager
    // launches and a graph captured once (the _devpos forms) reduce in the same order: bit-identical
    const int hint = pos_dev ? len_hint : position + 1;
    const int band_limit = hint > 0 ? band_bucket(hint, capacity) : capacity;
    const int band_max = (window > 0 && window < capacity) ? window : band_limit;
    p.splits = decode_splits(B, NH, NKV, HS, band_max);
    p.scale = scale;
    p.pos_dev = pos_dev;
    if (mfma_decode_applies(NH, NKV, HS, band_max) && !p.tickets && !p.no_combine && !p.warm_a && !p.warm_b)
    {
        // 16 heads on one KV head over a long band: the matrix-core decode (attn_decode_mfma_kernel).  The fused form is the CHAIN here -- q/k/v norm + RoPE + KV append
        // as its own launch, the roped q rows parked behind the partials -- the 4 us it costs are nothing against a band of thousands of keys
        p.splits = mfma_decode_splits(B, NH, NKV, band_max);
        const size_t part_floats = (size_t)B * NH * p.splits * (HS + 4);
        const size_t need = part_floats * sizeof(float) + (fused ? (size_t)B * NH * HS * 2 : 0);
        if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "%s: scratch %zu bytes < required %zu", who, scratch_bytes, need);
        if (fused)
        {
            uint16_t* q_tmp = reinterpret_cast<uint16_t*>(p.scratch + part_floats);
            for (int b = 0; b < B; ++b)
            {
                uint16_t* kc = Kc + (size_t)b * NKV * capacity * HS;
                uint16_t* vc = Vc + (size_t)b * NKV * capacity * HS;
                const size_t ro = (size_t)b * p.raw_b_stride;
                const int rc = pos_dev ? mila_cdna4_fused_qkv_post_devpos(q_tmp + (size_t)b * NH * HS, kc, vc, p.q_raw + ro, p.k_raw + ro, p.v_raw + ro, p.qw, p.kw, p.vw, p.cos_cache,
                                                                          p.sin_cache, NH, NKV, HS, pos_dev, capacity, p.eps, reinterpret_cast<mila_stream_t>(stream))
                                       : mila_cdna4_fused_qkv_post(q_tmp + (size_t)b * NH * HS, kc, vc, p.q_raw + ro, p.k_raw + ro, p.v_raw + ro, p.qw, p.kw, p.vw, p.cos_cache,
                                                                   p.sin_cache, NH, NKV, HS, position, capacity, p.eps, reinterpret_cast<mila_stream_t>(stream));
                if (rc) return rc;
            }
            p.Q = q_tmp;
            p.q_b_stride = 0;
        }
        return launch_decode_mfma(p, B, stream);
    }
    const bool split_kernel = HS == 64 || HS == 128 || HS == 256 || HS == 512;
    if (split_kernel && p.splits > 1)
    {
        const size_t need = (size_t)B * NH * p.splits * (HS + 4) * sizeof(float);
        if (!scratch || scratch_bytes < need)
            return set_error(MILA_E_SCRATCH_TOO_SMALL, "%s: scratch %zu bytes < required %zu", who, scratch_bytes, need);
    }
    {
        // the XCD-local grid needs the head groups to tile the 8 XCDs (8 groups, or a multiple), and none of the experiment hooks that address the 3-D grid
        const int GS_ = NH / NKV, ngrp = NKV * (GS_ / heads_per_group(GS_, HS));
        p.flat = (g_attn_xcd_local && split_kernel && ngrp % 8 == 0 && !p.tickets && !p.no_combine && !p.warm_a && !p.warm_b) ? 1 : 0;
    }
    return fused ? dispatch_hs<true>(HS, p, B, stream) : dispatch_hs<false>(HS, p, B, stream);
}

// unfused decode whose query rows are `q_b_stride` elements apart (a packed [B, 1, 3C] MHA projection)
static int run_decode_q(uint16_t* Y, const uint16_t* Q, int64_t q_b_stride, uint16_t* Kc, uint16_t* Vc, void* scratch, size_t scratch_bytes, int B,
                        int NH, int NKV, int HS, int capacity, int position, int window, float scale, const char* who, hipStream_t stream)
{
    AttnParams p{};
    p.Y = Y; p.Q = Q; p.q_b_stride = q_b_stride; p.K = Kc; p.V = Vc; p.scratch = reinterpret_cast<float*>(scratch);
    p.NH = NH; p.NKV = NKV; p.capacity = capacity; p.position = position; p.window = window;
    const int band_max = (window > 0 && window < capacity) ? window : capacity;
    p.splits = decode_splits(B, NH, NKV, HS, band_max);
    p.scale = scale;
    const bool split_kernel = HS == 64 || HS == 128 || HS == 256 || HS == 512;
    if (split_kernel && p.splits > 1)
    {
        const size_t need = (size_t)B * NH * p.splits * (HS + 4) * sizeof(float);
        if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "%s: scratch %zu bytes < required %zu", who, scratch_bytes, need);
    }
    return dispatch_hs<false>(HS, p, B, stream);
}

// packed [B, T, 3C] rows -> K / V rows of a [B, NH, capacity, HS] cache at positions start_pos .. start_pos + T - 1 (the K / V part of the reference's
// permute_qkv / permute_qkv_decode, CudaMhaOp.ixx:166-170, :276-280)
__global__ __launch_bounds__(256) void mha_kv_write_kernel(uint16_t* __restrict__ Kc, uint16_t* __restrict__ Vc, const uint16_t* __restrict__ QKV,
                                                           int64_t total, int T, int C, int HS, int start_pos, int capacity)
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
        const uint16_t* row = QKV + bt * 3 * (int64_t)C;
        Kc[dst] = row[C + c];
        Vc[dst] = row[2 * C + c];
    }
}
// the same with 16-byte accesses (HS % 8 == 0: a head row is whole 8-element vectors; C % 8 == 0 follows)
__global__ __launch_bounds__(256) void mha_kv_write_vec_kernel(uint16_t* __restrict__ Kc, uint16_t* __restrict__ Vc, const uint16_t* __restrict__ QKV,
                                                               int64_t total_vec, int T, int C, int HS, int start_pos, int capacity)
{
    const int NH = C / HS, cv = C / 8;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += stride)
    {
        const int c = (int)(i % cv) * 8;
        const int64_t bt = i / cv;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int h = c / HS, d = c - h * HS;
        const size_t dst = (((size_t)b * NH + h) * capacity + (start_pos + t)) * HS + d;
        const uint16_t* row = QKV + bt * 3 * (int64_t)C;
        st16(Kc + dst, ld16(row + C + c));
        st16(Vc + dst, ld16(row + 2 * C + c));
    }
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_kv_write_bf16(uint16_t* Kc, uint16_t* Vc, const uint16_t* k, const uint16_t* v, int B, int chunk, int NKV,
                             int HS, int start_pos, int capacity, mila_stream_t stream)
{
    MILA_REQUIRE(Kc && Vc && k && v, "kv_write_bf16: null pointer");
    MILA_REQUIRE(B > 0 && chunk > 0 && NKV > 0 && HS > 0 && capacity > 0, "kv_write_bf16: bad sizes");
    MILA_REQUIRE(HS % 8 == 0, "kv_write_bf16: HS=%d must be a multiple of 8", HS);
    MILA_REQUIRE(start_pos >= 0, "kv_write_bf16: negative start position");
    MILA_REQUIRE(chunk <= capacity, "kv_write_bf16: chunk %d exceeds the cache capacity %d", chunk, capacity);
    const int64_t total_vec = (int64_t)B * chunk * NKV * (HS / 8);
    int blocks = ceil_div(total_vec, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(kv_write_bf16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), Kc, Vc, k, v, total_vec, chunk,
                       NKV, HS, start_pos, capacity);
    MILA_LAUNCH_CHECK("kv_write_bf16");
}

size_t mila_cdna4_attn_decode_scratch_bytes(int B, int NH, int HS)
{
    if (B <= 0 || NH <= 0 || HS <= 0) return 0;
    const size_t scalar = (size_t)B * NH * kMaxSplits * (HS + 4) * sizeof(float);
    // the long-context MFMA decode (HS 512): up to kMaxSplitsMfma partials per head + the roped q rows of its fused form
    const size_t mfma = (HS == 512) ? (size_t)B * NH * kMaxSplitsMfma * (HS + 4) * sizeof(float) + (size_t)B * NH * HS * 2 : 0;
    return scalar > mfma ? scalar : mfma;
}

int mila_cdna4_attn_decode_bf16(uint16_t* Y, const uint16_t* Q, const uint16_t* Kc, const uint16_t* Vc, void* scratch,
                                size_t scratch_bytes, int B, int NH, int NKV, int HS, int capacity, int len, int window,
                                float scale, mila_stream_t stream)
{
    MILA_REQUIRE(Y && Q && Kc && Vc, "attn_decode_bf16: null pointer");
    MILA_REQUIRE(B > 0 && NH > 0 && NKV > 0 && NH % NKV == 0, "attn_decode_bf16: bad head counts (NH=%d NKV=%d)", NH, NKV);
    MILA_REQUIRE(len > 0 && capacity > 0, "attn_decode_bf16: len and capacity must be positive (len=%d capacity=%d)", len, capacity);
    MILA_REQUIRE(window >= 0, "attn_decode_bf16: negative window");
    const int band = (window > 0 && window < len) ? window : len;
    MILA_REQUIRE(band <= capacity, "attn_decode_bf16: live band %d exceeds the cache capacity %d", band, capacity);
    return run_decode(Y, Q, const_cast<uint16_t*>(Kc), const_cast<uint16_t*>(Vc), scratch, scratch_bytes, B, NH, NKV, HS, capacity,
                      len - 1, nullptr, window, scale, nullptr, "attn_decode_bf16", as_stream(stream));
}

int mila_cdna4_attn_decode_bf16_devpos(uint16_t* Y, const uint16_t* Q, const uint16_t* Kc, const uint16_t* Vc, void* scratch,
                                       size_t scratch_bytes, int B, int NH, int NKV, int HS, int capacity,
                                       const int32_t* position_dev, int max_len, int window, float scale,
                                       mila_stream_t stream)
{
    MILA_REQUIRE(Y && Q && Kc && Vc && position_dev, "attn_decode_bf16_devpos: null pointer");
    MILA_REQUIRE(B > 0 && NH > 0 && NKV > 0 && NH % NKV == 0, "attn_decode_bf16_devpos: bad head counts (NH=%d NKV=%d)", NH, NKV);
    MILA_REQUIRE(max_len > 0 && capacity > 0 && window >= 0, "attn_decode_bf16_devpos: bad sizes");
    const int band = (window > 0 && window < max_len) ? window : max_len;
    MILA_REQUIRE(band <= capacity, "attn_decode_bf16_devpos: live band %d exceeds the cache capacity %d", band, capacity);
    return run_decode(Y, Q, const_cast<uint16_t*>(Kc), const_cast<uint16_t*>(Vc), scratch, scratch_bytes, B, NH, NKV, HS, capacity, 0,
                      position_dev, window, scale, nullptr, "attn_decode_bf16_devpos", as_stream(stream), max_len);
}

// Fused decode attention for one token (B == 1): per-head q/k/v RMSNorm + RoPE + KV append + flash-decode.
// position_dev != NULL selects the graph-replay form (position read from device memory).
int mila_cdna4_fused_attn_decode_bf16(uint16_t* Y, uint16_t* Kc, uint16_t* Vc, const uint16_t* q_raw, const uint16_t* k_raw,
                                      const uint16_t* v_raw, const uint16_t* qw, const uint16_t* kw, const uint16_t* vw,
                                      const float* cos_cache, const float* sin_cache, void* scratch, size_t scratch_bytes,
                                      int NH, int NKV, int HS, int capacity, int position, const int32_t* position_dev,
                                      int window, float scale, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(Y && Kc && Vc && q_raw && k_raw && v_raw && qw && kw && cos_cache && sin_cache, "fused_attn_decode_bf16: null pointer");
    MILA_REQUIRE(NH > 0 && NKV > 0 && NH % NKV == 0, "fused_attn_decode_bf16: bad head counts (NH=%d NKV=%d)", NH, NKV);
    MILA_REQUIRE(capacity > 0 && window >= 0 && (position_dev || position >= 0), "fused_attn_decode_bf16: bad sizes");
    MILA_REQUIRE(HS % 16 == 0, "fused_attn_decode_bf16: HS=%d must be a multiple of 16", HS);
    if (!position_dev)
    {
        const int len = position + 1, band = (window > 0 && window < len) ? window : len;
        MILA_REQUIRE(band <= capacity, "fused_attn_decode_bf16: live band %d exceeds the cache capacity %d", band, capacity);
    }
    AttnParams f{};
    f.q_raw = q_raw; f.k_raw = k_raw; f.v_raw = v_raw; f.qw = qw; f.kw = kw; f.vw = vw; f.cos_cache = cos_cache; f.sin_cache = sin_cache;
    f.eps = eps;
    return run_decode(Y, nullptr, Kc, Vc, scratch, scratch_bytes, 1, NH, NKV, HS, capacity, position, position_dev, window, scale, &f,
                      "fused_attn_decode_bf16", as_stream(stream), position_dev ? position : 0);
}

// The same launch for B rows decoded at one position (Gqa.Decode.Bf16.cu:379-387 takes the batch in its grid): batch row b reads its raw projections at
// q_raw / k_raw / v_raw + b * raw_b_stride (the three pointers into one packed [B, 1, q | k | v] row, or separate dense tensors) and its own caches; Y [B, NH*HS].
int mila_cdna4_fused_attn_decode_batch_bf16(uint16_t* Y, uint16_t* Kc, uint16_t* Vc, const uint16_t* q_raw, const uint16_t* k_raw,
                                            const uint16_t* v_raw, int64_t raw_b_stride, const uint16_t* qw, const uint16_t* kw, const uint16_t* vw,
                                            const float* cos_cache, const float* sin_cache, void* scratch, size_t scratch_bytes, int B,
                                            int NH, int NKV, int HS, int capacity, int position, const int32_t* position_dev,
                                            int window, float scale, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(Y && Kc && Vc && q_raw && k_raw && v_raw && qw && kw && cos_cache && sin_cache, "fused_attn_decode_batch_bf16: null pointer");
    MILA_REQUIRE(B > 0 && NH > 0 && NKV > 0 && NH % NKV == 0, "fused_attn_decode_batch_bf16: bad sizes (B=%d NH=%d NKV=%d)", B, NH, NKV);
    MILA_REQUIRE(B == 1 || raw_b_stride >= (int64_t)NKV * HS, "fused_attn_decode_batch_bf16: raw_b_stride %lld is shorter than a K row", (long long)raw_b_stride);
    MILA_REQUIRE(capacity > 0 && window >= 0 && (position_dev || position >= 0), "fused_attn_decode_batch_bf16: bad sizes");
    MILA_REQUIRE(HS % 16 == 0, "fused_attn_decode_batch_bf16: HS=%d must be a multiple of 16", HS);
    if (!position_dev)
    {
        const int len = position + 1, band = (window > 0 && window < len) ? window : len;
        MILA_REQUIRE(band <= capacity, "fused_attn_decode_batch_bf16: live band %d exceeds the cache capacity %d", band, capacity);
    }
    AttnParams f{};
    f.q_raw = q_raw; f.k_raw = k_raw; f.v_raw = v_raw; f.raw_b_stride = raw_b_stride; f.qw = qw; f.kw = kw; f.vw = vw; f.cos_cache = cos_cache; f.sin_cache = sin_cache;
    f.eps = eps;
    return run_decode(Y, nullptr, Kc, Vc, scratch, scratch_bytes, B, NH, NKV, HS, capacity, position, position_dev, window, scale, &f,
                      "fused_attn_decode_batch_bf16", as_stream(stream), position_dev ? position : 0);
}

// ---- GPT-2 multi-head attention over a KV cache (CudaMhaOp.ixx:137-380: prefill / decode of IPositionalUnaryOp + IKvCacheLifecycle) ----
// The cache is [B, NH, capacity, HS] like the GQA one (NKV = NH); K / V come from the packed [B, T, 3C] projection.
int mila_cdna4_mha_kv_write_bf16(uint16_t* Kc, uint16_t* Vc, const uint16_t* QKV, int B, int T, int C, int NH, int start_pos, int capacity,
                                 mila_stream_t stream)
{
    MILA_REQUIRE(Kc && Vc && QKV, "mha_kv_write_bf16: null pointer");
    MILA_REQUIRE(B > 0 && T > 0 && C > 0 && NH > 0 && C % NH == 0 && capacity > 0, "mha_kv_write_bf16: bad sizes (B=%d T=%d C=%d NH=%d capacity=%d)", B, T, C, NH, capacity);
    MILA_REQUIRE(start_pos >= 0 && start_pos + T <= capacity, "mha_kv_write_bf16: positions [%d, %d) do not fit the cache capacity %d", start_pos, start_pos + T, capacity);
    const int HS = C / NH;
    const bool vec = HS % 8 == 0 && ((uintptr_t)QKV % 16 == 0) && ((uintptr_t)Kc % 16 == 0) && ((uintptr_t)Vc % 16 == 0);
    const int64_t total = vec ? (int64_t)B * T * (C / 8) : (int64_t)B * T * C;
    int blocks = ceil_div(total, 256);
    if (blocks > 4096) blocks = 4096;
    if (vec) hipLaunchKernelGGL(mha_kv_write_vec_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), Kc, Vc, QKV, total, T, C, HS, start_pos, capacity);
    else hipLaunchKernelGGL(mha_kv_write_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), Kc, Vc, QKV, total, T, C, HS, start_pos, capacity);
    MILA_LAUNCH_CHECK("mha_kv_write_bf16");
}

size_t mila_cdna4_mha_decode_scratch_bytes(int B, int C, int NH)
{
    if (B <= 0 || C <= 0 || NH <= 0 || C % NH != 0) return 0;
    return mila_cdna4_attn_decode_scratch_bytes(B, NH, C / NH);
}

int mila_cdna4_mha_decode_bf16(uint16_t* Y, const uint16_t* QKV, uint16_t* Kc, uint16_t* Vc, void* scratch, size_t scratch_bytes, int B, int C,
                               int NH, int capacity, int position, mila_stream_t stream)
{
    MILA_REQUIRE(Y && QKV && Kc && Vc, "mha_decode_bf16: null pointer");
    MILA_REQUIRE(B > 0 && C > 0 && NH > 0 && C % NH == 0 && capacity > 0, "mha_decode_bf16: bad sizes (B=%d C=%d NH=%d capacity=%d)", B, C, NH, capacity);
    MILA_REQUIRE(position >= 0 && position < capacity, "mha_decode_bf16: position %d out of range [0, %d)", position, capacity);     // CudaMhaOp.ixx:262-265
    const int HS = C / NH;
    int rc = mila_cdna4_mha_kv_write_bf16(Kc, Vc, QKV, B, 1, C, NH, position, capacity, stream);
    if (rc) return rc;
    return run_decode_q(Y, QKV, 3 * (int64_t)C, Kc, Vc, scratch, scratch_bytes, B, NH, NH, HS, capacity, position, 0, 1.0f / sqrtf((float)HS),
                        "mha_decode_bf16", as_stream(stream));
}

int mila_cdna4_attn_decode_split_count(int B, int NH, int NKV, int HS, int capacity, int window)
{
    if (B <= 0 || NH <= 0 || NKV <= 0 || NH % NKV != 0 || capacity <= 0 || window < 0) return 0;
    const int band_max = (window > 0 && window < capacity) ? window : capacity;
    return decode_splits(B, NH, NKV, HS, band_max);
}

int mila_cdna4_fused_attn_decode_partials_bf16(uint16_t* Kc, uint16_t* Vc, const uint16_t* q_raw, const uint16_t* k_raw,
                                               const uint16_t* v_raw, const uint16_t* qw, const uint16_t* kw, const uint16_t* vw,
                                               const float* cos_cache, const float* sin_cache, void* scratch, size_t scratch_bytes,
                                               int NH, int NKV, int HS, int capacity, int position, const int32_t* position_dev,
                                               int window, float scale, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(Kc && Vc && q_raw && k_raw && v_raw && qw && kw && cos_cache && sin_cache, "fused_attn_decode_partials_bf16: null pointer");
    MILA_REQUIRE(NH > 0 && NKV > 0 && NH % NKV == 0, "fused_attn_decode_partials_bf16: bad head counts (NH=%d NKV=%d)", NH, NKV);
    MILA_REQUIRE(capacity > 0 && window >= 0 && (position_dev || position >= 0), "fused_attn_decode_partials_bf16: bad sizes");
    MILA_REQUIRE(HS % 16 == 0, "fused_attn_decode_partials_bf16: HS=%d must be a multiple of 16", HS);
    MILA_REQUIRE(mila_cdna4_attn_decode_split_count(1, NH, NKV, HS, capacity, window) > 1,
                 "fused_attn_decode_partials_bf16: this (window, capacity) runs unsplit; use fused_attn_decode_bf16");
    if (!position_dev)
    {
        const int len = position + 1, band = (window > 0 && window < len) ? window : len;
        MILA_REQUIRE(band <= capacity, "fused_attn_decode_partials_bf16: live band %d exceeds the cache capacity %d", band, capacity);
    }
    AttnParams f{};
    f.q_raw = q_raw; f.k_raw = k_raw; f.v_raw = v_raw; f.qw = qw; f.kw = kw; f.vw = vw; f.cos_cache = cos_cache; f.sin_cache = sin_cache;
    f.eps = eps;
    f.no_combine = 1;
    return run_decode(nullptr, nullptr, Kc, Vc, scratch, scratch_bytes, 1, NH, NKV, HS, capacity, position, position_dev, window, scale, &f,
                      "fused_attn_decode_partials_bf16", as_stream(stream));
}

size_t mila_cdna4_attn_decode_ticket_count(int B, int NH)
{
    if (B <= 0 || NH <= 0) return 0;
    return (size_t)B * NH;     // one counter per (batch row, head-group); head-groups <= NH
}

// fused_attn_decode_bf16 without the combine launch: the workgroup whose split partials arrive last (an arrival ticket per
// head-group in `tickets`: uint32 [ticket_count], zero before the first call, re-armed by every call) merges its head-group's
// splits in the same launch, with the combine kernel's arithmetic => bit-identical to fused_attn_decode_bf16.
int mila_cdna4_fused_attn_decode_onepass_bf16(uint16_t* Y, uint16_t* Kc, uint16_t* Vc, const uint16_t* q_raw, const uint16_t* k_raw,
                                              const uint16_t* v_raw, const uint16_t* qw, const uint16_t* kw, const uint16_t* vw,
                                              const float* cos_cache, const float* sin_cache, void* scratch, size_t scratch_bytes,
                                              uint32_t* tickets, size_t ticket_count, int NH, int NKV, int HS, int capacity, int position,
                                              const int32_t* position_dev, int window, float scale, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(Y && Kc && Vc && q_raw && k_raw && v_raw && qw && kw && cos_cache && sin_cache && tickets, "fused_attn_decode_onepass_bf16: null pointer");
    MILA_REQUIRE(NH > 0 && NKV > 0 && NH % NKV == 0, "fused_attn_decode_onepass_bf16: bad head counts (NH=%d NKV=%d)", NH, NKV);
    MILA_REQUIRE(capacity > 0 && window >= 0 && (position_dev || position >= 0), "fused_attn_decode_onepass_bf16: bad sizes");
    MILA_REQUIRE(HS % 16 == 0, "fused_attn_decode_onepass_bf16: HS=%d must be a multiple of 16", HS);
    MILA_REQUIRE(ticket_count >= mila_cdna4_attn_decode_ticket_count(1, NH), "fused_attn_decode_onepass_bf16: %zu tickets < required %zu",
                 ticket_count, mila_cdna4_attn_decode_ticket_count(1, NH));
    if (!position_dev)
    {
        const int len = position + 1, band = (window > 0 && window < len) ? window : len;
        MILA_REQUIRE(band <= capacity, "fused_attn_decode_onepass_bf16: live band %d exceeds the cache capacity %d", band, capacity);
    }
    AttnParams f{};
    f.q_raw = q_raw; f.k_raw = k_raw; f.v_raw = v_raw; f.qw = qw; f.kw = kw; f.vw = vw; f.cos_cache = cos_cache; f.sin_cache = sin_cache;
    f.eps = eps;
    f.tickets = tickets;
    return run_decode(Y, nullptr, Kc, Vc, scratch, scratch_bytes, 1, NH, NKV, HS, capacity, position, position_dev, window, scale, &f,
                      "fused_attn_decode_onepass_bf16", as_stream(stream));
}

// fused_attn_decode_bf16 (or, with tickets, fused_attn_decode_onepass_bf16) from an argument block, with optional warm ranges
int mila_cdna4_fused_attn_decode_ex(const mila_fused_attn_args* a, mila_stream_t stream)
{
    MILA_REQUIRE(a != nullptr, "fused_attn_decode_ex: null arguments");
    MILA_REQUIRE(a->Y && a->Kc && a->Vc && a->q_raw && a->k_raw && a->v_raw && a->qw && a->kw && a->cos_cache && a->sin_cache, "fused_attn_decode_ex: null pointer");
    MILA_REQUIRE(a->NH > 0 && a->NKV > 0 && a->NH % a->NKV == 0, "fused_attn_decode_ex: bad head counts (NH=%d NKV=%d)", a->NH, a->NKV);
    MILA_REQUIRE(a->capacity > 0 && a->window >= 0 && (a->position_dev || a->position >= 0), "fused_attn_decode_ex: bad sizes");
    MILA_REQUIRE(a->HS % 16 == 0, "fused_attn_decode_ex: HS=%d must be a multiple of 16", a->HS);
    MILA_REQUIRE(a->warm_a_blocks >= 0 && a->warm_a_blocks <= 64 && a->warm_b_blocks >= 0 && a->warm_b_blocks <= 64, "fused_attn_decode_ex: warm block counts must be in [0, 64]");
    if (a->tickets)
        MILA_REQUIRE(a->ticket_count >= mila_cdna4_attn_decode_ticket_count(1, a->NH), "fused_attn_decode_ex: %zu tickets < required %zu",
                     a->ticket_count, mila_cdna4_attn_decode_ticket_count(1, a->NH));
    if (!a->position_dev)
    {
        const int len = a->position + 1, band = (a->window > 0 && a->window < len) ? a->window : len;
        MILA_REQUIRE(band <= a->capacity, "fused_attn_decode_ex: live band %d exceeds the cache capacity %d", band, a->capacity);
    }
    AttnParams f{};
    f.q_raw = a->q_raw; f.k_raw = a->k_raw; f.v_raw = a->v_raw; f.qw = a->qw; f.kw = a->kw; f.vw = a->vw; f.cos_cache = a->cos_cache; f.sin_cache = a->sin_cache;
    f.eps = a->eps;
    f.tickets = a->tickets;
    if (a->warm_a && a->warm_a_bytes >= 128 && a->warm_a_blocks > 0) { f.warm_a = (const uint8_t*)a->warm_a; f.warm_a_lines = (int64_t)(a->warm_a_bytes / 128); f.warm_a_blocks = a->warm_a_blocks; }
    if (a->warm_b && a->warm_b_bytes >= 128 && a->warm_b_blocks > 0) { f.warm_b = (const uint8_t*)a->warm_b; f.warm_b_lines = (int64_t)(a->warm_b_bytes / 128); f.warm_b_blocks = a->warm_b_blocks; f.warm_b_pair = (int64_t)a->warm_b_pair_offset; }
    return run_decode(a->Y, nullptr, a->Kc, a->Vc, a->scratch, a->scratch_bytes, 1, a->NH, a->NKV, a->HS, a->capacity, a->position, a->position_dev, a->window,
                      a->scale, &f, "fused_attn_decode_ex", as_stream(stream));
}

}  // extern "C"
