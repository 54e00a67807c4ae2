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

namespace mila {

constexpr int kMaxSplits = 64;

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

// ---- generic addressing so one kernel serves the cache layout and GPT-2's packed QKV -----------------
struct AttnParams
{
    uint16_t* Y;              // [B*Tq, NH*HS]
    const uint16_t* Q;        // row (b*Tq+t): Q + (b*Tq+t)*q_row_stride + h*HS
    const uint16_t* K;        // K + b*kv_b_stride + kvh*kv_h_stride + row*kv_r_stride
    const uint16_t* V;
    float* scratch;           // [B*Tq, NH, splits, HS+2] partials when splits > 1
    int64_t q_row_stride, kv_b_stride, kv_h_stride, kv_r_stride;
    int Tq, NH, NKV, capacity, pos_offset, window, splits;
    float scale;
    const int32_t* pos_dev;   // when set: position of query 0 is read from device memory (graph replay)
};

template <int EPL> struct RowVec;                        // EPL bf16 elements per lane
template <> struct RowVec<8> { typedef u32x4 type; };
template <> struct RowVec<4> { typedef u32x2 type; };
template <> struct RowVec<2> { typedef uint32_t type; };

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

// HS = 64 * EPL (EPL in {2,4,8}) or HS = 64 handled as EPL = 2 on 32 active lanes.
// GH = query heads handled per workgroup (<= 4): grid.y = NKV * (GS / GH); the head groups of one
// KV head re-read the same K/V rows from L2, which costs nothing next to the 4x cut in per-wave
// VALU work and registers for MQA (GS = 16).
template <int HS, int GH>
__global__ __launch_bounds__(256) void attn_rowwise_kernel(const AttnParams p)
{
    constexpr int EPL = (HS >= 128) ? HS / 64 : 2;
    constexpr int NPAIR = EPL / 2;
    constexpr int ACTIVE = HS / EPL;                       // lanes that own data (64, or 32 for HS = 64)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);        // [4 waves][GH][HS + 2]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool owner = lane < ACTIVE;
    const int GS = p.NH / p.NKV, hgroups = GS / GH;
    const int split = blockIdx.x, kvh = blockIdx.y / hgroups, hg = blockIdx.y % hgroups;
    const int h0 = kvh * GS + hg * GH;                     // first query head of this workgroup
    const int bt = blockIdx.z, b = bt / p.Tq, t = bt % p.Tq;
    const int pos = (p.pos_dev ? *p.pos_dev : p.pos_offset) + t;
    const int len = pos + 1;
    const int band_begin = (p.window > 0) ? max(0, len - p.window) : 0;
    const int band = len - band_begin;
    const int chunk = (band + p.splits - 1) / p.splits;
    const int begin = band_begin + split * chunk;
    const int end = min(begin + chunk, len);

    // q for the GH heads, packed bf16 pairs
    uint32_t q[GH][NPAIR];
#pragma unroll
    for (int g = 0; g < GH; ++g)
    {
        const uint16_t* qp = p.Q + (size_t)bt * p.q_row_stride + (size_t)(h0 + g) * HS + lane * EPL;
        if (owner) load_row<EPL>(q[g], qp);
        else
        {
#pragma unroll
            for (int e = 0; e < NPAIR; ++e) q[g][e] = 0u;
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

    const uint16_t* kbase = p.K + (size_t)b * p.kv_b_stride + (size_t)kvh * p.kv_h_stride + lane * EPL;
    const uint16_t* vbase = p.V + (size_t)b * p.kv_b_stride + (size_t)kvh * p.kv_h_stride + lane * EPL;

    // A wave owns positions base, base + 4, ... of the split, PG at a time.  All K/V rows of a group are
    // requested at once and the next group's rows are in flight while the current group is reduced, so a
    // split of <= 4*PG positions per wave costs a single HBM round trip.
    constexpr int PG = (HS >= 512) ? 4 : 8;
    struct KVG { uint32_t k[PG][NPAIR], v[PG][NPAIR]; };
    auto load_group = [&](KVG& gbuf, int base) {
#pragma unroll
        for (int j = 0; j < PG; ++j)
        {
            const int pos = base + 4 * j;
            if (owner && pos < end)
            {
                const size_t r = (size_t)(pos % p.capacity) * p.kv_r_stride;
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
    auto compute_group = [&](const KVG& gbuf, int base) {
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
                a[j] = (base + 4 * j < end) ? sc[j][g] * p.scale : -INFINITY;
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
    {
        int base = begin + wave;
        KVG ga, gb;
        if (base < end) load_group(ga, base);
        for (;;)
        {
            if (base >= end) break;
            int nb = base + 4 * PG;
            if (nb < end) load_group(gb, nb);
            compute_group(ga, base);
            base = nb;
            if (base >= end) break;
            nb = base + 4 * PG;
            if (nb < end) load_group(ga, nb);
            compute_group(gb, base);
            base = nb;
        }
    }

    // ---- merge the 4 waves through LDS; wave w finalises head w (GH <= 4) ----
    constexpr int STR = HS + 2;
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
        for (int w = 0; w < 4; ++w) M = fmaxf(M, sm[((size_t)w * GH + g) * STR + HS]);
        float L = 0.0f, acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
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
            uint16_t* y = p.Y + ((size_t)bt * p.NH + h) * HS + lane * EPL;
            if (owner)
            {
#pragma unroll
                for (int e = 0; e < EPL; e += 2)
                    *reinterpret_cast<uint32_t*>(y + e) = pack_bf16x2(acc[e] * inv, acc[e + 1] * inv);
            }
        }
        else
        {
            float* dst = p.scratch + (((size_t)bt * p.NH + h) * p.splits + split) * STR;
            if (owner)
            {
#pragma unroll
                for (int e = 0; e < EPL; ++e) dst[lane * EPL + e] = acc[e];
            }
            if (lane == 0) { dst[HS] = M; dst[HS + 1] = L; }
        }
    }
}

// combine split partials: grid (NH, B*Tq, HS/64), 64 threads -> 64 dims each
__global__ __launch_bounds__(64) void attn_combine_kernel(uint16_t* __restrict__ Y, const float* __restrict__ scratch, int NH,
                                                          int HS, int splits)
{
    const int h = blockIdx.x, bt = blockIdx.y;
    const int d = blockIdx.z * 64 + threadIdx.x;
    const int STR = HS + 2;
    const float* base = scratch + ((size_t)bt * NH + h) * splits * STR;
    // lane s holds split s's (m, l); splits <= 64
    const int s = threadIdx.x;
    const float ms = (s < splits) ? base[(size_t)s * STR + HS] : -INFINITY;
    const float ls = (s < splits) ? base[(size_t)s * STR + HS + 1] : 0.0f;
    const float M = wave_max(ms);
    const float fs = (ms == -INFINITY) ? 0.0f : __expf(ms - M);
    const float L = wave_sum(ls * fs);
    float acc = 0.0f;
#pragma unroll 8
    for (int i = 0; i < splits; ++i)
    {
        const float f = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fs), i));
        acc = fmaf(base[(size_t)i * STR + d], f, acc);
    }
    Y[((size_t)bt * NH + h) * HS + d] = f32_to_bf16_bits(L > 0.0f ? acc / L : 0.0f);
}

template <int HS, int GH>
static int launch_rowwise(const AttnParams& p, int B, hipStream_t s)
{
    const int hgroups = (p.NH / p.NKV) / GH;
    const size_t lds = (size_t)4 * GH * (HS + 2) * sizeof(float);
    hipLaunchKernelGGL((attn_rowwise_kernel<HS, GH>), dim3(p.splits, p.NKV * hgroups, B * p.Tq), dim3(256), lds, s, p);
    int rc = check_hip(hipGetLastError(), "attn_rowwise");
    if (rc) return rc;
    if (p.splits > 1)
    {
        hipLaunchKernelGGL(attn_combine_kernel, dim3(p.NH, B * p.Tq, HS / 64), dim3(64), 0, s, p.Y, p.scratch, p.NH, HS,
                           p.splits);
        rc = check_hip(hipGetLastError(), "attn_combine");
    }
    return rc;
}

static int heads_per_group(int GS) { return GS >= 4 ? 4 : GS; }

template <int HS>
static int dispatch_gs(const AttnParams& p, int B, hipStream_t s)
{
    const int GS = p.NH / p.NKV;
    switch (GS)
    {
        case 1: return launch_rowwise<HS, 1>(p, B, s);
        case 2: return launch_rowwise<HS, 2>(p, B, s);
        case 4: case 8: case 16: case 32: return launch_rowwise<HS, 4>(p, B, s);
        default: return set_error(MILA_E_UNSUPPORTED, "attention: group size %d (NH/NKV) must be 1,2,4,8,16 or 32", GS);
    }
}

static int dispatch_hs(int HS, const AttnParams& p, int B, hipStream_t s)
{
    switch (HS)
    {
        case 64: return dispatch_gs<64>(p, B, s);
        case 128: return dispatch_gs<128>(p, B, s);
        case 256: return dispatch_gs<256>(p, B, s);
        case 512: return dispatch_gs<512>(p, B, s);
        default: return set_error(MILA_E_UNSUPPORTED, "attention: head size %d must be 64, 128, 256 or 512", HS);
    }
}

static int decode_splits(int B, int NH, int NKV, int band)
{
    // ~256 workgroups (one per CU) of 32 positions (8 per wave: one or two load groups)
    const int hgroups = (NH / NKV) / heads_per_group(NH / NKV);
    int cap = 256 / (NKV * hgroups * B);
    if (cap < 1) cap = 1;
    int s = (band + 31) / 32;
    if (s > cap) s = cap;
    if (s > kMaxSplits) s = kMaxSplits;
    if (s < 1) s = 1;
    return s;
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
    return (size_t)B * NH * kMaxSplits * (HS + 2) * sizeof(float);
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
    // the split count depends only on (window, capacity), never on the current length, so that eager
    // launches and a graph captured once (the _devpos form) reduce in the same order: bit-identical
    const int band_max = (window > 0 && window < capacity) ? window : capacity;
    AttnParams p;
    p.Y = Y; p.Q = Q; p.K = Kc; p.V = Vc; p.scratch = reinterpret_cast<float*>(scratch);
    p.q_row_stride = (int64_t)NH * HS;
    p.kv_b_stride = (int64_t)NKV * capacity * HS;
    p.kv_h_stride = (int64_t)capacity * HS;
    p.kv_r_stride = HS;
    p.Tq = 1; p.NH = NH; p.NKV = NKV; p.capacity = capacity; p.pos_offset = len - 1; p.window = window;
    p.splits = decode_splits(B, NH, NKV, band_max);
    p.scale = scale;
    p.pos_dev = nullptr;
    if (p.splits > 1)
    {
        const size_t need = (size_t)B * NH * p.splits * (HS + 2) * sizeof(float);
        if (!scratch || scratch_bytes < need)
            return set_error(MILA_E_SCRATCH_TOO_SMALL, "attn_decode_bf16: scratch %zu bytes < required %zu", scratch_bytes, need);
    }
    return dispatch_hs(HS, p, B, as_stream(stream));
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
    const int band_max = (window > 0 && window < capacity) ? window : capacity;   // same rule as the eager form
    AttnParams p;
    p.Y = Y; p.Q = Q; p.K = Kc; p.V = Vc; p.scratch = reinterpret_cast<float*>(scratch);
    p.q_row_stride = (int64_t)NH * HS;
    p.kv_b_stride = (int64_t)NKV * capacity * HS;
    p.kv_h_stride = (int64_t)capacity * HS;
    p.kv_r_stride = HS;
    p.Tq = 1; p.NH = NH; p.NKV = NKV; p.capacity = capacity; p.pos_offset = 0; p.window = window;
    p.splits = decode_splits(B, NH, NKV, band_max);
    p.scale = scale;
    p.pos_dev = position_dev;
    if (p.splits > 1)
    {
        const size_t need = (size_t)B * NH * p.splits * (HS + 2) * sizeof(float);
        if (!scratch || scratch_bytes < need)
            return set_error(MILA_E_SCRATCH_TOO_SMALL, "attn_decode_bf16_devpos: scratch %zu bytes < required %zu", scratch_bytes, need);
    }
    return dispatch_hs(HS, p, B, as_stream(stream));
}

}  // extern "C"
