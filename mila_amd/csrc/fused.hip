// Fused decode-step glue for one token:
//   q <- rope(rmsnorm(q; qw));  K[pos] <- rope(rmsnorm(k; kw));  V[pos] <- rmsnorm(v_src; vw | 1)
// replaces, per layer, six launches of the unfused chain in GemmaBlock::decode
// (Components/Transformers/Gemma/Gemma.Block.ixx:315-337: q_norm, k_norm, rope.decode(q,k), v_norm,
// kvcache_write) with one.  Every step calls the canonical helpers of rms_common.h / rope_common.h
// with the same lane->element assignment as the standalone kernels, so each intermediate bf16
// rounding of the unfused chain is reproduced and the results are bit-identical.
#include "common.h"
#include "rms_common.h"
#include "rope_common.h"

namespace mila {

struct QkvPostParams
{
    uint16_t* q_out;
    uint16_t* Kc;
    uint16_t* Vc;
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* v_src;
    const uint16_t* qw;
    const uint16_t* kw;
    const uint16_t* vw;
    const float* cos_cache;
    const float* sin_cache;
    const int32_t* pos_dev;   // when set, the position is read from device memory (graph replay)
    int NH, NKV, HS, position, capacity;
    float eps;
    // prefill form: token t = blockIdx.y at position + t reads its q/k/v at + t * src_row_stride and writes q_out at + t * NH * HS
    int64_t src_row_stride;
};

__global__ void advance_position_kernel(int32_t* pos) { *pos += 1; }
// The decode-ahead loop's host side learns each sampled token from a ring in host-visible (pinned, mapped) memory: entry = sequence number << 32 | token,
// stored with one system-scope release store; the host polls the slot until it carries the sequence number it expects.  No event, no copy, no stream wait.
__device__ __forceinline__ void publish_token(unsigned long long* seq_dev, unsigned long long* ring, int ring_size, const int32_t* token)
{
    const unsigned long long seq = *seq_dev + 1ull;
    *seq_dev = seq;
    __hip_atomic_store(ring + (seq % (unsigned long long)ring_size), (seq << 32) | (unsigned long long)(uint32_t)*token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void advance_position_snapshot_kernel(int32_t* pos, const int32_t* token, unsigned long long* seq_dev, unsigned long long* ring, int ring_size)
{
    *pos += 1;
    publish_token(seq_dev, ring, ring_size, token);
}
__global__ void snapshot_token_kernel(const int32_t* token, unsigned long long* seq_dev, unsigned long long* ring, int ring_size) { publish_token(seq_dev, ring, ring_size, token); }

// sum over the hv = HS / 16 lanes of a lane group (hv a power of two <= 32): the first log2(hv) steps of wave_sum's butterfly
__device__ __forceinline__ float group_tree_sum(float v, int hv)
{
    if (hv >= 2) v += dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]
    if (hv >= 4) v += dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]
    if (hv >= 8) v += dpp_f32<0x141>(v);    // row_half_mirror
    if (hv >= 16) v += dpp_f32<0x140>(v);   // row_mirror
    if (hv >= 32)
    {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}

// Head rows [0,NH) = q, [NH,NH+NKV) = k, [NH+NKV, NH+2NKV) = v of token blockIdx.y.  HS <= 512: a row is handled by the hv = HS / 16
// lanes that also apply the result (lane g holds the rotation pair of chunks g and g + hv), 64 / hv rows per wave, every lane busy.
// The sum of squares is tree(lo chunks) + tree(hi chunks) over those hv lanes -- exactly what rms_rstd_wave's 64-lane butterfly
// computes for a row of 2 hv chunks (its remaining steps add zeros), so the bits match the standalone RMSNorm kernel.
__global__ __launch_bounds__(256) void qkv_post_kernel(const QkvPostParams p)
{
    const int HS = p.HS, half = HS / 2, hv = half / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = (hv <= 32) ? 64 / hv : 1;                       // rows per wave
    const int grp = (hv <= 32) ? lane / hv : 0, gl = (hv <= 32) ? lane % hv : lane;
    const int r = (blockIdx.x * 4 + wave) * rpw + grp;
    const int nrows = p.NH + 2 * p.NKV;
    const bool valid = r < nrows;
    const int rr = valid ? r : nrows - 1;
    const int t = blockIdx.y;
    const int position = (p.pos_dev ? *p.pos_dev : p.position) + t;
    const int row = position % p.capacity;
    const size_t src_t = (size_t)t * p.src_row_stride;
    const float* cos_row = p.cos_cache + (size_t)position * half;
    const float* sin_row = p.sin_cache + (size_t)position * half;
    const uint16_t* src;
    const uint16_t* w;
    uint16_t* dst;
    bool rotate;
    if (rr < p.NH)
    {
        src = p.q + src_t + (size_t)rr * HS; w = p.qw; dst = p.q_out + ((size_t)t * p.NH + rr) * HS; rotate = true;
    }
    else if (rr < p.NH + p.NKV)
    {
        const int n = rr - p.NH;
        src = p.k + src_t + (size_t)n * HS; w = p.kw; dst = p.Kc + ((size_t)n * p.capacity + row) * HS; rotate = true;
    }
    else
    {
        const int n = rr - p.NH - p.NKV;
        src = p.v_src + src_t + (size_t)n * HS; w = p.vw; dst = p.Vc + ((size_t)n * p.capacity + row) * HS; rotate = false;
    }
    float rstd;
    u32x4 xlo, xhi;
    // the norm weights are requested with the row (they used to follow the reduction: a second round trip); rr is clamped, so the addresses are valid on
    // every lane
    u32x4 wlo = u32x4{0u, 0u, 0u, 0u}, whi = wlo;
    if (w) { wlo = ld16(w + (size_t)gl * 8); whi = ld16(w + (size_t)(gl + hv) * 8); }
    f32x4 c0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, c1 = c0, s0 = c0, s1 = c0;
    if (rotate)
    {
        c0 = *reinterpret_cast<const f32x4*>(cos_row + gl * 8); c1 = *reinterpret_cast<const f32x4*>(cos_row + gl * 8 + 4);
        s0 = *reinterpret_cast<const f32x4*>(sin_row + gl * 8); s1 = *reinterpret_cast<const f32x4*>(sin_row + gl * 8 + 4);
    }
    if (hv <= 32)
    {
        xlo = ld16(src + (size_t)gl * 8);
        xhi = ld16(src + (size_t)(gl + hv) * 8);
        const float ss = group_tree_sum(sumsq8(xlo, 0.0f), hv) + group_tree_sum(sumsq8(xhi, 0.0f), hv);
        rstd = rsqrtf(ss / (float)HS + p.eps);
    }
    else
    {
        // HS = 1024: one row per wave, two chained chunks per lane (rms_rstd_wave's own order)
        xlo = ld16(src + (size_t)gl * 8);
        xhi = ld16(src + (size_t)(gl + hv) * 8);
        rstd = rms_rstd_wave(src, HS, p.eps);
    }
    if (!valid) return;
    u32x4 lo, hi;
    if (w)
    {
        lo = rms_apply8(xlo, wlo, rstd, 0.0f);
        hi = rms_apply8(xhi, whi, rstd, 0.0f);
    }
    else
    {
        lo = rms_apply8_now(xlo, rstd);
        hi = rms_apply8_now(xhi, rstd);
    }
    if (rotate) rope_rotate8_regs(lo, hi, c0, c1, s0, s1);
    st16(dst + (size_t)gl * 8, lo);
    st16(dst + (size_t)(gl + hv) * 8, hi);
}

static int qkv_post_rows_per_wave(int HS)
{
    const int hv = HS / 16;
    return hv <= 32 ? 64 / hv : 1;
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_fused_qkv_post(uint16_t* q_out, uint16_t* Kc, uint16_t* Vc, const uint16_t* q, const uint16_t* k,
                              const uint16_t* v_src, const uint16_t* qw, const uint16_t* kw, const uint16_t* vw,
                              const float* cos_cache, const float* sin_cache, int NH, int NKV, int HS, int position,
                              int capacity, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(q_out && Kc && Vc && q && k && v_src && qw && kw && cos_cache && sin_cache, "fused_qkv_post: null pointer");
    MILA_REQUIRE(NH > 0 && NKV > 0 && capacity > 0 && position >= 0, "fused_qkv_post: bad sizes");
    MILA_REQUIRE(HS >= 16 && HS <= 1024 && (HS & (HS - 1)) == 0, "fused_qkv_post: HS=%d must be a power of two in [16,1024]", HS);
    QkvPostParams p{q_out, Kc, Vc, q, k, v_src, qw, kw, vw, cos_cache, sin_cache, nullptr, NH, NKV, HS, position, capacity, eps, 0};
    const int rows = NH + 2 * NKV;
    hipLaunchKernelGGL(qkv_post_kernel, dim3(ceil_div(rows, 4 * qkv_post_rows_per_wave(HS))), dim3(256), 0, as_stream(stream), p);
    MILA_LAUNCH_CHECK("fused_qkv_post");
}

int mila_cdna4_fused_qkv_post_prefill(uint16_t* q_out, uint16_t* Kc, uint16_t* Vc, const uint16_t* q, const uint16_t* k,
                                      const uint16_t* v_src, int64_t src_row_stride, const uint16_t* qw, const uint16_t* kw,
                                      const uint16_t* vw, const float* cos_cache, const float* sin_cache, int T, int NH, int NKV,
                                      int HS, int pos_offset, int capacity, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(q_out && Kc && Vc && q && k && v_src && qw && kw && cos_cache && sin_cache, "fused_qkv_post_prefill: null pointer");
    MILA_REQUIRE(T > 0 && T <= 65535 && NH > 0 && NKV > 0 && capacity > 0 && pos_offset >= 0, "fused_qkv_post_prefill: bad sizes");
    MILA_REQUIRE(T <= capacity, "fused_qkv_post_prefill: %d tokens do not fit the cache capacity %d", T, capacity);
    MILA_REQUIRE(HS >= 16 && HS <= 1024 && (HS & (HS - 1)) == 0, "fused_qkv_post_prefill: HS=%d must be a power of two in [16,1024]", HS);
    MILA_REQUIRE(src_row_stride % 8 == 0 && src_row_stride >= (int64_t)HS, "fused_qkv_post_prefill: row stride %lld must be a multiple of 8", (long long)src_row_stride);
    QkvPostParams p{q_out, Kc, Vc, q, k, v_src, qw, kw, vw, cos_cache, sin_cache, nullptr, NH, NKV, HS, pos_offset, capacity, eps, src_row_stride};
    const int rows = NH + 2 * NKV;
    hipLaunchKernelGGL(qkv_post_kernel, dim3(ceil_div(rows, 4 * qkv_post_rows_per_wave(HS)), T), dim3(256), 0, as_stream(stream), p);
    MILA_LAUNCH_CHECK("fused_qkv_post_prefill");
}

int mila_cdna4_fused_qkv_post_devpos(uint16_t* q_out, uint16_t* Kc, uint16_t* Vc, const uint16_t* q, const uint16_t* k,
                                     const uint16_t* v_src, const uint16_t* qw, const uint16_t* kw, const uint16_t* vw,
                                     const float* cos_cache, const float* sin_cache, int NH, int NKV, int HS,
                                     const int32_t* position_dev, int capacity, float eps, mila_stream_t stream)
{
    MILA_REQUIRE(q_out && Kc && Vc && q && k && v_src && qw && kw && cos_cache && sin_cache && position_dev,
                 "fused_qkv_post_devpos: null pointer");
    MILA_REQUIRE(NH > 0 && NKV > 0 && capacity > 0, "fused_qkv_post_devpos: bad sizes");
    MILA_REQUIRE(HS >= 16 && HS <= 1024 && (HS & (HS - 1)) == 0, "fused_qkv_post_devpos: HS=%d must be a power of two in [16,1024]", HS);
    QkvPostParams p{q_out, Kc, Vc, q, k, v_src, qw, kw, vw, cos_cache, sin_cache, position_dev, NH, NKV, HS, 0, capacity, eps, 0};
    const int rows = NH + 2 * NKV;
    hipLaunchKernelGGL(qkv_post_kernel, dim3(ceil_div(rows, 4 * qkv_post_rows_per_wave(HS))), dim3(256), 0, as_stream(stream), p);
    MILA_LAUNCH_CHECK("fused_qkv_post_devpos");
}

int mila_cdna4_advance_position(int32_t* position_dev, mila_stream_t stream)
{
    MILA_REQUIRE(position_dev != nullptr, "advance_position: null pointer");
    hipLaunchKernelGGL(advance_position_kernel, dim3(1), dim3(1), 0, as_stream(stream), position_dev);
    MILA_LAUNCH_CHECK("advance_position");
}

int mila_cdna4_advance_position_snapshot(int32_t* position_dev, const int32_t* token, unsigned long long* seq_dev, unsigned long long* ring, int ring_size, mila_stream_t stream)
{
    MILA_REQUIRE(position_dev != nullptr && token != nullptr && seq_dev != nullptr && ring != nullptr, "advance_position_snapshot: null pointer");
    MILA_REQUIRE(ring_size > 0, "advance_position_snapshot: ring_size must be positive");
    hipLaunchKernelGGL(advance_position_snapshot_kernel, dim3(1), dim3(1), 0, as_stream(stream), position_dev, token, seq_dev, ring, ring_size);
    MILA_LAUNCH_CHECK("advance_position_snapshot");
}

int mila_cdna4_snapshot_token(const int32_t* token, unsigned long long* seq_dev, unsigned long long* ring, int ring_size, mila_stream_t stream)
{
    MILA_REQUIRE(token != nullptr && seq_dev != nullptr && ring != nullptr, "snapshot_token: null pointer");
    MILA_REQUIRE(ring_size > 0, "snapshot_token: ring_size must be positive");
    hipLaunchKernelGGL(snapshot_token_kernel, dim3(1), dim3(1), 0, as_stream(stream), token, seq_dev, ring, ring_size);
    MILA_LAUNCH_CHECK("snapshot_token");
}

}  // extern "C"
