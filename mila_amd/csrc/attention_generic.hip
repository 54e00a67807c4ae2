// Causal (optionally sliding-window) attention for ANY head size: the kernel behind the head sizes the MFMA flash kernels
// (attention_prefill.hip: 64 / 128 / 256 / 512) and the split-K flash-decode kernel (attention.hip) do not take -- e.g. the reference
// tests' own geometries (MultiHeadAttention.Cuda.cpp: model_dim 8, 2 heads => HS = 4; CudaGqaOp.Cuda.cpp: HS = 8).  Same semantics
// (SURVEY.md Appendix A): q head h reads KV head h / (NH / NKV); a query at absolute position t sees keys max(0, t - window + 1) .. t
// (window 0 = all); score = dot(q, k) * scale before max / exp; fp32 scores, probabilities and accumulators, bf16 only at the store.
//
// One wave per (batch, head, query row).  Keys are walked in blocks of 256: lanes own keys (score = a serial fp32 dot over HS), the
// block's scores go through LDS, the running maximum / sum are kept online, and for the P V product lanes own output dimensions
// (d = lane, lane + 64, ...: HS <= 512).  Not a fast kernel -- it reads K once per query row -- and not meant to be: every benchmarked
// shape takes the MFMA / split-K kernels; this one makes the op's contract hold for every geometry the reference accepts.
#include "common.h"
#include "attention_generic.h"

namespace mila {

constexpr int kGenKeys = 256;      // keys per block
constexpr int kGenMaxD = 8;        // output dimensions per lane: HS <= 512

__device__ __forceinline__ float elem_f32(uint16_t v) { return bf16_bits_to_f32(v); }
__device__ __forceinline__ float elem_f32(float v) { return v; }
__device__ __forceinline__ void elem_store(uint16_t* p, float v) { *p = f32_to_bf16_bits(v); }
__device__ __forceinline__ void elem_store(float* p, float v) { *p = v; }

template <typename E>
__global__ __launch_bounds__(256) void attn_generic_kernel(const GenericAttnParamsT<E> p)
{
    __shared__ float sc[4][kGenKeys];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + w;                  // (b, h, t)
    const int64_t nrows = (int64_t)p.B * p.NH * p.Tq;
    if (row >= nrows) return;
    const int t = (int)(row % p.Tq);
    const int h = (int)((row / p.Tq) % p.NH);
    const int b = (int)(row / ((int64_t)p.Tq * p.NH));
    const int HS = p.HS, kvh = h / (p.NH / p.NKV);
    const int pos = p.pos_offset + t;
    const int first = (p.window > 0) ? max(0, pos - p.window + 1) : 0;
    const E* q = p.Q + (int64_t)b * p.q_b_stride + (int64_t)t * p.q_row_stride + (int64_t)h * HS;
    const E* Kb = p.K + (int64_t)b * p.kv_b_stride + (int64_t)kvh * p.kv_h_stride;
    const E* Vb = p.V + (int64_t)b * p.kv_b_stride + (int64_t)kvh * p.kv_h_stride;
    float m = -INFINITY, l = 0.0f, acc[kGenMaxD];
#pragma unroll
    for (int i = 0; i < kGenMaxD; ++i) acc[i] = 0.0f;
    for (int k0 = first; k0 <= pos; k0 += kGenKeys)
    {
        const int nkeys = min(kGenKeys, pos - k0 + 1);
        float bm = -INFINITY;
        for (int j = lane; j < nkeys; j += 64)
        {
            const E* kr = Kb + (int64_t)((k0 + j) % p.capacity) * p.kv_r_stride;
            float s = 0.0f;
            for (int d = 0; d < HS; ++d) s = fmaf(elem_f32(q[d]), elem_f32(kr[d]), s);
            s *= p.scale;
            sc[w][j] = s;
            bm = fmaxf(bm, s);
        }
        bm = wave_max(bm);
        const float mn = fmaxf(m, bm);
        const float alpha = (m == -INFINITY) ? 0.0f : __expf(m - mn);
        float ls = 0.0f;
        for (int j = lane; j < nkeys; j += 64)
        {
            const float e = __expf(sc[w][j] - mn);
            sc[w][j] = e;
            ls += e;
        }
        ls = wave_sum(ls);
        l = l * alpha + ls;
        m = mn;
        // the wave's own LDS writes above are read by its other lanes below: LDS serves one wave's operations in issue order, the compiler is held to that order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < kGenMaxD; ++i)
        {
            const int d = lane + 64 * i;
            if (d < HS)
            {
                float a = acc[i] * alpha;
                for (int j = 0; j < nkeys; ++j)
                    a = fmaf(sc[w][j], elem_f32(Vb[(int64_t)((k0 + j) % p.capacity) * p.kv_r_stride + d]), a);
                acc[i] = a;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    E* y = p.Y + ((int64_t)b * p.Tq + t) * ((int64_t)p.NH * HS) + (int64_t)h * HS;
#pragma unroll
    for (int i = 0; i < kGenMaxD; ++i)
    {
        const int d = lane + 64 * i;
        if (d < HS) elem_store(y + d, acc[i] * inv);
    }
}

int launch_attn_generic(const GenericAttnParams& p, hipStream_t s)
{
    if (p.HS > 64 * kGenMaxD) return set_error(MILA_E_UNSUPPORTED, "attention: head size %d exceeds %d", p.HS, 64 * kGenMaxD);
    const int64_t rows = (int64_t)p.B * p.NH * p.Tq;
    hipLaunchKernelGGL(attn_generic_kernel<uint16_t>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, p);
    MILA_LAUNCH_CHECK("attn_generic");
}
int launch_attn_generic_f32(const GenericAttnParamsT<float>& p, hipStream_t s)
{
    if (p.HS > 64 * kGenMaxD) return set_error(MILA_E_UNSUPPORTED, "attention (fp32): head size %d exceeds %d", p.HS, 64 * kGenMaxD);
    const int64_t rows = (int64_t)p.B * p.NH * p.Tq;
    hipLaunchKernelGGL(attn_generic_kernel<float>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, p);
    MILA_LAUNCH_CHECK("attn_generic_f32");
}

}  // namespace mila
