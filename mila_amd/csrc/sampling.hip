// Greedy device sampler: argmax over fp32 / bf16 logits, ties to the LOWEST index -- the reference's semantics
// (OPS/Sampling/Kernels/Sampling.cu:23-75: strict '>' while scanning in index order, lower index wins the reduction).
// Integer output => bit-exact.  Two stages so that 1 MB of logits is read by the whole chip, not by one CU.
#include <cfloat>

#include "common.h"

namespace mila {

constexpr int kArgmaxBlocks = kArgmaxPartials;      // (common.h) 512: the lm_head matvec runs two workgroups per CU

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<uint16_t>(uint16_t v) { return bf16_bits_to_f32(v); }

__device__ __forceinline__ void better(float& bv, int& bi, float v, int i)
{
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

__device__ __forceinline__ void wave_argmax(float& bv, int& bi)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        better(bv, bi, ov, oi);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_partial_kernel(const T* __restrict__ logits, float* __restrict__ pv, int* __restrict__ pi,
                                                             int vocab)
{
    __shared__ float sv[4];
    __shared__ int si[4];
    float bv = -FLT_MAX;
    int bi = 0x7fffffff;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) better(bv, bi, to_f32(logits[i]), i);
    wave_argmax(bv, bi);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int w = 1; w < 4; ++w) better(bv, bi, sv[w], si[w]);
        pv[blockIdx.x] = bv;
        pi[blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(256) void argmax_final_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int n,
                                                           int32_t* __restrict__ token_out)
{
    __shared__ float sv[4];
    __shared__ int si[4];
    float bv = -FLT_MAX;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) better(bv, bi, pv[i], pi[i]);
    wave_argmax(bv, bi);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int w = 1; w < 4; ++w) better(bv, bi, sv[w], si[w]);
        token_out[0] = (bi == 0x7fffffff) ? 0 : bi;      // all -FLT_MAX / NaN: the reference leaves index 0
    }
}

// the last node of a captured decode step: the final reduction of the greedy sampler, the position bump of the next step and (ring != NULL) the publication
// of the sampled token to the host -- what argmax_final + advance_position[_snapshot] did in two launches (round 3: one launch fewer per token)
__global__ __launch_bounds__(256) void argmax_final_advance_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int n, int32_t* __restrict__ token_out,
                                                                   int32_t* __restrict__ pos, unsigned long long* seq_dev, unsigned long long* ring, int ring_size)
{
    __shared__ float sv[4];
    __shared__ int si[4];
    float bv = -FLT_MAX;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) better(bv, bi, pv[i], pi[i]);
    wave_argmax(bv, bi);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int w = 1; w < 4; ++w) better(bv, bi, sv[w], si[w]);
        const int tok = (bi == 0x7fffffff) ? 0 : bi;
        token_out[0] = tok;
        *pos += 1;
        if (ring != nullptr)
        {
            const unsigned long long seq = *seq_dev + 1ull;
            *seq_dev = seq;
            __hip_atomic_store(ring + (seq % (unsigned long long)ring_size), (seq << 32) | (unsigned long long)(uint32_t)tok, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <typename T>
static int run_argmax(const T* logits, int32_t* token_out, int vocab, void* scratch, size_t scratch_bytes, hipStream_t s, const char* who)
{
    MILA_REQUIRE(logits && token_out, "%s: null pointer", who);
    MILA_REQUIRE(vocab > 0, "%s: vocab must be positive", who);
    const size_t need = (size_t)kArgmaxBlocks * 8;
    if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "%s: scratch %zu bytes < required %zu", who, scratch_bytes, need);
    float* pv = reinterpret_cast<float*>(scratch);
    int* pi = reinterpret_cast<int*>(pv + kArgmaxBlocks);
    int blocks = (vocab + 255) / 256;
    if (blocks > kArgmaxBlocks) blocks = kArgmaxBlocks;
    hipLaunchKernelGGL(argmax_partial_kernel<T>, dim3(blocks), dim3(256), 0, s, logits, pv, pi, vocab);
    int rc = check_hip(hipGetLastError(), who);
    if (rc) return rc;
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, s, pv, pi, blocks, token_out);
    return check_hip(hipGetLastError(), who);
}

// ================================================================================================
// Stochastic sampler: softcap + temperature, top-k, top-p (nucleus), inverse CDF in token-index order.
// Semantics of the reference's multinomial kernel (OPS/Sampling/Kernels/Sampling.cu:760-905), whose two 40-step value
// bisections converge to:  top-k survivors = values strictly above the (k+1)-th largest scaled logit;  nucleus = the
// smallest set of highest-probability survivors whose mass exceeds top_p * total (a tie enters as a whole);  token = the
// first index with e > 0 and cumulative >= r * total (vocab - 1 otherwise).
//
// CDNA4 design (not the reference's histogram pipeline): both thresholds are found EXACTLY by a 16-ary search over
// the 32-bit order-preserving key of the value -- 8 launches of one kernel, each evaluating 15 candidate thresholds in
// one pass over the (L2-resident, 1 MB) vector with 256 workgroups; counts are integers, masses are summed in a fixed
// order (per-thread sequential, DPP butterfly, ascending workgroup index), so a launch sequence is deterministic.
// Every kernel derives the current search state itself from the previous launch's partial sums: no host round trip.
constexpr int kSampBlocks = 256;
constexpr int kSampCand = 15;          // candidate thresholds per search step (16-ary)
constexpr int kSampSteps = 8;          // 16^8 = 2^32

struct SampParams
{
    float* w;                 // [vocab] scaled logits, then probabilities, then the masked probabilities
    uint32_t* part;           // [2][kSampCand][kSampBlocks] partial counts / masses (bit patterns)
    float* red;               // [kSampBlocks] per-workgroup max / sum partials
    uint32_t* lo_hist;        // [kSampSteps] search state after each step
    float* state;             // [0] max  [1] total  [2] k threshold key (bits)  [3] p threshold key (bits)
    int32_t* token_out;
    int vocab, top_k, nblocks;      // nblocks: workgroups of the scale / search / prob launches (<= kSampBlocks)
    float softcap, temperature, top_p, r;
};

__device__ __forceinline__ uint32_t ordered_key(float x)      // monotone: x < y  <=>  key(x) < key(y)
{
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// fixed-order sum of n <= 256 per-workgroup partials by one wave: lane l adds entries 4l .. 4l+3, then the butterfly
template <bool AS_FLOAT>
__device__ __forceinline__ float wave_sum_partials(const uint32_t* p, int n, int lane)
{
    float f = 0.0f;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int b = 4 * lane + i;
        const uint32_t v = b < n ? p[b] : 0u;
        if (AS_FLOAT) f += __uint_as_float(v); else c += v;
    }
    if (AS_FLOAT) return wave_sum(f);
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(c)));
}

// the search state at the start of step `step`: previous state advanced by the previous step's partials.
// COND: top-k  -> count(key >= t) >= k + 1 ;  top-p -> mass(key >= t) > target
template <bool MASS>
__device__ uint32_t advance_search(const SampParams& p, int step, float target, uint32_t* sh)
{
    if (step == 0) return 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lo_prev = p.lo_hist[step - 1];
    const uint32_t* prev = p.part + (size_t)((step - 1) & 1) * kSampCand * kSampBlocks;
    // wave w evaluates candidates w, w + 4, ...; sh[j] = 1 when candidate j + 1 still satisfies the condition
    for (int j = wave; j < kSampCand; j += 4)
    {
        const float v = wave_sum_partials<MASS>(prev + (size_t)j * kSampBlocks, p.nblocks, lane);
        bool ok;
        if (MASS) ok = v > target;
        else ok = __float_as_uint(v) >= (uint32_t)(p.top_k + 1);
        if (lane == 0) sh[j] = ok ? 1u : 0u;
    }
    __syncthreads();
    int jstar = 0;
    for (int j = 0; j < kSampCand; ++j) if (sh[j]) jstar = j + 1;      // monotone: the last satisfied candidate
    __syncthreads();
    const int shift = 28 - 4 * (step - 1);
    return lo_prev + ((uint32_t)jstar << shift);
}

template <typename T>
__global__ __launch_bounds__(256) void samp_scale_kernel(const T* __restrict__ logits, const SampParams p)
{
    __shared__ float sv[4];
    float mx = -FLT_MAX;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.vocab; i += gridDim.x * 256)
    {
        float x = to_f32(logits[i]);
        if (p.softcap > 0.0f) x = p.softcap * tanhf(x / p.softcap);
        x = x / p.temperature;
        p.w[i] = x;
        mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) p.red[blockIdx.x] = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
}

// one search step: evaluate the 15 candidates lo + j * 16^(7 - step) (j = 1..15) over this workgroup's elements
template <bool MASS>
__global__ __launch_bounds__(256) void samp_search_kernel(const SampParams p, int step)
{
    __shared__ uint32_t sh[16];
    __shared__ float sf[4][kSampCand];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float target = 0.0f;
    if (MASS) target = p.top_p * p.state[1];
    const uint32_t lo = advance_search<MASS>(p, step, target, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) p.lo_hist[step] = lo;
    const int shift = 28 - 4 * step;
    float accf[kSampCand];
    uint32_t accc[kSampCand];
#pragma unroll
    for (int j = 0; j < kSampCand; ++j) { accf[j] = 0.0f; accc[j] = 0u; }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.vocab; i += gridDim.x * 256)
    {
        const float v = p.w[i];
        const uint32_t key = MASS ? __float_as_uint(v) : ordered_key(v);      // probabilities are >= 0: bits are ordered
#pragma unroll
        for (int j = 0; j < kSampCand; ++j)
        {
            const uint32_t t = lo + ((uint32_t)(j + 1) << shift);
            const bool ge = (t > lo) && key >= t;            // t <= lo: the candidate wrapped past 2^32 (nothing is >= it)
            if (MASS) accf[j] += ge ? v : 0.0f; else accc[j] += ge ? 1u : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < kSampCand; ++j)
    {
        const float r = MASS ? wave_sum(accf[j]) : __uint_as_float(wave_sum_u32(accc[j]));
        if (lane == 0) sf[wave][j] = r;
    }
    __syncthreads();
    if (threadIdx.x < kSampCand)
    {
        const int j = threadIdx.x;
        uint32_t out;
        if (MASS) out = __float_as_uint(((sf[0][j] + sf[1][j]) + sf[2][j]) + sf[3][j]);
        else out = __float_as_uint(sf[0][j]) + __float_as_uint(sf[1][j]) + __float_as_uint(sf[2][j]) + __float_as_uint(sf[3][j]);
        p.part[(size_t)(step & 1) * kSampCand * kSampBlocks + (size_t)j * kSampBlocks + blockIdx.x] = out;
    }
}

// probabilities of the top-k survivors: e = x above the k threshold ? expf(x - max) : 0; per-workgroup sums
__global__ __launch_bounds__(256) void samp_prob_kernel(const SampParams p, int use_k, int nblocks_scale)
{
    __shared__ uint32_t sh[16];
    __shared__ float sv[4];
    const int lane = threadIdx.x & 63;
    // max over the scale kernel's partials (every workgroup computes it the same way)
    float mx = -FLT_MAX;
    for (int b = threadIdx.x; b < nblocks_scale; b += 256) mx = fmaxf(mx, p.red[b]);
    mx = wave_max(mx);
    if (lane == 0) sv[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
    __syncthreads();
    uint32_t kthr = 0u;
    if (use_k) kthr = advance_search<false>(p, kSampSteps, 0.0f, sh);    // key of the (k+1)-th largest value
    float sum = 0.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.vocab; i += gridDim.x * 256)
    {
        const float x = p.w[i];
        const float e = (!use_k || ordered_key(x) > kthr) ? expf(x - mx) : 0.0f;
        p.w[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) sv[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        // red[] is still being read by late workgroups as the max partials: the sums go to the second half
        p.red[kSampBlocks + blockIdx.x] = ((sv[0] + sv[1]) + sv[2]) + sv[3];
        if (blockIdx.x == 0) { p.state[0] = mx; p.state[2] = __uint_as_float(kthr); }
    }
}

// total = fixed-order sum of the per-workgroup sums (one wave)
__global__ __launch_bounds__(64) void samp_total_kernel(const SampParams p, int nblocks)
{
    const float t = wave_sum_partials<true>(reinterpret_cast<const uint32_t*>(p.red + kSampBlocks), nblocks, threadIdx.x);
    if (threadIdx.x == 0) p.state[1] = t;
}

// nucleus mask + contiguous-chunk sums for the index-order CDF: workgroup b owns tokens [b * chunk, (b + 1) * chunk)
__global__ __launch_bounds__(256) void samp_mask_kernel(const SampParams p, int use_p, int chunk)
{
    __shared__ uint32_t sh[16];
    __shared__ float sv[4];
    uint32_t pthr = 0u;
    if (use_p) pthr = advance_search<true>(p, kSampSteps, p.top_p * p.state[1], sh);     // bits of the nucleus boundary probability
    const int i0 = blockIdx.x * chunk, i1 = min(p.vocab, i0 + chunk);
    float sum = 0.0f;
    for (int i = i0 + threadIdx.x; i < i1; i += 256)
    {
        float e = p.w[i];
        if (use_p && __float_as_uint(e) < pthr) { e = 0.0f; p.w[i] = 0.0f; }
        sum += e;
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        p.red[blockIdx.x] = ((sv[0] + sv[1]) + sv[2]) + sv[3];
        if (blockIdx.x == 0) p.state[3] = __uint_as_float(pthr);
    }
}

// inverse CDF in token-index order: chunk sums locate the chunk, one lane walks it (and, if rounding left the target just
// beyond it, the following ones) with the reference's guard e > 0 && cumulative >= target; vocab - 1 otherwise
__global__ __launch_bounds__(64) void samp_cdf_kernel(const SampParams p, int nchunks, int chunk)
{
    if (threadIdx.x != 0) return;
    float total = 0.0f;
    for (int b = 0; b < nchunks; ++b) total += p.red[b];
    const float target = p.r * total;
    float before = 0.0f;
    int b = 0;
    while (b < nchunks - 1 && before + p.red[b] < target) { before += p.red[b]; ++b; }
    int result = p.vocab - 1;
    float cum = before;
    bool found = false;
    for (; b < nchunks && !found; ++b)
    {
        const int i1 = min(p.vocab, (b + 1) * chunk);
        for (int i = b * chunk; i < i1; ++i)
        {
            const float e = p.w[i];
            cum += e;
            if (e > 0.0f && cum >= target) { result = i; found = true; break; }
        }
    }
    p.token_out[0] = result;
}

constexpr size_t kSampHeaderFloats = 64;
static size_t samp_scratch_floats(int vocab)
{
    return kSampHeaderFloats + (size_t)2 * kSampBlocks /* red */ + (size_t)2 * kSampCand * kSampBlocks /* part */ + (size_t)vocab;
}

template <typename T>
static int run_stochastic(const T* logits, int32_t* token_out, int vocab, float softcap, float temperature, int top_k, float top_p, float r,
                          void* scratch, size_t scratch_bytes, hipStream_t s, const char* who)
{
    MILA_REQUIRE(logits && token_out, "%s: null pointer", who);
    MILA_REQUIRE(vocab > 0, "%s: vocab must be positive", who);
    MILA_REQUIRE(temperature > 0.0f, "%s: temperature must be > 0 (temperature <= 0 is the greedy sampler: sample_argmax)", who);
    MILA_REQUIRE(top_k >= 0 && top_p > 0.0f && r >= 0.0f && r <= 1.0f, "%s: need top_k >= 0, top_p > 0, 0 <= r <= 1", who);
    const size_t need = samp_scratch_floats(vocab) * 4;
    if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "%s: scratch %zu bytes < required %zu", who, scratch_bytes, need);
    float* f = reinterpret_cast<float*>(scratch);
    SampParams p{};
    p.state = f;
    p.lo_hist = reinterpret_cast<uint32_t*>(f + 16);
    p.red = f + kSampHeaderFloats;
    p.part = reinterpret_cast<uint32_t*>(p.red + 2 * kSampBlocks);
    p.w = reinterpret_cast<float*>(p.part + (size_t)2 * kSampCand * kSampBlocks);
    p.token_out = token_out; p.vocab = vocab; p.top_k = top_k; p.softcap = softcap; p.temperature = temperature; p.top_p = top_p; p.r = r;
    const int use_k = (top_k > 0 && top_k < vocab) ? 1 : 0;
    const int use_p = top_p < 1.0f ? 1 : 0;
    int blocks = (vocab + 255) / 256;
    if (blocks > kSampBlocks) blocks = kSampBlocks;
    p.nblocks = blocks;
    hipLaunchKernelGGL(samp_scale_kernel<T>, dim3(blocks), dim3(256), 0, s, logits, p);
    if (use_k)
        for (int step = 0; step < kSampSteps; ++step) hipLaunchKernelGGL(samp_search_kernel<false>, dim3(blocks), dim3(256), 0, s, p, step);
    hipLaunchKernelGGL(samp_prob_kernel, dim3(blocks), dim3(256), 0, s, p, use_k, blocks);
    hipLaunchKernelGGL(samp_total_kernel, dim3(1), dim3(64), 0, s, p, blocks);
    if (use_p)
        for (int step = 0; step < kSampSteps; ++step) hipLaunchKernelGGL(samp_search_kernel<true>, dim3(blocks), dim3(256), 0, s, p, step);
    const int chunk = (vocab + kSampBlocks - 1) / kSampBlocks;
    const int nchunks = (vocab + chunk - 1) / chunk;
    hipLaunchKernelGGL(samp_mask_kernel, dim3(nchunks), dim3(256), 0, s, p, use_p, chunk);
    hipLaunchKernelGGL(samp_cdf_kernel, dim3(1), dim3(64), 0, s, p, nchunks, chunk);
    return check_hip(hipGetLastError(), who);
}

}  // namespace mila

using namespace mila;

extern "C" {

size_t mila_cdna4_sample_scratch_bytes(void) { return (size_t)kArgmaxBlocks * 8; }

int mila_cdna4_sample_argmax_fp32(const float* logits, int32_t* token_out, int vocab, void* scratch, size_t scratch_bytes,
                                  mila_stream_t stream)
{
    return run_argmax<float>(logits, token_out, vocab, scratch, scratch_bytes, as_stream(stream), "sample_argmax_fp32");
}

int mila_cdna4_sample_argmax_advance_fp32(const float* logits, int32_t* token_out, int vocab, void* scratch, size_t scratch_bytes, int32_t* position_dev,
                                          unsigned long long* seq_dev, unsigned long long* ring, int ring_size, mila_stream_t stream)
{
    MILA_REQUIRE(logits && token_out && position_dev, "sample_argmax_advance_fp32: null pointer");
    MILA_REQUIRE(vocab > 0, "sample_argmax_advance_fp32: vocab must be positive");
    MILA_REQUIRE((ring == nullptr) == (seq_dev == nullptr) && (ring == nullptr || ring_size > 0), "sample_argmax_advance_fp32: ring, seq_dev and ring_size go together");
    const size_t need = (size_t)kArgmaxBlocks * 8;
    if (!scratch || scratch_bytes < need) return set_error(MILA_E_SCRATCH_TOO_SMALL, "sample_argmax_advance_fp32: scratch %zu bytes < required %zu", scratch_bytes, need);
    float* pv = reinterpret_cast<float*>(scratch);
    int* pi = reinterpret_cast<int*>(pv + kArgmaxBlocks);
    int blocks = (vocab + 255) / 256;
    if (blocks > kArgmaxBlocks) blocks = kArgmaxBlocks;
    hipLaunchKernelGGL(argmax_partial_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), logits, pv, pi, vocab);
    int rc = check_hip(hipGetLastError(), "sample_argmax_advance_fp32");
    if (rc) return rc;
    hipLaunchKernelGGL(argmax_final_advance_kernel, dim3(1), dim3(256), 0, as_stream(stream), pv, pi, blocks, token_out, position_dev, seq_dev, ring, ring_size);
    MILA_LAUNCH_CHECK("sample_argmax_advance_fp32");
}

// only the final stage, over `blocks` partials a lm_head launch left in `scratch` (fused_norm_matvec with argmax_scratch): the captured greedy step's tail in ONE launch
int mila_cdna4_sample_argmax_final_advance(int32_t* token_out, const void* scratch, size_t scratch_bytes, int blocks, int32_t* position_dev,
                                           unsigned long long* seq_dev, unsigned long long* ring, int ring_size, mila_stream_t stream)
{
    MILA_REQUIRE(token_out && scratch && position_dev, "sample_argmax_final_advance: null pointer");
    MILA_REQUIRE(blocks > 0 && blocks <= kArgmaxBlocks, "sample_argmax_final_advance: blocks %d out of range (1 .. %d)", blocks, kArgmaxBlocks);
    MILA_REQUIRE((ring == nullptr) == (seq_dev == nullptr) && (ring == nullptr || ring_size > 0), "sample_argmax_final_advance: seq_dev, ring and ring_size go together");
    if (scratch_bytes < (size_t)kArgmaxBlocks * 8) return set_error(MILA_E_SCRATCH_TOO_SMALL, "sample_argmax_final_advance: scratch %zu bytes < required %zu", scratch_bytes, (size_t)kArgmaxBlocks * 8);
    const float* pv = reinterpret_cast<const float*>(scratch);
    const int* pi = reinterpret_cast<const int*>(pv + kArgmaxBlocks);
    hipLaunchKernelGGL(argmax_final_advance_kernel, dim3(1), dim3(256), 0, as_stream(stream), pv, pi, blocks, token_out, position_dev, seq_dev, ring, ring_size);
    MILA_LAUNCH_CHECK("sample_argmax_final_advance");
}

int mila_cdna4_sample_argmax_bf16(const uint16_t* logits, int32_t* token_out, int vocab, void* scratch, size_t scratch_bytes,
                                  mila_stream_t stream)
{
    return run_argmax<uint16_t>(logits, token_out, vocab, scratch, scratch_bytes, as_stream(stream), "sample_argmax_bf16");
}

size_t mila_cdna4_sample_stochastic_scratch_bytes(int vocab) { return vocab > 0 ? samp_scratch_floats(vocab) * 4 : 0; }

int mila_cdna4_sample_stochastic_fp32(const float* logits, int32_t* token_out, int vocab, float softcap, float temperature, int top_k,
                                      float top_p, float r, void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    return run_stochastic<float>(logits, token_out, vocab, softcap, temperature, top_k, top_p, r, scratch, scratch_bytes, as_stream(stream),
                                 "sample_stochastic_fp32");
}

int mila_cdna4_sample_stochastic_bf16(const uint16_t* logits, int32_t* token_out, int vocab, float softcap, float temperature, int top_k,
                                      float top_p, float r, void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    return run_stochastic<uint16_t>(logits, token_out, vocab, softcap, temperature, top_k, top_p, r, scratch, scratch_bytes,
                                    as_stream(stream), "sample_stochastic_bf16");
}

}  // extern "C"
