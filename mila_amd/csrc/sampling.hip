// Greedy device sampler: argmax over fp32 / bf16 logits, ties to the LOWEST index -- the reference's semantics
// (OPS/Sampling/Kernels/Sampling.cu:23-75: strict '>' while scanning in index order, lower index wins the reduction).
// Integer output => bit-exact.  Two stages so that 1 MB of logits is read by the whole chip, not by one CU.
#include <cfloat>

#include "common.h"

namespace mila {

constexpr int kArgmaxBlocks = 256;

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

}  // namespace mila

using namespace mila;

extern "C" {

size_t mila_cdna4_sample_scratch_bytes(void) { return (size_t)kArgmaxBlocks * 8; }

int mila_cdna4_sample_argmax_fp32(const float* logits, int32_t* token_out, int vocab, void* scratch, size_t scratch_bytes,
                                  mila_stream_t stream)
{
    return run_argmax<float>(logits, token_out, vocab, scratch, scratch_bytes, as_stream(stream), "sample_argmax_fp32");
}

int mila_cdna4_sample_argmax_bf16(const uint16_t* logits, int32_t* token_out, int vocab, void* scratch, size_t scratch_bytes,
                                  mila_stream_t stream)
{
    return run_argmax<uint16_t>(logits, token_out, vocab, scratch, scratch_bytes, as_stream(stream), "sample_argmax_bf16");
}

}  // extern "C"
