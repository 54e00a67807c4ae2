// the any-head-size attention kernel (attention_generic.hip), shared by the prefill / MHA and the decode dispatchers
#pragma once
#include "common.h"

namespace mila {

// E = the element type of Q / K / V / Y: uint16_t (bf16 bit patterns) or float (the FP32 validation rows, OPS/OperationTraits.Cuda.ixx:274-282)
template <typename E>
struct GenericAttnParamsT
{
    E* Y;                     // [B * Tq, NH * HS]
    const E* Q;               // row (b, t): Q + b * q_b_stride + t * q_row_stride + h * HS
    const E* K;               // K + b * kv_b_stride + kvh * kv_h_stride + (pos % capacity) * kv_r_stride
    const E* V;
    int64_t q_b_stride, q_row_stride, kv_b_stride, kv_h_stride, kv_r_stride;
    int B, Tq, NH, NKV, HS, capacity, pos_offset, window;
    float scale;
};
using GenericAttnParams = GenericAttnParamsT<uint16_t>;

int launch_attn_generic(const GenericAttnParams& p, hipStream_t s);
int launch_attn_generic_f32(const GenericAttnParamsT<float>& p, hipStream_t s);

}  // namespace mila
