// the any-head-size attention kernel (attention_generic.hip), shared by the prefill / MHA and the decode dispatchers
#pragma once
#include "common.h"

namespace mila {

struct GenericAttnParams
{
    uint16_t* Y;              // [B * Tq, NH * HS]
    const uint16_t* Q;        // row (b, t): Q + b * q_b_stride + t * q_row_stride + h * HS
    const uint16_t* K;        // K + b * kv_b_stride + kvh * kv_h_stride + (pos % capacity) * kv_r_stride
    const uint16_t* V;
    int64_t q_b_stride, q_row_stride, kv_b_stride, kv_h_stride, kv_r_stride;
    int B, Tq, NH, NKV, HS, capacity, pos_offset, window;
    float scale;
};

int launch_attn_generic(const GenericAttnParams& p, hipStream_t s);

}  // namespace mila
