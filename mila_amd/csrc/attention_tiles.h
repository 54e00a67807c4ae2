// LDS tile images shared by the MFMA attention kernels (flash prefill, long-context MFMA decode): row-major [32 keys][HS] bf16, the 16-byte chunk index XORed
// with a function of the row.  The same involution is applied on write and on both kinds of read.
#pragma once

#include "common.h"

namespace mila {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kKeysPerTile = 32;

template <int HS>
__device__ __forceinline__ int k_off(int row, int chunk)      // ds_read_b128 of 16 rows x same chunk
{
    constexpr int ROWB = HS * 2, NCH = ROWB / 16;
    const int x = (NCH >= 16) ? (row & 15) : ((row >> 1) & (NCH - 1));
    return row * ROWB + (((chunk & ~(NCH >= 16 ? 15 : NCH - 1)) | ((chunk ^ x) & (NCH >= 16 ? 15 : NCH - 1))) << 4);
}
template <int HS>
__device__ __forceinline__ int v_off(int row, int chunk)      // ds_read_b64_tr_b16 blocks of 4 rows x 16 cols
{
    constexpr int ROWB = HS * 2, NCH = ROWB / 16;
    const int f = (((row & 3) << 2) | ((row >> 2) & 3)) & (NCH >= 16 ? 15 : NCH - 1);
    return row * ROWB + ((chunk ^ f) << 4);
}

}  // namespace mila
