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
// V image: ds_read_b64_tr_b16 blocks of 4 rows x 16 columns.  A transposing read is served in two halves of 32 lanes, each lane 8 bytes = one bank PAIR of the 32
// pairs: lane (g, q4 = row & 3 within its 4-row block, pp) of a half reads row 4 g + q4, columns 16 d + 4 pp .., i.e. pair index
//     [ (row pitch in dwords) row  mod 64 ] / 2  +  2 ((chunk ^ f) & 15) + (pp & 1),      chunk = 2 d + (pp >> 1).
// pp supplies the two low bits; the other three must come from q4 (2 bits) and the half's one bit of g = (row >> 2) & 1.  Round 4: f = q4 << 2 | g's bit << 1.
// (Rounds 1-3 put g into bit 0, where pp >> 1 already sits: lanes (pp >> 1, g) = (0, 0) and (1, 1) met on one pair -- every transposing read 2-way conflicted,
// SQ_LDS_BANK_CONFLICT = a third of the flash prefill's LDS cycles, profiles/r04_flash_prefill.txt.)  A 128-byte row (HS = 64) contributes row & 1 to bit 4 itself:
// f = (q4 >> 1) << 2 | g's bit << 1 there.
template <int HS>
__device__ __forceinline__ int v_off(int row, int chunk)
{
    constexpr int ROWB = HS * 2, NCH = ROWB / 16;
    static_assert(NCH >= 8, "head sizes from 64");
    const int gbit = ((row >> 2) & 1) << 1;
    const int f = (NCH >= 16) ? (((row & 3) << 2) | gbit) : ((((row >> 1) & 1) << 2) | gbit);
    return row * ROWB + ((chunk ^ f) << 4);
}

// ds_read_b64_tr_b16 as inline assembly, for kernels whose tiles arrive by LDS-DMA.  In front of the INTRINSIC form of this read the compiler places an
// s_waitcnt vmcnt(0) whenever a global_load_lds is outstanding (it cannot tell that the read's buffer is not the one being filled; plain ds_read_b128 of the same
// buffers get no such wait) -- in a double-buffered kernel that is a wait for the NEXT tile, issued a few instructions earlier, in front of every tile's V fragments:
// the prefetch was serialised (round 4: 1 125 of a tile's 3 541 cycles).  The compiler does not count these reads in lgkmcnt: lds_tr_wait() retires them and ties
// the fragment registers to the wait (a consumer cannot be scheduled above it); the compiler's own waits for its counted reads only become stricter.
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void lds_tr_wait(s16x8 (&v)[N])
{
    static_assert(N % 8 == 0, "fragments in groups of eight (the operand limit of one asm statement)");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; i += 8)
        asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]), "+v"(v[i + 4]), "+v"(v[i + 5]), "+v"(v[i + 6]), "+v"(v[i + 7]));
}

}  // namespace mila
