// EXPERIMENT (round 3), MEASURED SLOWER than the eight-wave ping-pong kernel it was meant to replace: 942-1147 TFLOP/s against 1214-1358 on the Gemma shapes
// (M 2048; N 8192 / 30720, K 1920 .. 7680; profiles/r03_gemm4w.txt), bit-identical results.  The K-tile slope went from 1.55 to 1.75 us: with ONE wave per SIMD
// every LDS-DMA request (16 per wave and K-tile, ~60 issue cycles each) and every ds_read stalls that SIMD's only MFMA stream, where the eight-wave form
// hides them under the partner wave's MFMAs; and hipcc spends 56 v_accvgpr copies per K-tile keeping 256 accumulators in 256 AGPRs.  Kept in
// libmila_cdna4_experiments.so (mila_cdna4_exp_gemm4w_bf16 / _geglu_bf16; tools/bench_gemm4w.py); the product does not use it.
//
// bf16 GEMM, 256 x 256 x 64 tile, FOUR waves -- one per SIMD, each owning 128 x 128 of the tile -- instead of the eight of gemm256.hip.
//   Y[M,N] = X[M,K] * W[N,K]^T (+ bias)  /  GeGLU form,   M % 256 == 0, N % 256 == 0 (GeGLU: F % 128), K % 64 == 0.
//
// Why (round 3; profiles/r03_ksweep.txt): time(K) of the eight-wave ping-pong kernel is a line whose slope is 1.50-1.57 us per K-tile where the matrix cores
// alone need 1.0 us at the clock the chip holds -- the K loop itself runs at ~65 %, and the rest is what a K-tile costs around its MFMAs: 24 ds_read_b128 per
// wave (192 KB of LDS reads per K-tile and CU), four barriers, and two waves of every SIMD that can only alternate.  With 128 x 128 per wave
//   * a fragment feeds 8 MFMAs instead of 2-4: 32 ds_read_b128 per wave and K-tile = 128 KB per CU (a third less LDS traffic and energy);
//   * ONE barrier per K-tile: it sits between the two k-steps of a K-tile, where every wave has read its last fragment of K-tile t (the slot pair may be
//     restaged with t + 2) and its share of K-tile t + 1 has landed (it may be read);
//   * no two waves share a SIMD, so nothing is handed over: a wave's own ds_reads and LDS-DMA requests are issued in the gaps of its own MFMA stream
//     (an MFMA 16x16x32 holds the issue port for 8 of its 16 cycles), the fragments of k-step s + 1 arriving under the MFMAs of k-step s.
// Registers: 256 accumulators + 2 x 64 fragment registers + addresses: one wave per SIMD (512 registers).
// Same instruction (v_mfma_f32_16x16x32_bf16), same operand layout and the same K order per output element as gemm256_kernel => bit-identical results.
//
// LDS image, swizzle and tile order are gemm256.hip's (2 K-tile buffers x {W0, W1, X0, X1} x 16 KB; 16-byte slot ^= (row >> 1) & 7 on the SOURCE address and
// on the fragment read).  Staging is `buffer_load_dwordx4 ... lds` with the K offset in an SGPR: the per-lane offsets are computed once, no vector ALU work
// per K-tile.  A wave's 128 W rows are rows wr * 64 .. + 63 of BOTH W half-tiles, so that in the GeGLU form (half-tile 0 = gate rows, 1 = the matching up rows)
// gate and up of an output meet in one lane.
#include <type_traits>

#include "../common.h"
#include "../internal.h"

// MILA_G4W_DIRECTIVES: pin the MFMA / DS-read / LDS-DMA interleave with sched_group_barrier.  Off by default: with 256 accumulator registers (every AGPR) the
// directive form makes hipcc (ROCm 7.2) rotate accumulators through v_accvgpr copies -- hundreds of vector-ALU instructions per K-tile.
#ifdef MILA_G4W_DIRECTIVES
#define G4W_GROUP(mask, n, id) __builtin_amdgcn_sched_group_barrier(mask, n, id)
#else
#define G4W_GROUP(mask, n, id) ((void)0)
#endif

namespace mila {

struct Gemm256Params;      // gemm256.hip (same parameter block)
struct Gemm4wParams
{
    uint16_t* Y;
    const uint16_t* X;
    const uint16_t* W;
    const uint16_t* bias;
    int M, K, N, tiles_m, tiles_n;     // GEGLU: N = F output columns, W has 2 F rows [gate | up]
};

constexpr int k4HalfBytes = 128 * 128;
constexpr int k4BufBytes = 4 * k4HalfBytes;

__device__ __forceinline__ void grouped_tile4(int tile, int tiles_m, int tiles_n, int& tm, int& tn)
{
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int grp = tile / per_group, in = tile - grp * per_group;
    const int gm = min(GM, tiles_m - grp * GM);
    tm = grp * GM + in % gm;
    tn = in / gm;
}

__device__ __forceinline__ void store_pair16_4(uint16_t* row_pt, int g, u32x2 a, u32x2 b)
{
    const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    st16(row_pt + (g & 1) * 16 + 4 * (g & ~1), u32x4{r0[0], r1[0], r0[1], r1[1]});
}

template <bool GEGLU>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const Gemm4wParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int ntiles = p.tiles_m * p.tiles_n;
    int m0, n0, wrow1;
    {
        const int idv = blockIdx.x, xcd = idv & 7, qd = ntiles >> 3, rem = ntiles & 7;
        const int tile = ((xcd < rem) ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (idv >> 3);
        int tm, tn;
        grouped_tile4(tile, p.tiles_m, p.tiles_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * (GEGLU ? 128 : 256);
        wrow1 = GEGLU ? p.N + n0 : n0 + 128;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int K = p.K, nk = K / 64;

    // ---- staging: per-lane byte offsets inside a half-tile's 128 rows, computed once; the half-tile's first row and the K-tile go into the scalar offset ----
    const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.X), 0, 0x7fffffff, 0x00020000);
    const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.W), 0, 0x7fffffff, 0x00020000);
    int voff[4];
    {
        const int srow = lane >> 3, sslot = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int row = (i * 4 + wave) * 8 + srow;                     // chunk i * 4 + wave: 8 rows
            voff[i] = row * K * 2 + ((sslot ^ ((row >> 1) & 7)) << 4);
        }
    }
    const int sbase[4] = {n0 * K * 2, wrow1 * K * 2, m0 * K * 2, (m0 + 128) * K * 2};      // W0, W1, X0, X1 (bytes; < 2^31 checked by the launcher)
    auto stage_half = [&](int kt, int which) {
        unsigned char* dst = smem + (kt & 1) * k4BufBytes + which * k4HalfBytes;
        const int soff = __builtin_amdgcn_readfirstlane(sbase[which] + kt * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            if (which < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr)(dst + (i * 4 + wave) * 1024), 16, voff[i], soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr)(dst + (i * 4 + wave) * 1024), 16, voff[i], soff, 0, 0);
        }
    };
    auto stage_all = [&](int kt) { stage_half(kt, 0); stage_half(kt, 2); stage_half(kt, 3); stage_half(kt, 1); };

    // ---- fragments: lane (l15, g) reads row (.. + l15), logical slot 4 ks + g; the swizzle depends on l15 only ((pt * 16 + l15) >> 1 & 7 == (l15 >> 1) & 7) ----
    const int sw = (l15 >> 1) & 7;
    int aoff[2], boff[2];       // byte offsets of (pt = 0 / qt = 0, ks) inside a K-tile buffer; the other fragments are + immediates
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
    {
        aoff[ks] = (wr * 64 + l15) * 128 + (((4 * ks + g) ^ sw) << 4);
        boff[ks] = (2 + wc) * k4HalfBytes + l15 * 128 + (((4 * ks + g) ^ sw) << 4);
    }
    s16x8 fa0[8], fb0[8], fa1[8], fb1[8];
    auto read_frags = [&](s16x8 (&fa)[8], s16x8 (&fb)[8], int kt, int ks) {
        const unsigned char* buf = smem + (kt & 1) * k4BufBytes;
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) fa[pt] = *reinterpret_cast<const s16x8*>(buf + aoff[ks] + (pt >> 2) * k4HalfBytes + (pt & 3) * 2048);
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) fb[qt] = *reinterpret_cast<const s16x8*>(buf + boff[ks] + qt * 2048);
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto mfma_block = [&](const s16x8 (&fa)[8], const s16x8 (&fb)[8]) {
#pragma unroll
        for (int pt = 0; pt < 8; ++pt)
#pragma unroll
            for (int qt = 0; qt < 8; ++qt)
                acc[pt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[pt]), __builtin_bit_cast(bf16x8, fb[qt]), acc[pt][qt], 0, 0, 0);
    };

    // ---- prologue: K-tiles 0 and 1 requested, K-tile 0 landed and visible, its first k-step's fragments requested ----
    stage_all(0);
    if (nk > 1) { stage_all(1); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(fa0, fb0, 0, 0);

    // one K-tile; STAGE / NEXT are compile-time so that each k-step is ONE basic block (the schedule directives below act inside a block): the last two
    // K-tiles, which stage nothing / read nothing ahead, are peeled
    auto ktile = [&](int t, auto stage_c, auto next_c) {
        constexpr bool STAGE = decltype(stage_c)::value, NEXT = decltype(next_c)::value;
        // k-step 0: the MFMAs of (t, ks 0) with the fragment reads of (t, ks 1) in their gaps
        read_frags(fa1, fb1, t, 1);
        mfma_block(fa0, fb0);
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            G4W_GROUP(0x008, 2, 0);      // 2 MFMA
            G4W_GROUP(0x100, 1, 0);      // 1 DS read
        }
        G4W_GROUP(0x008, 32, 0);
        __builtin_amdgcn_sched_barrier(0);
        // every fragment of K-tile t is in registers, this wave's share of K-tile t + 1 has landed: one barrier makes both true of every wave
        if constexpr (NEXT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // k-step 1: the MFMAs of (t, ks 1) with the staging of K-tile t + 2 (into the slots just released) and the fragment reads of (t + 1, ks 0) in their gaps
        if constexpr (STAGE) stage_all(t + 2);
        if constexpr (NEXT) read_frags(fa0, fb0, t + 1, 0);
        mfma_block(fa1, fb1);
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            G4W_GROUP(0x008, 2, 1);      // 2 MFMA
            if constexpr (STAGE) G4W_GROUP(0x020, 1, 1);      // 1 LDS-DMA request
            G4W_GROUP(0x008, 2, 1);      // 2 MFMA
            if constexpr (NEXT) G4W_GROUP(0x100, 1, 1);       // 1 DS read
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    for (int t = 0; t + 2 < nk; ++t) ktile(t, yes{}, yes{});
    if (nk >= 2) ktile(nk - 2, no{}, yes{});
    ktile(nk - 1, no{}, no{});

    // ---- epilogue: D[p = 4 g + r][q = l15] -> Y[m0 + wc * 128 + qt * 16 + q][n0 + (pt >> 2) * 128 + wr * 64 + (pt & 3) * 16 + p], two sub-tiles per 16-byte store ----
    if constexpr (GEGLU)
    {
#pragma unroll
        for (int qt = 0; qt < 8; ++qt)
        {
            const int m = m0 + wc * 128 + qt * 16 + l15;
#pragma unroll
            for (int pp = 0; pp < 4; pp += 2)
            {
                auto out4 = [&](int pt) -> u32x2 {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(round_bf16(acc[pt][qt][e])) * round_bf16(acc[pt + 4][qt][e]);
                    return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                };
                store_pair16_4(p.Y + (size_t)m * p.N + n0 + wr * 64 + pp * 16, g, out4(pp), out4(pp + 1));
            }
        }
    }
    else
    {
#pragma unroll
        for (int qt = 0; qt < 8; ++qt)
        {
            const int m = m0 + wc * 128 + qt * 16 + l15;
#pragma unroll
            for (int pp = 0; pp < 8; pp += 2)
            {
                const int nb = n0 + (pp >> 2) * 128 + wr * 64 + (pp & 3) * 16;
                auto out4 = [&](int pt, int n) -> u32x2 {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[pt][qt][e];
                    if (p.bias)
                    {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]) + bf16_bits_to_f32(p.bias[n + e]);
                    }
                    return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                };
                store_pair16_4(p.Y + (size_t)m * p.N + nb, g, out4(pp, nb + 4 * g), out4(pp + 1, nb + 16 + 4 * g));
            }
        }
    }
}

template <bool GEGLU>
static int launch_gemm4w_t(const Gemm4wParams& p, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set)
    {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w_kernel<GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * k4BufBytes), "hipFuncSetAttribute(gemm4w)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm4w_kernel<GEGLU>), dim3(p.tiles_m * p.tiles_n), dim3(256), 2 * k4BufBytes, s, p);
    MILA_LAUNCH_CHECK("gemm4w");
}

// the operand tensors are addressed through 32-bit buffer offsets
bool gemm4w_addressable(int M, int K, int N_rows_of_W) { return (int64_t)M * K * 2 < 0x7fffffffll && (int64_t)N_rows_of_W * K * 2 < 0x7fffffffll; }

int launch_gemm4w(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s)
{
    Gemm4wParams p{Y, X, W, bias, M, K, N, M / 256, N / 256};
    return launch_gemm4w_t<false>(p, s);
}
int launch_gemm4w_geglu(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, hipStream_t s)
{
    Gemm4wParams p{Y, X, W, nullptr, M, K, F, M / 256, F / 128};
    return launch_gemm4w_t<true>(p, s);
}

}  // namespace mila

extern "C" {

int mila_cdna4_exp_gemm4w_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && W, "exp_gemm4w_bf16: null pointer");
    MILA_REQUIRE(M > 0 && M % 256 == 0 && N > 0 && N % 256 == 0 && K > 0 && K % 64 == 0, "exp_gemm4w_bf16: M, N multiples of 256 and K a multiple of 64 (%d, %d, %d)", M, N, K);
    MILA_REQUIRE(mila::gemm4w_addressable(M, K, N), "exp_gemm4w_bf16: operands beyond 2 GiB");
    return mila::launch_gemm4w(Y, X, W, bias, M, K, N, mila::as_stream(stream));
}

int mila_cdna4_exp_gemm4w_geglu_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, mila_stream_t stream)
{
    MILA_REQUIRE(Y && X && W, "exp_gemm4w_geglu_bf16: null pointer");
    MILA_REQUIRE(M > 0 && M % 256 == 0 && F > 0 && F % 128 == 0 && K > 0 && K % 64 == 0, "exp_gemm4w_geglu_bf16: M a multiple of 256, F of 128, K of 64 (%d, %d, %d)", M, F, K);
    MILA_REQUIRE(mila::gemm4w_addressable(M, K, 2 * F), "exp_gemm4w_geglu_bf16: operands beyond 2 GiB");
    return mila::launch_gemm4w_geglu(Y, X, W, M, K, F, mila::as_stream(stream));
}

}  // extern "C"
