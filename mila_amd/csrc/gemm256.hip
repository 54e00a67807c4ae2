// bf16 GEMM, 256 x 256 x 64 tile, direct-to-LDS staging with a counted-vmcnt pipeline.
//   Y[M,N] = X[M,K] * W[N,K]^T (+ bias), M % 256 == 0, N % 256 == 0, K % 64 == 0 (the Gemma prefill shapes);
// other shapes and the quantized weight formats take gemm.hip's 128 x 128 register-staged kernel.
//
// Structure (re-derived from the MI355X guide's description of an 8-wave, 4-phase-per-K-tile schedule):
//   * 512 threads = 8 waves as 2 (P = W rows, the MFMA "M" side) x 4 (Q = X rows, the MFMA "N" side); the
//     product is computed transposed (A operand = W, B operand = X) so a lane's 4 accumulator registers are 4
//     consecutive output columns n -> 8-byte stores.
//   * LDS: 2 K-tile buffers x {W half 0, W half 1, X half 0, X half 1} x 16 KB = 128 KB.  A half-tile is
//     128 rows x 64 bf16, written by 16 wave-level global_load_lds_dwordx4 (1 KiB each, lane-linear); the
//     XOR swizzle (16-byte slot ^= (row >> 1) & 7) is applied to the SOURCE address and to the fragment read.
//   * every wave's 128 x 64 output is 2 x 2 quadrants of 64 x 32 taken from (W half hA, X half hB), so phase
//     (hA, hB) touches exactly two half-tiles; the quadrant order (0,0) (0,1) (1,1) (1,0) frees W0 after phase
//     1, X1 after phase 2, W1 and X0 after phase 3, and each phase issues the refill of one freed slot two
//     K-tiles ahead:   ph0: X0(t+1)  ph1: W1(t+1)  ph2: W0(t+2)  ph3: X1(t+2).
//     Every half-tile is requested >= 4 phases before its first use; three half-tiles (6 wave-instructions)
//     stay in flight across every barrier: s_waitcnt vmcnt(6) + one raw s_barrier per phase, never vmcnt(0) in
//     the steady state.  A staged slot is read one phase after the wait + barrier that retires it.
//   * fragments: ds_read_b128, 12 / 4 / 8 / 4 per phase (the W or X fragments of the previous phase are reused).
#include "common.h"
#include <type_traits>

namespace mila {

struct Gemm256Params
{
    uint16_t* Y;
    const uint16_t* X;
    const uint16_t* W;
    const uint16_t* bias;
    int M, K, N, tiles_m, tiles_n;     // GEGLU: N = F output columns, W has 2 F rows [gate | up]
    // FP8 mode (W4A8 / W8A8 prefill): X and W are e4m3 bytes [M, K] / [N, K]; y = bf16(float(bf16(acc * *w_scale)) * x_scales[m] + bias)
    const float* x_scales;   // [M] per-token activation scales
    const float* w_scale;    // device scalar: per-tensor weight scale
    int rowwise = 0;         // 256 x 256 plain bf16, one workgroup per tile: the tile goes through LDS and is written row by row (see rowwise_epilogue)
    int act = 0;             // bf16 plain epilogue: 1 = tanh-GELU on the stored Linear output, y = bf16(gelu(bf16(acc) [+ bias, rounded again])): Linear + Gelu of MLP.ixx:148-161 in one kernel
    float* partials = nullptr;      // split-K form of the 256 x 128 ring: [splitk][M][N] fp32 accumulators (workspace), summed and finished by splitk_reduce_kernel
    int splitk = 0;
    int ldy = 0;             // row pitch of Y in elements (0 = N): a column range of a wider output (the column split of launch_*_colsplit: Y + n_first, pitch = the whole N)
    int w_pc = 0;            // FP8 mode: w_scale is a per-channel vector over the W rows (W8A8: y = bf16((acc * w_scale[n]) * x_scales[m] + bias), common.h) instead of the W4A8 scalar
#ifdef MILA_GEMM_SKIP
    int dbg = 0;             // diagnostic build (tools/experiments/gemm_skip.sh): leave out the staging (1), the fragment reads (2), the MFMAs (4), the plain epilogue's stores (8)
#endif
};
typedef int i32x8 __attribute__((ext_vector_type(8)));
enum { G_PLAIN = 0, G_GEGLU = 1, G_FP8 = 2, G_FP8_GEGLU = 3 };

constexpr int kHalfBytes = 128 * 128;          // 128 rows x 64 bf16
constexpr int kBufBytes = 4 * kHalfBytes;      // W0 W1 X0 X1

__device__ __forceinline__ int half_off(bool isX, int half) { return (isX ? 2 * kHalfBytes : 0) + half * kHalfBytes; }

// GEGLU: the tile's two W half-tiles are 128 gate rows (n0 ..) and the matching 128 up rows (F + n0 ..), so a lane's
// accumulators acc[0][..] / acc[1][..] hold gate and up of the SAME outputs and the epilogue writes
// bf16(gelu_tanh(bf16(gate)) * bf16(up)) -- the Linear + GeGLU pair of Gemma.Block.ixx:343-348 without the [M, 2F] round trip.
// MODE G_FP8: the same tile geometry in BYTES (a half-tile is 128 rows x 128 B = 128 e4m3 of K), one
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) per 16 x 16 sub-tile and K-tile instead of two 16x16x32 bf16 MFMAs
// over half the K: twice the FLOPs per LDS byte and per barrier.  Lane l supplies row l & 15 and the 32 bytes
// k = 32 (l >> 4) .. + 31 of both operands (selftest_mfma_fp8 pins that layout).
// tile index -> (tm, tn) in GROUPED order: groups of up to 8 row-blocks, tm fastest inside a group.  After the XCD remap the 32
// workgroups an XCD runs together are then 8 row-blocks x 4 column-blocks (12 distinct operand blocks per K-step through its L2
// instead of 1 + 32), and across the launch every XCD streams 1/8 of W instead of all of it.
__device__ __forceinline__ void grouped_tile(int tile, int tiles_m, int tiles_n, int& tm, int& tn)
{
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int grp = tile / per_group, in = tile - grp * per_group;
    const int gm = min(GM, tiles_m - grp * GM);
    tm = grp * GM + in % gm;
    tn = in / gm;
}

// (Round 3, same-box A/B of two builds -- tools/experiments/ab_prev_lib.py, profiles/r03_epilogue_ab.txt: the ORDER of a quadrant's four stores (a 128-byte line's halves in
// consecutive instructions, or 16 rows apart) changes nothing in the GEMM although it is worth 3 x in a store-only kernel (tools/experiments/store_drain.hip: 1.5 vs 4.4 us per
// 128 KB tile and CU); the bias as one 8-byte load per four outputs instead of four 2-byte loads is 10-13 % SLOWER on the biased GPT-2 projections and, hoisted over the
// 256 x 256 epilogue, spilled 127 registers: both left out.)
// Epilogue store of two 16-column sub-tiles (pt, pt + 1) of one output row as ONE 16-byte store per lane.  A lane (l15, g) holds columns 4 g .. 4 g + 3 of both
// sub-tiles (a, b); v_permlane16_swap exchanges the odd lane rows of a with the even lane rows of b, after which an even-g lane holds columns 4 g .. 4 g + 7 of
// sub-tile pt and an odd-g lane columns 4 (g - 1) .. 4 (g - 1) + 7 of sub-tile pt + 1: half the store instructions, 64 contiguous bytes per row and instruction.
template<bool ALIGNED = true>
__device__ __forceinline__ void store_pair16(uint16_t* row_pt, int g, u32x2 a, u32x2 b, bool ok = true)
{
    const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);      // every lane takes part in the exchange; `ok` (row < M) only guards the store
    const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    if (!ok) return;
    if constexpr (ALIGNED) st16(row_pt + (g & 1) * 16 + 4 * (g & ~1), u32x4{r0[0], r1[0], r0[1], r1[1]});
    else st16_a2(row_pt + (g & 1) * 16 + 4 * (g & ~1), u32x4{r0[0], r1[0], r0[1], r1[1]});      // odd row pitch: rows start on 2-byte boundaries
}

// a lane's byte offsets into W for its two staging requests per half-tile (chunks wave and 8 + wave of 16 chunks of 8 rows; the XOR swizzle lives on the source
// address).  WABS: absolute rows, clamped to the last row of W (ragged N); else relative to the half-tile's first row
template <bool WABS, bool FP8>
__device__ __forceinline__ void gemm256_w_offsets(int (&vo)[2][2], int wrow0, int wrow1, int wave, int srow, int sslot, int rowbytes, int n_rows)
{
#pragma unroll
    for (int half = 0; half < (WABS ? 2 : 1); ++half)
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
            const int row = (i * 8 + wave) * 8 + srow;             // chunk i * 8 + wave: 16 chunks of 8 rows
            const int lslot = sslot ^ ((row >> 1) & 7);            // swizzle on the source address: LDS slot sslot of this row holds logical slot lslot
            // fp8: a lane's operand is the 32 contiguous bytes k = 32 g .. (source chunks 2 g, 2 g + 1); they are kept at the logical slots g and 4 + g the
            // bf16 fragments use, so the fragment reads are the same conflict-free pattern (read as 2 g + ks they pair up on the banks: 2-way conflicts)
            const int kslot = FP8 ? (((lslot & 3) << 1) | (lslot >> 2)) : lslot;
            vo[half][i] = (WABS ? min((half ? wrow1 : wrow0) + row, n_rows - 1) : row) * rowbytes + kslot * 16;
        }
}

// PP (ping-pong): waves 4-7 run one barrier behind waves 0-3 and every phase has TWO barriers, [stage + fragment reads + waits] | A |
// [16 MFMAs] | B |, so that while one wave of a SIMD multiplies, the other one reads its fragments and issues the staging: LDS
// reads (28 ds_read_b128 per wave and K-tile, as many LDS cycles per CU as a SIMD has MFMA cycles) leave the critical path.
// Hazards under the stagger (phases p of both groups; group 1's phase p runs half a phase later):
//   RAW  a half-tile staged in phase ps is waited for by every wave at the end of its reads of phase ps + 3 (vmcnt(6)) and
//        read from phase ps + 4 on -- behind barrier B(ps + 3) of group 0 = A(ps + 3) of group 1, which all eight waves reach
//        only after that wait;
//   WAR  a slot is restaged >= 2 phases after its last read, except X0 (read in ph3, restaged in ph0 of the next K-tile): every
//        wave retires its own fragment reads (lgkmcnt(0)) BEFORE barrier A of the reading phase, and group 0's ph0 staging is
//        issued behind its barrier B(ph3) = group 1's A(ph3).
//
// EPILOGUE_LOADS_FIRST (round 4).  Everything an epilogue loads (per-token and per-channel scales, the ring's bias) is pinned into its registers -- an empty asm with the
// values as in / out operands -- BEFORE the first store.  The compiler waits for a load at its first use and its vmcnt bookkeeping does not count the stores issued in
// between, while the hardware counts loads and stores in one order: a scale first used behind five stores came with an s_waitcnt vmcnt(0) that retired all five before the
// sixth was issued -- the tile's write burst ran one store at a time (fp8 qkv: 17.7 of 75 us were the epilogue, 10 of the bf16 kernel's 116; tools/experiments/gemm_skip.py).
// PP == 2: the same stagger with TWO phases per K-tile (32 MFMAs per slot, half the barriers, 24 fragment reads per K-tile: the
// fragments of both X halves stay live).  Phase A(t): stage W1(t+1); read W0, X0, X1 of t; quadrants (0,0) (0,1).
// Phase B(t): stage W0, X0, X1 of t+2; read W1 of t; quadrants (1,0) (1,1).  vmcnt(8) = the stages of the two youngest phases stay
// in flight; every slot is restaged one phase after its last read (lgkmcnt(0) before barrier A) and read one phase after its wait.
template <int MODE, int PP>
__global__ __launch_bounds__(512) void gemm256_kernel(const Gemm256Params p)
{
    constexpr bool GEGLU = (MODE == G_GEGLU || MODE == G_FP8_GEGLU);
    constexpr bool FP8 = (MODE == G_FP8 || MODE == G_FP8_GEGLU);
    constexpr int ES = FP8 ? 1 : 2;            // bytes per element
    constexpr int KT = 128 / ES;               // elements of K per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // Workgroup id -> tiles id, id + gridDim.x, ... (the launcher starts one workgroup per CU when the two-phase schedule runs PERSISTENT, else one per tile).
    // The XCD remap is taken over the whole tile list: a workgroup's later tiles are the ones the dispatcher would have handed to its XCD anyway.
    const int ntiles = p.tiles_m * p.tiles_n;
    auto tile_origin = [&](int idv, int& m0_, int& n0_, int& wrow1_) {
        const int xcd = idv & 7, qd = ntiles >> 3, rem = ntiles & 7;
        const int tile = ((xcd < rem) ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (idv >> 3);
        int tm, tn;
        grouped_tile(tile, p.tiles_m, p.tiles_n, tm, tn);
        m0_ = tm * 256;
        n0_ = tn * (GEGLU ? 128 : 256);
        wrow1_ = GEGLU ? p.N + n0_ : n0_ + 128;            // first W row of half-tile 1
    };
    int m0, n0, wrow1, xm0 = 0, xn0 = 0, xwrow1 = 0;       // this tile and (persistent form) the next one
    tile_origin(blockIdx.x, m0, n0, wrow1);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    const int K = p.K, nk = K / KT;
    const int ldy = p.ldy ? p.ldy : p.N;                   // output row pitch
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(p.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);

    // ---- staging: half-tile (isX, half) of K-tile kt into buffer kt & 1 ----
    // `buffer_load_dwordx4 ... lds` with the K-tile (and, for W, the tile's first row) in the SCALAR offset: a lane's two byte offsets per operand are computed once per
    // tile, so a staging request costs no vector ALU work.  (Round 3: the global_load_lds form recomputed a 64-bit address per request -- two v_mul_lo_u32, a
    // v_mad_u64_u32 and five more vector instructions in front of each of the 16 requests of a K-tile, ~500 issue cycles per wave and K-tile inside the very
    // phases that must fit under the partner wave's 512 MFMA cycles; profiles/r03_ksweep.txt: the K loop ran at 1.5 us per K-tile where the MFMAs need 1.0.)
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Xb), 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Wb), 0, 0x7fffffff, 0x00020000);
    const int srow = lane >> 3, sslot = lane & 7;          // this lane's row within a 1 KiB chunk / 16-byte slot
    const int rowbytes = K * ES;
    // absolute byte offsets of this lane's two requests per half-tile, for this tile and (persistent form) the next one.  Rows past the tensors are CLAMPED: X rows past
    // M re-read row M - 1 (a ragged last tile-row), W rows past the last one re-read it (a ragged last tile-column, GPT-2's N = 50257) -- their products are never stored
    // (W: only the plain bf16 mode takes a ragged N and keeps absolute, clamped W offsets -- WABS; the other modes keep two offsets relative to the half-tile's first
    // row, which then rides in the scalar offset: they have no registers to spare)
    constexpr bool WABS = MODE == G_PLAIN;
    int voffW[2][2], voffWn[2][2], voffX[2][2], voffXn[2][2];      // (relative form: row [0] only)
    auto x_offsets = [&](int (&vo)[2][2], int mm0) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 2; ++i)
            {
                const int row = (i * 8 + wave) * 8 + srow;
                const int lslot = sslot ^ ((row >> 1) & 7);
                const int kslot = FP8 ? (((lslot & 3) << 1) | (lslot >> 2)) : lslot;
                vo[half][i] = min(mm0 + half * 128 + row, p.M - 1) * rowbytes + kslot * 16;
            }
    };
    auto w_offsets = [&](int (&vo)[2][2], int wrow0_, int wrow1_) {
        gemm256_w_offsets<WABS, FP8>(vo, wrow0_, wrow1_, wave, srow, sslot, rowbytes, p.N);
    };
    w_offsets(voffW, n0, wrow1);
    x_offsets(voffX, m0);
    auto stage = [&](int kt_flat, bool isX, int half) {
        // persistent form: K-tile nk + k is K-tile k of the workgroup's NEXT tile (nk is even there, so the buffer parity runs on)
        const bool nxt = kt_flat >= nk;
        const int kt = nxt ? kt_flat - nk : kt_flat;
        unsigned char* dst_half = smem + (kt & 1) * kBufBytes + half_off(isX, half);
        const int wrow = nxt ? (half ? xwrow1 : xn0) : (half ? wrow1 : n0);
        const int soff = __builtin_amdgcn_readfirstlane((isX || WABS) ? kt * 128 : wrow * rowbytes + kt * 128);
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
            const int chunk = i * 8 + wave;
            if (isX) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(dst_half + chunk * 1024), 16, nxt ? voffXn[half][i] : voffX[half][i], soff, 0, 0);
            else if constexpr (WABS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(dst_half + chunk * 1024), 16, nxt ? voffWn[half][i] : voffW[half][i], soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(dst_half + chunk * 1024), 16, voffW[0][i], soff, 0, 0);
        }
    };

    f32x4 acc[2][2][4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    zero_acc();

    // fragment registers: [ks] = the logical 16-B slot 4 ks + g of the row (bf16: k = 8 (4 ks + g) ..; fp8: source chunk 2 g + ks, the halves of one 32-byte operand)
    constexpr bool P2 = PP >= 2;            // two phases per K-tile (PP == 3: static priority for waves 4-7 instead of a raise around every MFMA block)
    s16x8 fa[4][2], fb[2][2], fb1[P2 ? 2 : 1][2];
    auto frag_slot = [&](int ks) { return ks * 4 + g; };
    auto load_a = [&](int kt, int hA) {
        const unsigned char* hb = smem + (kt & 1) * kBufBytes + half_off(false, hA);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
        {
            const int r = wr * 64 + pt * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fa[pt][ks] = *reinterpret_cast<const s16x8*>(hb + r * 128 + ((frag_slot(ks) ^ ((r >> 1) & 7)) << 4));
        }
    };
    auto load_b = [&](int kt, int hB) {
        const unsigned char* hb = smem + (kt & 1) * kBufBytes + half_off(true, hB);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
        {
            const int r = wc * 32 + qt * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
            {
                const s16x8 v = *reinterpret_cast<const s16x8*>(hb + r * 128 + ((frag_slot(ks) ^ ((r >> 1) & 7)) << 4));
                if (P2 && hB == 1) fb1[P2 ? qt : 0][ks] = v;
                else fb[qt][ks] = v;
            }
        }
    };
    auto mma = [&](int hA, int hB) {
        s16x8 (&fbx)[2][2] = *((P2 && hB == 1) ? reinterpret_cast<s16x8 (*)[2][2]>(&fb1) : &fb);
        if constexpr (PP != 3) __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8)
        {
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                {
                    struct Pair { s16x8 lo, hi; };
                    const i32x8 a8 = __builtin_bit_cast(i32x8, (Pair{fa[pt][0], fa[pt][1]}));
                    const i32x8 b8 = __builtin_bit_cast(i32x8, (Pair{fbx[qt][0], fbx[qt][1]}));
                    acc[hA][hB][pt][qt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[hA][hB][pt][qt], 0, 0, 0, 127, 0, 127);
                }
            // PIN the products to their phase (round 4).  The scaled-MFMA calls are pure, and the compiler SANK the 16 of phase A below that phase's end barrier and
            // phase B's reads, into phase B's MFMA slot (the kernel's barrier-to-barrier instruction counts read 0 / 32 MFMAs where the bf16 modes read 16 / 16
            // -- sched_barrier(0) does not hold them): one slot of every K-tile then held 32 x 32 cycles of matrix work with the partner wave idle behind it, the next
            // none -- the staggered schedule was a lockstep one for every fp8 x fp8 GEMM (1.79 us per K-tile against bf16's 1.46 for the same bytes and MFMA cycles).
            // An empty volatile statement that takes the accumulators as in / out operands keeps each block in front of its own barrier; no instruction is emitted.
            asm volatile("" : "+v"(acc[hA][hB][0][0]), "+v"(acc[hA][hB][0][1]), "+v"(acc[hA][hB][1][0]), "+v"(acc[hA][hB][1][1]),
                              "+v"(acc[hA][hB][2][0]), "+v"(acc[hA][hB][2][1]), "+v"(acc[hA][hB][3][0]), "+v"(acc[hA][hB][3][1]));
        }
        else
        {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        acc[hA][hB][pt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[pt][ks]), __builtin_bit_cast(bf16x8, fbx[qt][ks]), acc[hA][hB][pt][qt], 0, 0, 0);
        }
        if constexpr (PP != 3) __builtin_amdgcn_s_setprio(0);
    };
    // end of a phase: retire everything but the 3 youngest half-tiles, then let every wave see it
    auto phase_end = [&](bool steady) {
        if (steady) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    auto epilogue = [&]() {
        // ---- epilogue: D[p = 4 g + r][q = l15] -> Y[m0 + q][n0 + p], two sub-tiles per 16-byte store (store_pair16) ----
        if constexpr (GEGLU)
        {
            // fp8 epilogue scales, fetched ONCE before the stores: read inside the store loop they are re-fetched after every store (the compiler cannot
            // prove that Y does not alias them) -- a dependent global load in front of each of the 16 stores, 7-11 us per tile
            float tsv[2][2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
            f32x4 wsg[4], wsu[4];      // weight scales of this lane's gate / up columns per sub-tile pt (W4A8: the per-tensor scalar in every slot)
            const bool pc = FP8 && p.w_pc;
            if constexpr (FP8)
            {
                if (pc)
                {
                    // the lane id re-derived HERE (mbcnt on an opaque zero): the column arithmetic below is then not hoisted out of the tile loop, where it -- or the
                    // lane id kept live for it -- cost the one register this mode does not have (a spill whose reload sits behind an s_waitcnt vmcnt(0))
                    int lz = 0;
                    asm volatile("" : "+v"(lz));
                    const int lq = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)lz));
            #pragma unroll
                    for (int pt_ = 0; pt_ < 4; ++pt_)
                    {
                        const int n_ = n0 + wr * 64 + pt_ * 16 + 4 * (lq >> 4);
                        wsg[pt_] = *reinterpret_cast<const f32x4*>(p.w_scale + n_);
                        wsu[pt_] = *reinterpret_cast<const f32x4*>(p.w_scale + p.N + n_);
                    }
                    asm volatile("" : "+v"(wsg[0]), "+v"(wsg[1]), "+v"(wsg[2]), "+v"(wsg[3]), "+v"(wsu[0]), "+v"(wsu[1]), "+v"(wsu[2]), "+v"(wsu[3]));
                }
                else
                {
                    const float ws_ = *p.w_scale;
            #pragma unroll
                    for (int pt_ = 0; pt_ < 4; ++pt_) wsg[pt_] = wsu[pt_] = f32x4{ws_, ws_, ws_, ws_};
                }
            #pragma unroll
                for (int hb_ = 0; hb_ < 2; ++hb_)
            #pragma unroll
                    for (int qt_ = 0; qt_ < 2; ++qt_) tsv[hb_][qt_] = p.x_scales[min(m0 + hb_ * 128 + wc * 32 + qt_ * 16 + l15, p.M - 1)];
                asm volatile("" : "+v"(tsv[0][0]), "+v"(tsv[0][1]), "+v"(tsv[1][0]), "+v"(tsv[1][1]));      // (in their registers before the first store: EPILOGUE_LOADS_FIRST)
            }
            auto out4 = [&](int hB, int pt, int qt, int m, auto pc_c) -> u32x2 {
                float v[4];
                if constexpr (FP8)
                {
                    // gate / up as the Linear stores them (W4A8: bf16(float(bf16(acc * sB)) * s_m); W8A8: bf16((acc * s_c[n]) * s_m)), then the GeGLU kernel's product
                    // (the weight-scale kind is chosen once per tile, not per element)
                    constexpr bool PC = decltype(pc_c)::value;
                    const float ts = tsv[hB][qt];
    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = gelu_tanh(fp8_linear_out(PC, acc[0][hB][pt][qt][e], wsg[pt][e], ts)) * fp8_linear_out(PC, acc[1][hB][pt][qt][e], wsu[pt][e], ts);
                }
                else
                {
    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = gelu_tanh(round_bf16(acc[0][hB][pt][qt][e])) * round_bf16(acc[1][hB][pt][qt][e]);
                }
                return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            };
            auto stores = [&](auto pc_c) {
    #pragma unroll
                for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                    for (int pp = 0; pp < 4; pp += 2)
    #pragma unroll
                        for (int qt = 0; qt < 2; ++qt)
                        {
                            const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                            store_pair16(p.Y + (size_t)m * ldy + n0 + wr * 64 + pp * 16, g, out4(hB, pp, qt, m, pc_c), out4(hB, pp + 1, qt, m, pc_c), m < p.M);
                        }
            };
            if (pc) stores(std::true_type{}); else stores(std::false_type{});
        }
        else
        {
            // fp8 epilogue scales, fetched ONCE before the stores: read inside the store loop they are re-fetched after every store (the compiler cannot
            // prove that Y does not alias them) -- a dependent global load in front of each of the 16 stores, 7-11 us per tile
            float tsv[2][2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
            f32x4 wsv[2][4];      // weight scales of this lane's columns per (W half, sub-tile) (W4A8: the per-tensor scalar in every slot)
            const bool pc = FP8 && p.w_pc;
            if constexpr (FP8)
            {
                if (pc)
                {
            #pragma unroll
                    for (int ha_ = 0; ha_ < 2; ++ha_)
            #pragma unroll
                        for (int pt_ = 0; pt_ < 4; ++pt_) wsv[ha_][pt_] = *reinterpret_cast<const f32x4*>(p.w_scale + n0 + ha_ * 128 + wr * 64 + pt_ * 16 + 4 * g);
                    asm volatile("" : "+v"(wsv[0][0]), "+v"(wsv[0][1]), "+v"(wsv[0][2]), "+v"(wsv[0][3]), "+v"(wsv[1][0]), "+v"(wsv[1][1]), "+v"(wsv[1][2]), "+v"(wsv[1][3]));
                }
                else
                {
                    const float ws_ = *p.w_scale;
            #pragma unroll
                    for (int ha_ = 0; ha_ < 2; ++ha_)
            #pragma unroll
                        for (int pt_ = 0; pt_ < 4; ++pt_) wsv[ha_][pt_] = f32x4{ws_, ws_, ws_, ws_};
                }
            #pragma unroll
                for (int hb_ = 0; hb_ < 2; ++hb_)
            #pragma unroll
                    for (int qt_ = 0; qt_ < 2; ++qt_) tsv[hb_][qt_] = p.x_scales[min(m0 + hb_ * 128 + wc * 32 + qt_ * 16 + l15, p.M - 1)];
                asm volatile("" : "+v"(tsv[0][0]), "+v"(tsv[0][1]), "+v"(tsv[1][0]), "+v"(tsv[1][1]));      // (in their registers before the first store: EPILOGUE_LOADS_FIRST)
            }
            // fp8: the weight-scale kind and the bias are chosen ONCE per tile (fp8_stores below), not per element: with both as runtime flags inside the store loop
            // every output carried both formulas, a select and two branches -- 153 instructions per 16-byte store where the arithmetic needs ~40
            auto out4f = [&](int hA, int hB, int pt, int qt, int m, int n, auto pc_c, auto bias_c) -> u32x2 {
                float v[4];
    #pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[hA][hB][pt][qt][e];
                if constexpr (FP8)
                {
                    // W4A8: the reference's two steps -- the GEMM stores bf16(acc * weight scale), the per-token pass rescales (+ bias); W8A8: (acc * s_c[n]) * s_m (+ bias), one rounding
                    constexpr bool PC = decltype(pc_c)::value, HB = decltype(bias_c)::value;
                    const float ts = tsv[hB][qt];
    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                    {
                        const float b = HB ? bf16_bits_to_f32(p.bias[n + e]) : 0.0f;
                        v[e] = PC ? w8a8_scale_bias(v[e], wsv[hA][pt][e], ts, HB, b) : w4a8_scale_bias(v[e], wsv[hA][pt][e], ts, HB, b);
                    }
                }
                else if (p.bias)
                {
    #pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]) + bf16_bits_to_f32(p.bias[n + e]);
                }
                if (!FP8 && p.act)
                {
    #pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(round_bf16(v[e]));
                }
                return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            };
            auto out4 = [&](int hA, int hB, int pt, int qt, int m, int n) -> u32x2 { return out4f(hA, hB, pt, qt, m, n, std::false_type{}, std::false_type{}); };      // (the bf16 forms)
            if constexpr (!FP8)
            {
                if (n0 + 256 > p.N)
                {
                    // the ragged last tile-column (N % 256 != 0): element stores under a column mask
    #pragma unroll
                    for (int hA = 0; hA < 2; ++hA)
    #pragma unroll
                        for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                            for (int pt = 0; pt < 4; ++pt)
    #pragma unroll
                                for (int qt = 0; qt < 2; ++qt)
                                {
                                    const int n = n0 + hA * 128 + wr * 64 + pt * 16 + 4 * g;
                                    const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                                    uint16_t* y = p.Y + (size_t)m * ldy + n;
    #pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (n + e < p.N && m < p.M)
                                        {
                                            float v = acc[hA][hB][pt][qt][e];
                                            if (p.bias) v = round_bf16(v) + bf16_bits_to_f32(p.bias[n + e]);
                                            if (p.act) v = gelu_tanh(round_bf16(v));
                                            y[e] = f32_to_bf16_bits(v);
                                        }
                                }
                    return;
                }
                if ((p.N & 7) != 0)
                {
                    // whole tile of an output whose row pitch is no multiple of 16 bytes: the paired 16-byte stores at 2-byte alignment
    #pragma unroll
                    for (int hA = 0; hA < 2; ++hA)
    #pragma unroll
                        for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                            for (int pp = 0; pp < 4; pp += 2)
    #pragma unroll
                                for (int qt = 0; qt < 2; ++qt)
                                {
                                    const int nb = n0 + hA * 128 + wr * 64 + pp * 16;
                                    const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                                    store_pair16<false>(p.Y + (size_t)m * ldy + nb, g, out4(hA, hB, pp, qt, m, nb + 4 * g), out4(hA, hB, pp + 1, qt, m, nb + 16 + 4 * g), m < p.M);
                                }
                    return;
                }
            }
            auto stores = [&](auto pc_c, auto bias_c) {
    #pragma unroll
                for (int hA = 0; hA < 2; ++hA)
    #pragma unroll
                    for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                        for (int pp = 0; pp < 4; pp += 2)
    #pragma unroll
                            for (int qt = 0; qt < 2; ++qt)
                            {
                                const int nb = n0 + hA * 128 + wr * 64 + pp * 16;       // first column of sub-tile pp
                                const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
    #ifdef MILA_GEMM_SKIP
                                if ((p.dbg & 8) && acc[hA][hB][pp][qt][0] != 12345.678f) continue;      // diagnostic: no epilogue stores (8)
    #endif
                                store_pair16(p.Y + (size_t)m * ldy + nb, g, out4f(hA, hB, pp, qt, m, nb + 4 * g, pc_c, bias_c), out4f(hA, hB, pp + 1, qt, m, nb + 16 + 4 * g, pc_c, bias_c), m < p.M);
                            }
            };
            if constexpr (FP8)
            {
                if (pc) { if (p.bias) stores(std::true_type{}, std::true_type{}); else stores(std::true_type{}, std::false_type{}); }
                else { if (p.bias) stores(std::false_type{}, std::true_type{}); else stores(std::false_type{}, std::false_type{}); }
            }
            else stores(std::false_type{}, std::false_type{});
        }
    };

    // ---- row-wise epilogue (plain bf16, one workgroup per tile, called behind the last barrier: LDS is free) ----
    // An output whose row pitch is no multiple of 128 bytes (GPT-2's logits: N = 50257) makes every 64-byte piece of the direct epilogue a partial cache line, and a line's
    // pieces come from two waves at different moments: they reach HBM as read-modify-writes -- 1.4 TB/s where whole lines go at 5.7 (tools/experiments/unaligned_store.hip:
    // the store WIDTH is irrelevant, 2-byte and 16-byte stores take the same time; what counts is that the pieces of a line are issued back to back).  So the tile is
    // transposed through LDS ([256][256] bf16 = the two K-tile buffers, 16-byte chunks XOR-swizzled by the row: conflict-free both ways) and every wave writes
    // whole 512-byte row segments, two rows per instruction: 3.7 TB/s on the odd pitch.
    auto rowwise_epilogue = [&]() {
        if constexpr (WABS)
        {
#pragma unroll
            for (int hA = 0; hA < 2; ++hA)
#pragma unroll
                for (int hB = 0; hB < 2; ++hB)
#pragma unroll
                    for (int pp = 0; pp < 4; pp += 2)
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt)
                        {
                            u32x2 ab[2];
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                            {
                                const int n = n0 + hA * 128 + wr * 64 + (pp + h) * 16 + 4 * g;
                                float v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                {
                                    v[e] = acc[hA][hB][pp + h][qt][e];
                                    if (p.bias) v[e] = round_bf16(v[e]) + bf16_bits_to_f32(p.bias[min(n + e, p.N - 1)]);
                                    if (p.act) v[e] = gelu_tanh(round_bf16(v[e]));
                                }
                                ab[h] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                            }
                            // the exchange of store_pair16: an even-g lane ends with columns 4 g .. 4 g + 7 of sub-tile pp, an odd-g lane with 4 (g - 1) .. + 7 of sub-tile pp + 1
                            const auto r0 = __builtin_amdgcn_permlane16_swap(ab[0][0], ab[1][0], false, false);
                            const auto r1 = __builtin_amdgcn_permlane16_swap(ab[0][1], ab[1][1], false, false);
                            const int lr = hB * 128 + wc * 32 + qt * 16 + l15;
                            const int chunk = (hA * 128 + wr * 64 + pp * 16 + (g & 1) * 16 + 4 * (g & ~1)) >> 3;
                            *reinterpret_cast<u32x4*>(smem + lr * 512 + ((chunk ^ (lr & 31)) << 4)) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                        }
            __syncthreads();
            const int half = lane >> 5, chunk = lane & 31;
            const int n = n0 + chunk * 8;
#pragma unroll 4
            for (int i = 0; i < 16; ++i)
            {
                const int lr = wave * 32 + i * 2 + half;
                const int m = m0 + lr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + lr * 512 + ((chunk ^ (lr & 31)) << 4));
                if (m >= p.M) continue;
                uint16_t* y = p.Y + (size_t)m * ldy + n;
                if (n + 8 <= p.N) st16_a2(y, v);
                else
                {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < p.N) y[e] = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
                }
            }
        }
    };

    // ---- prologue: K-tile 0 complete, W0 / X1 of K-tile 1 in flight (PP == 2: W0 / X0 / X1 of K-tile 1) ----
    stage(0, false, 0); stage(0, true, 0); stage(0, true, 1); stage(0, false, 1);
    if (nk > 1) { stage(1, false, 0); if constexpr (P2) stage(1, true, 0); stage(1, true, 1); }
    if (nk > 1) { if constexpr (P2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if constexpr (P2)
    {
        auto mma_end = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        if constexpr (PP == 3) { if (wr == 1) __builtin_amdgcn_s_setprio(1); }
        if (wr == 1) __builtin_amdgcn_s_barrier();
#ifdef MILA_GEMM_SKIP
        const bool dns = p.dbg & 1, dnr = p.dbg & 2, dnm = p.dbg & 4;
        if (dnr) { load_a(0, 0); load_b(0, 0); load_b(0, 1); }
#else
        constexpr bool dns = false, dnr = false, dnm = false;
#endif
        // PERSISTENT: the K loop runs on across the workgroup's tiles.  The last two K-tiles of a tile stage the first two of the next one in the slots that used
        // to stay empty (no prologue bubble), the epilogue's stores are issued and NOT waited for -- they drain under the next tile's first K-tile, whose two phases
        // allow kEpilogueStores more operations in flight (vmcnt counts loads and stores together, in order) -- so a tile costs its K loop and the issue of its
        // stores, not a prologue's memory latency plus the write burst of every CU at once.
        // (kEpilogueStores = 8 with the GeGLU epilogue, 16 with the plain one: the vmcnt(8 + kEpilogueStores) below)
        const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
        bool prev_edge = false;
        for (int j = 0; j < my_tiles; ++j)
        {
            const bool has_next = j + 1 < my_tiles;
            if (has_next)
            {
                tile_origin(blockIdx.x + (j + 1) * gridDim.x, xm0, xn0, xwrow1);
                x_offsets(voffXn, xm0);
                if constexpr (WABS) w_offsets(voffWn, xn0, xwrow1);
            }
            // INTERIOR K-tiles (round 4): 1 <= t <= nk - 3 -- every staging request belongs to THIS tile, the waits are the steady ones, nothing of the previous
            // tile's stores is counted -- run in pairs with the buffer parity as a template constant: no choice of wait, no next-tile test, no parity arithmetic
            // on the LDS addresses.  (SQ counters: 60 of a K-tile's ~165 instructions per wave were scalar, 12 more vector adds for the parity; a wave issues one
            // instruction per four cycles and a phase's reads + requests must fit under the partner wave's 512 MFMA cycles.)  Same instructions on the data: same bits.
            const int sw0 = WABS ? 0 : n0 * rowbytes, sw1 = WABS ? 0 : wrow1 * rowbytes;      // (relative W offsets: the half-tile's first row rides in the scalar offset)
            auto stage_in = [&](auto par_c, int koff, bool isX, int half) {
                constexpr int PARB = decltype(par_c)::value;
                unsigned char* dst_half = smem + PARB * kBufBytes + half_off(isX, half);
                int soff = (isX || WABS) ? koff : (half ? sw1 : sw0) + koff;
                // (fp8 GeGLU: its scalar registers are full, the allocator had moved the half-tile's row offset into a VECTOR register, and every request of an interior
                // K-tile became a readfirstlane WATERFALL LOOP around a scratch reload -- 49 spilled registers, fc_gate_up fp8 234 -> 507 us; pinned scalar here, the other
                // modes stay as they are: an explicit readfirstlane costs them 1-5 %)
                if constexpr (MODE == G_FP8_GEGLU) soff = __builtin_amdgcn_readfirstlane(soff);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                {
                    const int chunk = i * 8 + wave;
                    if (isX) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(dst_half + chunk * 1024), 16, voffX[half][i], soff, 0, 0);
                    else if constexpr (WABS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(dst_half + chunk * 1024), 16, voffW[half][i], soff, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(dst_half + chunk * 1024), 16, voffW[0][i], soff, 0, 0);
                }
            };
            auto ktile_in = [&](int t, auto par_c) {
                constexpr int PARB = decltype(par_c)::value;
                constexpr std::integral_constant<int, PARB> same{};
                constexpr std::integral_constant<int, PARB ^ 1> other{};
                auto reads_done = [&]() {
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                };
                const int k1 = (t + 1) * 128, k2 = k1 + 128;
                // phase A: quadrants (0,0) (0,1)
                stage_in(other, k1, false, 1);
                load_a(PARB, 0); load_b(PARB, 0); load_b(PARB, 1);
                reads_done();
                mma(0, 0); mma(0, 1);
                mma_end();
                // phase B: quadrants (1,0) (1,1)
                stage_in(same, k2, false, 0); stage_in(same, k2, true, 0); stage_in(same, k2, true, 1);
                load_a(PARB, 1);
                reads_done();
                mma(1, 0); mma(1, 1);
                mma_end();
            };
            auto ktile_general = [&](int t) {
                const bool steady = (t + 2 < nk) || has_next;
                // the previous tile's stores are still in flight -- kEpilogueStores of them, counted; a tile on the ragged edge of Y issues another number (masked rows:
                // possibly fewer), so behind such a tile the wait is the plain one (its stores are retired with it)
                const bool post = j > 0 && t == 0 && !prev_edge;
                auto reads_done = [&]() {
                    if (!steady) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (post) { if constexpr (GEGLU) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                };
                // phase A: quadrants (0,0) (0,1)
                if ((t + 1 < nk || has_next) && !dns) stage(t + 1, false, 1);
                if (!dnr) { load_a(t, 0); load_b(t, 0); load_b(t, 1); }
                reads_done();
                if (!dnm) { mma(0, 0); mma(0, 1); }
                mma_end();
                // phase B: quadrants (1,0) (1,1)
                if ((t + 2 < nk || has_next) && !dns) { stage(t + 2, false, 0); stage(t + 2, true, 0); stage(t + 2, true, 1); }
                if (!dnr) load_a(t, 1);
                reads_done();
                if (!dnm) { mma(1, 0); mma(1, 1); }
                mma_end();
            };
            {
                int t = 0;
                ktile_general(t++);                                   // K-tile 0: behind the previous tile's stores
#ifndef MILA_GEMM_SKIP
                // (the fp8 GeGLU mode takes them since its staging offsets are pinned scalar -- stage_in -- and its epilogue re-derives the lane id: 216 -> 202 us)
                for (; t + 4 <= nk; t += 2)                          // (t odd here)
                {
                    ktile_in(t, std::integral_constant<int, 1>{});
                    ktile_in(t + 1, std::integral_constant<int, 0>{});
                }
#endif
                for (; t < nk; ++t) ktile_general(t);                // the last K-tiles: the next tile's first requests, the un-steady waits
            }
            if (!(WABS && p.rowwise)) epilogue();
            prev_edge = m0 + 256 > p.M || n0 + (GEGLU ? 128 : 256) > p.N;
            if (has_next)
            {
                m0 = xm0; n0 = xn0; wrow1 = xwrow1;
#pragma unroll
                for (int h_ = 0; h_ < 2; ++h_)
#pragma unroll
                    for (int i_ = 0; i_ < 2; ++i_)
                    {
                        voffX[h_][i_] = voffXn[h_][i_];
                        if constexpr (WABS) voffW[h_][i_] = voffWn[h_][i_];
                    }
                zero_acc();
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        if (WABS && p.rowwise) rowwise_epilogue();      // one tile per workgroup (the launcher's rule): every wave is behind its last fragment read, nothing is staged
    }
    else if constexpr (PP == 1)
    {
        // reads of a phase end with both waits (own staging share of phase p - 3, own fragment reads), then barrier A
        auto reads_end = [&](bool steady) {
            if (steady) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma_end = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        if (wr == 1) __builtin_amdgcn_s_barrier();          // the stagger: group 1 starts one barrier late ...
        for (int t = 0; t < nk; ++t)
        {
            const bool steady = t + 2 < nk;
            if (t + 1 < nk) stage(t + 1, true, 0);          // ph0: quadrant (0,0)
            load_b(t, 0);
            load_a(t, 0);
            reads_end(steady);
            mma(0, 0);
            mma_end();
            if (t + 1 < nk) stage(t + 1, false, 1);         // ph1: quadrant (0,1)
            load_b(t, 1);
            reads_end(steady);
            mma(0, 1);
            mma_end();
            if (t + 2 < nk) stage(t + 2, false, 0);         // ph2: quadrant (1,1)
            load_a(t, 1);
            reads_end(steady);
            mma(1, 1);
            mma_end();
            if (t + 2 < nk) stage(t + 2, true, 1);          // ph3: quadrant (1,0)
            load_b(t, 0);
            reads_end(steady);
            mma(1, 0);
            mma_end();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();          // ... and group 0 waits one barrier at the end: equal counts
    }
    else
    {
    for (int t = 0; t < nk; ++t)
        {
            const bool steady = t + 2 < nk;                    // the full issue sequence is still running
            // ph0: quadrant (0,0)
            if (t + 1 < nk) stage(t + 1, true, 0);
            load_b(t, 0);
            load_a(t, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(0, 0);
            phase_end(steady);
            // ph1: quadrant (0,1)
            if (t + 1 < nk) stage(t + 1, false, 1);
            load_b(t, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(0, 1);
            phase_end(steady);
            // ph2: quadrant (1,1)
            if (t + 2 < nk) stage(t + 2, false, 0);
            load_a(t, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(1, 1);
            phase_end(steady);
            // ph3: quadrant (1,0)
            if (t + 2 < nk) stage(t + 2, true, 1);
            load_b(t, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(1, 0);
            phase_end(steady);
        }

    }

    if constexpr (!P2) epilogue();
}

// ---- 256 (M) x 128 (N) x 64 tile: shapes whose N leaves the 256 x 256 grid half empty (Gemma o_proj / fc_down, N = 3840:
// 8 x 15 = 120 tiles) get 8 x 30 = 240 tiles, one per CU.  Same 8 waves (2 over the 128 W rows x 4 over 32 X rows of each
// X half), same swizzled half-tiles and fragment reads; the K pipeline is a 3-stage ring of {W, X0, X1} half-tiles
// (3 x 48 KB): K-tile t + 2 is requested at the top of K-tile t, ONE s_waitcnt vmcnt(6) + barrier per K-tile retires
// K-tile t + 1 while t + 2 stays in flight.
constexpr int kStage3Bytes = 3 * kHalfBytes;   // W, X0, X1
#ifndef MILA_RING_GLOBAL_LOADS
#define MILA_RING_GLOBAL_LOADS 1
#endif
constexpr bool kRingGlobalLoads = MILA_RING_GLOBAL_LOADS;      // how the 256 x 128 ring stages (see its stage())

// GEGLU: the tile's 128 W rows are, per wave half wr, 32 gate rows then the matching 32 up rows (64 output columns per tile:
// n0 = 64 tn), so a lane's sub-tiles pt = 0,1 / 2,3 hold gate / up of the same outputs.
// PP: the same stagger as gemm256_kernel<.., PP>: [stage K-tile t + 2, all 16 fragment reads of K-tile t, waits] | A | [32 MFMAs] | B |,
// waves 4-7 one barrier behind waves 0-3.  RAW: K-tile t + 1 is waited for (vmcnt(6)) at the end of the reads of phase t and read in
// phase t + 1; WAR: ring slot (t + 2) % 3 held K-tile t - 1, whose reads every wave retired before barrier A(t - 1), and group 0
// restages it behind B(t - 1) = group 1's A(t - 1).
// WALK: the persistent tile walk is compiled in (one workgroup per CU); without it the kernel is the one-tile form, whose K loop carries none of the walk's selects
// (the walk's bookkeeping in the K loop -- next-tile offsets, the flat ring index -- cost the one-round shapes 15 %: o_proj + fc_down 130 -> 152 us average)
// SPLITK (plain one-tile forms, bf16 and fp8): the grid is p.splitk copies of the tile list; copy ks takes K-tiles [ks nk / S, (ks + 1) nk / S) of its tile and writes its fp32
// accumulators to p.partials[ks] -- few-row prompts and remainders, whose tile list covers a fraction of the CUs, then spread each tile's K over the idle ones
template <bool FP8, bool GEGLU, int PP, bool WALK, bool SPLITK = false>      // PP: 0 lockstep, 1 staggered groups, 2 staggered with a static priority for waves 4-7
__global__ __launch_bounds__(512) void gemm256x128_kernel(const Gemm256Params p)
{
    static_assert(!SPLITK || (!GEGLU && !WALK), "split-K: the plain one-tile forms (bf16, fp8)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = FP8 ? 1 : 2;
    constexpr int KT = 128 / ES;

    // Workgroup id -> tiles id, id + gridDim.x, ...: the launcher starts one workgroup per tile, or (staggered schedules, PERSISTENT) one per CU walking its tiles --
    // the XCD remap is taken over the whole tile list, so a workgroup's later tiles are the ones the dispatcher would have handed to its XCD anyway
    const int ntiles = p.tiles_m * p.tiles_n;
    auto tile_origin = [&](int idv, int& m0_, int& n0_) {
        const int xcd = idv & 7, qd = ntiles >> 3, rem = ntiles & 7;
        const int tile = ((xcd < rem) ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (idv >> 3);
        int tm, tn;
        grouped_tile(tile, p.tiles_m, p.tiles_n, tm, tn);
        m0_ = tm * 256;
        n0_ = tn * (GEGLU ? 64 : 128);
    };
    int m0, n0, xm0 = 0, xn0 = 0;       // this tile and (persistent form) the next one
    int ks = 0;
    if constexpr (SPLITK) { ks = (int)blockIdx.x / ntiles; tile_origin((int)blockIdx.x - ks * ntiles, m0, n0); }
    else tile_origin(blockIdx.x, m0, n0);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    const int K = p.K, nk_all = K / KT;
    const int ldy = p.ldy ? p.ldy : p.N;                   // output row pitch
    // this workgroup's K-tiles and the byte offset of its first one in a row
    const int kt_first = SPLITK ? ks * nk_all / max(p.splitk, 1) : 0;
    const int nk = SPLITK ? (ks + 1) * nk_all / max(p.splitk, 1) - kt_first : nk_all;
    const int kbyte0 = kt_first * 128;
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(p.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);

    const int srow = lane >> 3, sslot = lane & 7;
    // which: 0 = W rows n0 .., 1 = X rows m0 .., 2 = X rows m0 + 128 ..   `buffer_load ... lds`, per-lane offsets computed once per tile (see gemm256_kernel)
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Xb), 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Wb), 0, 0x7fffffff, 0x00020000);
    const int rowbytes = K * ES;
    int voff[3][2], voffn[3][2];
    auto offsets = [&](int (&vo)[3][2], int mm0, int nn0) {
#pragma unroll
        for (int which = 0; which < 3; ++which)
#pragma unroll
            for (int i = 0; i < 2; ++i)
            {
                const int row = (i * 8 + wave) * 8 + srow;
                const int lslot = sslot ^ ((row >> 1) & 7);
                const int kslot = FP8 ? (((lslot & 3) << 1) | (lslot >> 2)) : lslot;      // as in gemm256_kernel
                int grow;
                if (which == 0)
                {
                    if constexpr (GEGLU) { const int q = row & 63; grow = (q < 32 ? 0 : p.N - 32) + nn0 + (row >> 6) * 32 + q; }   // N = F: up rows start at F
                    else grow = min(nn0 + row, p.N - 1);      // W rows past N (a ragged last column tile) re-read row N - 1, never stored
                }
                else grow = min(mm0 + (which - 1) * 128 + row, p.M - 1);      // X rows past M (ragged last tile-row) re-read row M - 1, never stored
                vo[which][i] = grow * rowbytes + kslot * 16 + (SPLITK ? kbyte0 : 0);
            }
    };
    offsets(voff, m0, n0);
    // K-tile `kt` of this tile (kt < nk) or K-tile kt - nk of the workgroup's next tile, into ring slot `flat` % 3 (flat = K-tiles since the workgroup started)
    auto stage = [&](int flat, int kt, int which) {
        const bool nxt = WALK && kt >= nk;
        unsigned char* dst_half = smem + (flat % 3) * kStage3Bytes + which * kHalfBytes;
        const int soff = __builtin_amdgcn_readfirstlane((nxt ? kt - nk : kt) * 128);
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
            const int chunk = i * 8 + wave;
            if constexpr (kRingGlobalLoads)
            {
                // `global_load_lds` on base + (offset computed once per tile) + K-tile: measured against the buffer form on one box with the weights rotating through HBM
                // (tools/experiments/ab_r02_lib.py, profiles/r03_gemm_bisect.txt) -- in THIS kernel the buffer form is 4-5 % slower (o_proj 57.7 vs 60.7 us, fc_down 195.5 vs
                // 203.9; round 2's per-request address arithmetic: 57.1 / 196.7), in the 256 x 256 kernel it is 2 % faster
                const unsigned char* src = (which == 0 ? Wb : Xb) + (size_t)(uint32_t)(nxt ? voffn[which][i] : voff[which][i]) + (size_t)(uint32_t)soff;
                __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(dst_half + chunk * 1024), 16, 0, 0);
            }
            else if (which == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(dst_half + chunk * 1024), 16, nxt ? voffn[0][i] : voff[0][i], soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(dst_half + chunk * 1024), 16, nxt ? voffn[which][i] : voff[which][i], soff, 0, 0);
        }
    };
    auto stage_all = [&](int flat, int kt) { stage(flat, kt, 0); stage(flat, kt, 1); stage(flat, kt, 2); };

    f32x4 acc[2][4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[b][c][d] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    zero_acc();

    s16x8 fa[4][2], fb[2][2], fb1[PP ? 2 : 1][2];          // PP keeps the fragments of both X halves live
    auto frag_slot = [&](int ks) { return ks * 4 + g; };
    auto load_a = [&](int kt) {
        const unsigned char* hb = smem + (kt % 3) * kStage3Bytes;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
        {
            const int r = wr * 64 + pt * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fa[pt][ks] = *reinterpret_cast<const s16x8*>(hb + r * 128 + ((frag_slot(ks) ^ ((r >> 1) & 7)) << 4));
        }
    };
    auto load_b = [&](int kt, int hB) {
        const unsigned char* hb = smem + (kt % 3) * kStage3Bytes + (1 + hB) * kHalfBytes;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
        {
            const int r = wc * 32 + qt * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
            {
                const s16x8 v = *reinterpret_cast<const s16x8*>(hb + r * 128 + ((frag_slot(ks) ^ ((r >> 1) & 7)) << 4));
                if (PP && hB == 1) fb1[PP ? qt : 0][ks] = v;
                else fb[qt][ks] = v;
            }
        }
    };
    auto mma = [&](int hB) {
        s16x8 (&fbx)[2][2] = *((PP && hB == 1) ? reinterpret_cast<s16x8 (*)[2][2]>(&fb1) : &fb);
        if constexpr (PP != 2) __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8)
        {
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                {
                    struct Pair { s16x8 lo, hi; };
                    const i32x8 a8 = __builtin_bit_cast(i32x8, (Pair{fa[pt][0], fa[pt][1]}));
                    const i32x8 b8 = __builtin_bit_cast(i32x8, (Pair{fbx[qt][0], fbx[qt][1]}));
                    acc[hB][pt][qt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[hB][pt][qt], 0, 0, 0, 127, 0, 127);
                }
            // (the products pinned in front of their barrier, as in gemm256_kernel: the compiler sinks the pure scaled-MFMA calls across barriers)
            asm volatile("" : "+v"(acc[hB][0][0]), "+v"(acc[hB][0][1]), "+v"(acc[hB][1][0]), "+v"(acc[hB][1][1]),
                              "+v"(acc[hB][2][0]), "+v"(acc[hB][2][1]), "+v"(acc[hB][3][0]), "+v"(acc[hB][3][1]));
        }
        else
        {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        acc[hB][pt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[pt][ks]), __builtin_bit_cast(bf16x8, fbx[qt][ks]), acc[hB][pt][qt], 0, 0, 0);
        }
        if constexpr (PP != 2) __builtin_amdgcn_s_setprio(0);
    };

    auto epilogue = [&]() {
        if constexpr (SPLITK)
        {
            // the raw fp32 accumulators of this K range: 16 bytes (4 columns) per lane and sub-tile, 64 contiguous bytes per row and instruction
            float* P = p.partials + (size_t)ks * p.M * p.N;
    #pragma unroll
            for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                for (int pt = 0; pt < 4; ++pt)
    #pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                    {
                        const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                        if (m < p.M) *reinterpret_cast<f32x4*>(P + (size_t)m * p.N + n0 + wr * 64 + pt * 16 + 4 * g) = acc[hB][pt][qt];
                    }
            return;
        }
        if constexpr (GEGLU)
        {
            // fp8 epilogue scales, fetched ONCE before the stores: read inside the store loop they are re-fetched after every store (the compiler cannot
            // prove that Y does not alias them) -- a dependent global load in front of each of the 16 stores, 7-11 us per tile
            float tsv[2][2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
            f32x4 wsg[2], wsu[2];      // weight scales of this lane's gate / up columns per sub-tile pt (W4A8: the per-tensor scalar in every slot)
            const bool pc = FP8 && p.w_pc;
            if constexpr (FP8)
            {
                if (pc)
                {
            #pragma unroll
                    for (int pt_ = 0; pt_ < 2; ++pt_)
                    {
                        const int n_ = n0 + wr * 32 + pt_ * 16 + 4 * g;
                        wsg[pt_] = *reinterpret_cast<const f32x4*>(p.w_scale + n_);
                        wsu[pt_] = *reinterpret_cast<const f32x4*>(p.w_scale + p.N + n_);
                    }
                    asm volatile("" : "+v"(wsg[0]), "+v"(wsg[1]), "+v"(wsu[0]), "+v"(wsu[1]));
                }
                else
                {
                    const float ws_ = *p.w_scale;
                    wsg[0] = wsg[1] = wsu[0] = wsu[1] = f32x4{ws_, ws_, ws_, ws_};
                }
            #pragma unroll
                for (int hb_ = 0; hb_ < 2; ++hb_)
            #pragma unroll
                    for (int qt_ = 0; qt_ < 2; ++qt_) tsv[hb_][qt_] = p.x_scales[min(m0 + hb_ * 128 + wc * 32 + qt_ * 16 + l15, p.M - 1)];
                asm volatile("" : "+v"(tsv[0][0]), "+v"(tsv[0][1]), "+v"(tsv[1][0]), "+v"(tsv[1][1]));      // (in their registers before the first store: EPILOGUE_LOADS_FIRST)
            }
            auto out4 = [&](int hB, int pt, int qt, int m, auto pc_c) -> u32x2 {
                float v[4];
                if constexpr (FP8)
                {
                    constexpr bool PC = decltype(pc_c)::value;
                    const float ts = tsv[hB][qt];
    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = gelu_tanh(fp8_linear_out(PC, acc[hB][pt][qt][e], wsg[pt][e], ts)) * fp8_linear_out(PC, acc[hB][pt + 2][qt][e], wsu[pt][e], ts);
                }
                else
                {
    #pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(round_bf16(acc[hB][pt][qt][e])) * round_bf16(acc[hB][pt + 2][qt][e]);
                }
                return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            };
            auto stores = [&](auto pc_c) {
    #pragma unroll
                for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                    {
                        const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                        store_pair16(p.Y + (size_t)m * ldy + n0 + wr * 32, g, out4(hB, 0, qt, m, pc_c), out4(hB, 1, qt, m, pc_c), m < p.M);
                    }
            };
            if (pc) stores(std::true_type{}); else stores(std::false_type{});
            return;
        }
        // fp8 epilogue scales, fetched ONCE before the stores: read inside the store loop they are re-fetched after every store (the compiler cannot
        // prove that Y does not alias them) -- a dependent global load in front of each of the 16 stores, 7-11 us per tile
        float tsv[2][2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
        f32x4 wsv[4];      // weight scales of this lane's columns per sub-tile (W4A8: the per-tensor scalar in every slot)
        const bool pc = FP8 && p.w_pc;
        if constexpr (FP8)
        {
            if (pc)
            {
        #pragma unroll
                for (int pt_ = 0; pt_ < 4; ++pt_) wsv[pt_] = *reinterpret_cast<const f32x4*>(p.w_scale + n0 + wr * 64 + pt_ * 16 + 4 * g);
                asm volatile("" : "+v"(wsv[0]), "+v"(wsv[1]), "+v"(wsv[2]), "+v"(wsv[3]));
            }
            else
            {
                const float ws_ = *p.w_scale;
        #pragma unroll
                for (int pt_ = 0; pt_ < 4; ++pt_) wsv[pt_] = f32x4{ws_, ws_, ws_, ws_};
            }
        #pragma unroll
            for (int hb_ = 0; hb_ < 2; ++hb_)
        #pragma unroll
                for (int qt_ = 0; qt_ < 2; ++qt_) tsv[hb_][qt_] = p.x_scales[min(m0 + hb_ * 128 + wc * 32 + qt_ * 16 + l15, p.M - 1)];
            asm volatile("" : "+v"(tsv[0][0]), "+v"(tsv[0][1]), "+v"(tsv[1][0]), "+v"(tsv[1][1]));      // (in their registers before the first store: EPILOGUE_LOADS_FIRST)
        }
        // the bias of this lane's 16 columns (4 sub-tiles x 4), fetched ONCE before the stores like the scales above: read inside the store loop every value is fetched again
        // after every store (the compiler cannot prove that Y does not alias the bias) -- 64 dependent 2-byte loads per lane and tile, 15 % of GPT-2's biased projections
        // (same-box A/B, profiles/r03_epilogue_ab.txt run 4: qkv 53.1 -> 45.5 us, fc_1 57.9 -> 48.8).  Columns past N (a ragged last tile-column) read column N - 1; they
        // are never stored.  (The 256 x 256 plain kernel keeps the loads in its loop: hoisted there, even one W half at a time, they cost 138 spilled registers.)
        float bv[4][4] = {};
        if (p.bias)
        {
    #pragma unroll
            for (int pt_ = 0; pt_ < 4; ++pt_)
    #pragma unroll
                for (int e = 0; e < 4; ++e) bv[pt_][e] = bf16_bits_to_f32(p.bias[min(n0 + wr * 64 + pt_ * 16 + 4 * g + e, p.N - 1)]);
            asm volatile("" : "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[0][2]), "+v"(bv[0][3]), "+v"(bv[1][0]), "+v"(bv[1][1]), "+v"(bv[1][2]), "+v"(bv[1][3]),
                              "+v"(bv[2][0]), "+v"(bv[2][1]), "+v"(bv[2][2]), "+v"(bv[2][3]), "+v"(bv[3][0]), "+v"(bv[3][1]), "+v"(bv[3][2]), "+v"(bv[3][3]));
        }
        // (fp8: weight-scale kind and bias chosen once per tile, as in gemm256_kernel's epilogue)
        auto out4f = [&](int hB, int pt, int qt, int m, int n, auto pc_c, auto bias_c) -> u32x2 {
            float v[4];
    #pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[hB][pt][qt][e];
            if constexpr (FP8)
            {
                constexpr bool PC = decltype(pc_c)::value, HB = decltype(bias_c)::value;
                const float ts = tsv[hB][qt];
    #pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = PC ? w8a8_scale_bias(v[e], wsv[pt][e], ts, HB, bv[pt][e]) : w4a8_scale_bias(v[e], wsv[pt][e], ts, HB, bv[pt][e]);
            }
            else if (p.bias)
            {
    #pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]) + bv[pt][e];
            }
            if (!FP8 && p.act)
            {
    #pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(round_bf16(v[e]));
            }
            return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        };
        auto out4 = [&](int hB, int pt, int qt, int m, int n) -> u32x2 { return out4f(hB, pt, qt, m, n, std::false_type{}, std::false_type{}); };      // (the bf16 forms)
        if (!FP8 && (p.N & 7) != 0 && n0 + 128 <= p.N)
        {
            // an output row pitch that is not a multiple of 16 bytes (GPT-2's lm_head: N = 50257), whole tile: the paired 16-byte stores at 2-byte alignment
    #pragma unroll
            for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                for (int pp = 0; pp < 4; pp += 2)
    #pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                    {
                        const int nb = n0 + wr * 64 + pp * 16;
                        const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                        store_pair16<false>(p.Y + (size_t)m * ldy + nb, g, out4(hB, pp, qt, m, nb + 4 * g), out4(hB, pp + 1, qt, m, nb + 16 + 4 * g), m < p.M);
                    }
            return;
        }
        if (!FP8 && n0 + 128 > p.N)
        {
            // the ragged last column tile: element stores under a column mask
    #pragma unroll
            for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                for (int pt = 0; pt < 4; ++pt)
    #pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                    {
                        const int n = n0 + wr * 64 + pt * 16 + 4 * g;
                        const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                        if (m >= p.M) continue;
                        uint16_t* y = p.Y + (size_t)m * ldy + n;
    #pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N)
                            {
                                float v = acc[hB][pt][qt][e];
                                if (p.bias) v = round_bf16(v) + bf16_bits_to_f32(p.bias[n + e]);
                                if (p.act) v = gelu_tanh(round_bf16(v));
                                y[e] = f32_to_bf16_bits(v);
                            }
                    }
            return;
        }
        auto stores = [&](auto pc_c, auto bias_c) {
    #pragma unroll
            for (int hB = 0; hB < 2; ++hB)
    #pragma unroll
                for (int pp = 0; pp < 4; pp += 2)
    #pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                    {
                        const int nb = n0 + wr * 64 + pp * 16;
                        const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                        store_pair16(p.Y + (size_t)m * ldy + nb, g, out4f(hB, pp, qt, m, nb + 4 * g, pc_c, bias_c), out4f(hB, pp + 1, qt, m, nb + 16 + 4 * g, pc_c, bias_c), m < p.M);
                    }
        };
        if constexpr (FP8)
        {
            if (pc) { if (p.bias) stores(std::true_type{}, std::true_type{}); else stores(std::true_type{}, std::false_type{}); }
            else { if (p.bias) stores(std::false_type{}, std::true_type{}); else stores(std::false_type{}, std::false_type{}); }
        }
        else stores(std::false_type{}, std::false_type{});
    };

    stage_all(0, 0);
    if (nk > 1) stage_all(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if constexpr (PP != 0)
    {
        // PERSISTENT (round 3): the K loop runs on across the workgroup's tiles.  The ring keeps turning (slot = K-tiles since the start % 3), the last two K-tiles
        // of a tile request the first two of the next one, and the epilogue's stores are issued and NOT waited for: they drain under the next tile's first K-tile,
        // whose wait allows kStores more operations in flight (vmcnt counts loads and stores together, in order).  A tile then costs its K loop and the issue of its
        // stores -- not a launch, a first-load latency and a store drain per tile, which at K = 768 (GPT-2: 12 K-tiles) were as long as the K loop itself.
        constexpr int kStores = GEGLU ? 4 : 8;            // 16-byte stores per lane of a whole tile's epilogue
        if constexpr (PP == 2) { if (wr == 1) __builtin_amdgcn_s_setprio(1); }
        if (wr == 1) __builtin_amdgcn_s_barrier();
        const int my_tiles = WALK ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 1;
        int base = 0;                                     // K-tiles before this tile
        bool prev_edge = false;
        for (int j = 0; j < my_tiles; ++j)
        {
            const bool has_next = WALK && j + 1 < my_tiles;
            if (has_next) { tile_origin(blockIdx.x + (j + 1) * gridDim.x, xm0, xn0); offsets(voffn, xm0, xn0); }
            auto ktile_general = [&](int t) {
                const bool more = t + 2 < nk || (has_next && t + 2 - nk < nk);
                // the previous tile's stores sit between K-tile 1's requests and K-tile 2's: counted, unless that tile lay on a ragged edge of Y (another number of
                // stores, possibly fewer: the plain wait retires them all)
                const bool post = WALK && j > 0 && t == 0 && !prev_edge;
                if (more) stage_all(base + t + 2, t + 2);
                load_a(base + t);
                load_b(base + t, 0);
                load_b(base + t, 1);
                if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (post) { if constexpr (GEGLU) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                mma(0);
                mma(1);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            };
            // INTERIOR K-tiles (round 4, as in gemm256_kernel): 1 <= t <= nk - 3 -- the request is this tile's K-tile t + 2, the wait the steady one -- run in triples
            // with the ring slot as a template constant (three rotations of the loop, chosen once per tile by the slot of its K-tile 1): no modulo, no next-tile
            // test, no choice of wait per K-tile.  Same instructions on the data: same bits.
            auto ktile_in = [&](int t, auto slot_c) {
                constexpr int SLOT = decltype(slot_c)::value % 3, NEXT = (SLOT + 2) % 3;
                // the fragment reads go FIRST, the six staging requests behind them: an LDS-DMA request costs the issuing wave 100-185 cycles while the phase's ds_reads are
                // queued behind it and 25-60 once they are out (MI355X_MICROARCH.md, LDS-DMA piece issue cost); this phase must fit under the partner group's 512 MFMA
                // cycles and does not (SQ counters: the matrix cores are busy 65 % of this kernel's cycles, 79-85 % of the 256 x 256 kernel's, which stages a third fewer
                // bytes per FLOP).  A/B of two builds: o_proj 51.1 -> 50.1, fc_down 171.3 -> 167.2, fp8 o_proj / fc_down 32.3 / 97.5 -> 31.2 / 94.1 us, same bits.
                // (Two of the six requests between the MFMA blocks instead: +2.4 %.  The buffer_load form of the requests: +0.5-1 %.  Without any request -- timing
                // only -- fc_down takes 143 us, without the wait for them the same 167: the requests' issue, not their latency, is what the phase pays.)
                load_a(SLOT);
                load_b(SLOT, 0);
                load_b(SLOT, 1);
                stage(NEXT, t + 2, 0); stage(NEXT, t + 2, 1); stage(NEXT, t + 2, 2);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                mma(0);
                mma(1);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            };
            auto interior = [&](int& t, auto s0c) {
                constexpr int S0 = decltype(s0c)::value;
                for (; t + 5 <= nk; t += 3)
                {
                    ktile_in(t, std::integral_constant<int, S0>{});
                    ktile_in(t + 1, std::integral_constant<int, S0 + 1>{});
                    ktile_in(t + 2, std::integral_constant<int, S0 + 2>{});
                }
            };
            {
                int t = 0;
                ktile_general(t++);
                if constexpr (!(FP8 && WALK))                        // (the walking fp8 forms spill with the extra loop bodies)
                {
                    const int s1 = (base + 1) % 3;                   // ring slot of K-tile 1
                    if (s1 == 0) interior(t, std::integral_constant<int, 0>{});
                    else if (s1 == 1) interior(t, std::integral_constant<int, 1>{});
                    else interior(t, std::integral_constant<int, 2>{});
                }
                for (; t < nk; ++t) ktile_general(t);
            }
            epilogue();
            prev_edge = m0 + 256 > p.M || n0 + (GEGLU ? 64 : 128) > p.N;
            if (has_next)
            {
                m0 = xm0; n0 = xn0; base += nk;
#pragma unroll
                for (int w_ = 0; w_ < 3; ++w_)
#pragma unroll
                    for (int i_ = 0; i_ < 2; ++i_) voff[w_][i_] = voffn[w_][i_];
                zero_acc();
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        static_assert(kStores == (GEGLU ? 4 : 8), "the post waits above are 6 + kStores");
    }
    else
    {
        for (int t = 0; t < nk; ++t)
        {
            const bool more = t + 2 < nk;
            if (more) stage_all(t + 2, t + 2);
            load_a(t);
            load_b(t, 0);
            mma(0);
            load_b(t, 1);
            mma(1);
            if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        epilogue();
    }
}

// the LDS-DMA kernels address their operands through 32-bit buffer offsets (bytes, signed int arithmetic): both tensors must stay below 2 GiB
static bool lds_dma_addressable(int M, int K, int w_rows) { return (int64_t)M * K * 2 < 0x7fffffffll && (int64_t)w_rows * K * 2 < 0x7fffffffll; }

int g_ldsdma_loose_tiles = 30;     // the 256 x 128 ring applies from this many tiles on whatever the fill of its last round (tuning "gemm.ldsdma_loose_tiles"; 0 = the fill rule only).
MILA_TUNE("gemm.ldsdma_loose_tiles", g_ldsdma_loose_tiles);
                                   // Measured with tools/experiments/bf16_ragged_rules.py (profiles/r03_bf16_ragged.txt): even a 30-tile ring beats the register-staged 128-tile kernel --
                                   // bf16-policy prefill of 100 / 300 / 511 / 1000 tokens 23.6 / 25.1 / 26.4 / 31.6 -> 16.3 / 17.9 / 21.4 / 26.7 ms
// taken when the 256 x 256 grid does not apply and the 256 x 128 grid fills most of one round of CUs (or several)
bool gemm256x128_applicable(int M, int K, int N)
{
    if (M <= 0 || N % 128 != 0 || K % 64 != 0 || !lds_dma_addressable(M, K, N)) return false;
    const int tiles = ((M + 255) / 256) * (N / 128);      // a ragged last tile-row counts (and costs) whole: rows past M re-read row M - 1, their stores are masked
    const int rounds = (tiles + kNumCU - 1) / kNumCU;
    if (g_ldsdma_loose_tiles > 0 && tiles >= g_ldsdma_loose_tiles) return true;      // experiment / ragged-M rule: see g_ldsdma_loose_tiles
    return tiles >= 160 && tiles >= 0.70 * rounds * kNumCU;      // (192 tiles -- GPT-2's 768-wide projections at B T = 8192 -- beat the 128-tile kernel's 384: 18 / 51 vs 29 / 74 us)
}

// bf16 only: N of any size (ragged last column tile, any row pitch) when the grid is many rounds deep -- GPT-2's lm_head (N = 50257, 12 576 tiles at M = 8192)
bool gemm256x128_ragged_n_applicable(int M, int K, int N)
{
    if (M <= 0 || K % 64 != 0 || N % 128 == 0 || !lds_dma_addressable(M, K, N)) return false;
    return (int64_t)((M + 255) / 256) * ((N + 127) / 128) >= 4 * kNumCU;
}

extern int g_gemm_pingpong;
extern int g_gemm_persistent;
// (round 4, with the interior K-tiles in both forms: 576 tiles 45.9 one workgroup per tile / 47.5 walking, 768 tiles 59.8 / 60.4, 544 tiles (N = 8704) 137.0 / 150.2 us,
// 12 576 tiles equal -- profiles/r04_persistent_walk.txt: the walk starts above three rounds)
int g_gemm_walk_min_tiles = 3 * kNumCU + 1;      // tuning "gemm.walk_min_tiles": the 256 x 128 ring walks its tiles from this many on
MILA_TUNE("gemm.walk_min_tiles", g_gemm_walk_min_tiles);
extern int g_gemm_fp8_tail_form;      // gemm_fp8_tail.hip: != 0 sends every row through the tail kernels

template <bool FP8, bool GEGLU, int PP>
static int launch_gemm256x128_tt(const Gemm256Params& p, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set)
    {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256x128_kernel<FP8, GEGLU, PP, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               3 * kStage3Bytes), "hipFuncSetAttribute(gemm256x128)");
        if (rc) return rc;
        if constexpr (PP != 0)
        {
            rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256x128_kernel<FP8, GEGLU, PP, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               3 * kStage3Bytes), "hipFuncSetAttribute(gemm256x128 walk)");
            if (rc) return rc;
        }
        attr_set = true;
    }
    // staggered schedules walk their tiles PERSISTENTLY (one workgroup per CU) once there are more tiles than CUs and a tile has at least two K-tiles
    const int tiles = p.tiles_m * p.tiles_n, nk = p.K / (FP8 ? 128 : 64);
    // (A/B on one box, variants interleaved, tools/bench_gemm_persistent.py -> profiles/r03_persistent_walk.txt: GPT-2 qkv 56.5 -> 54.4 us, fc_1 + GELU 68.5 -> 65.9, the fp4
    // policy's fc_gate_up + GeGLU 222.4 -> 215.6.  A walking workgroup must retire its tile's stores before its third K-tile -- vmcnt counts loads and stores in one
    // order -- where a new workgroup starts with a fresh counter, so the walk pays only when several tiles share the saved launches.  Start phases staggered over the
    // CUs, to spread the store bursts of equal tiles, measured 2-7 % SLOWER on every shape: not kept)
    const int grid = (PP != 0 && nk >= 2 && g_gemm_persistent && tiles >= g_gemm_walk_min_tiles) ? kNumCU : tiles;
    if constexpr (PP != 0)
    {
        if (grid < tiles)
        {
            hipLaunchKernelGGL((gemm256x128_kernel<FP8, GEGLU, PP, true>), dim3(grid), dim3(512), 3 * kStage3Bytes, s, p);
            MILA_LAUNCH_CHECK("gemm256x128 (walk)");
        }
    }
    hipLaunchKernelGGL((gemm256x128_kernel<FP8, GEGLU, PP, false>), dim3(tiles), dim3(512), 3 * kStage3Bytes, s, p);
    MILA_LAUNCH_CHECK("gemm256x128");
}
template <bool FP8, bool GEGLU = false>
static int launch_gemm256x128_t(const Gemm256Params& p, hipStream_t s)
{
    if (g_gemm_pingpong == 5) return launch_gemm256x128_tt<FP8, GEGLU, 2>(p, s);
    return g_gemm_pingpong ? launch_gemm256x128_tt<FP8, GEGLU, 1>(p, s) : launch_gemm256x128_tt<FP8, GEGLU, 0>(p, s);
}
int launch_gemm256x128(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act)
{
    note_form("gemm256x128");
    Gemm256Params p{Y, X, W, bias, M, K, N, (M + 255) / 256, (N + 127) / 128, nullptr, nullptr, 0, act};
    return launch_gemm256x128_t<false>(p, s);
}

// ---- split-K (round 3): short prompts and the remainders of long ones.  A 300-row prompt gives fc_down (N = 3840, K = 15360) 60 tiles of 240 K-tiles each: 60 CUs
// busy for a full-length K loop, 196 idle.  The split form starts S copies of the tile list, copy ks taking 1 / S of K, and a second kernel sums the S fp32 partials in
// a fixed order and applies the epilogue (bias, GELU, bf16) -- same bits whatever the timing.  The partials live in caller workspace (S M N floats, <= 32 MiB since
// tiles x S <= 256 CUs): the entry that takes one is mila_cdna4_gemm_bf16_ws, the counterpart of the cuBLASLt workspace CudaLinearOp hands its plans
// (CudaLinearOp.ixx:637-638, CudaExecutionContext.ixx:337).
int g_gemm_splitk = 1;            // tuning "gemm.splitk": the split-K forms of gemm_bf16_ws / gemm_fp8_scaled_ws
MILA_TUNE("gemm.splitk", g_gemm_splitk);
int gemm_splitk_for(int M, int K, int N)      // S (>= 2), or 0: no split-K form for this shape
{
    if (!g_gemm_splitk || g_gemm_pingpong != 5) return 0;
    if (M <= 0 || N % 128 != 0 || K % 64 != 0 || !lds_dma_addressable(M, K, N)) return 0;
    const int tiles = ((M + 255) / 256) * (N / 128), nk = K / 64;
    if (tiles > kNumCU / 2) return 0;
    const int S = min(min(kNumCU / tiles, nk / 8), 16);      // at least 8 K-tiles per copy: the ring's fill and drain are 3
    return S >= 2 ? S : 0;
}

// y = epilogue(sum over s of P[s]), 8 columns per thread; the epilogue is the 256 x 128 kernel's: bf16(acc) [+ bias, rounded again] [-> GELU of the rounded value]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(uint16_t* __restrict__ Y, const float* __restrict__ P, const uint16_t* __restrict__ bias, int64_t MN, int N, int S,
                                                            int act, int ldy)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= MN) return;
    f32x4 a = *reinterpret_cast<const f32x4*>(P + i), b = *reinterpret_cast<const f32x4*>(P + i + 4);
    for (int s_ = 1; s_ < S; ++s_)
    {
        a += *reinterpret_cast<const f32x4*>(P + (size_t)s_ * MN + i);
        b += *reinterpret_cast<const f32x4*>(P + (size_t)s_ * MN + i + 4);
    }
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (bias)
    {
        const int n = (int)(i % N);
        const u32x4 bb = *reinterpret_cast<const u32x4*>(bias + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = round_bf16(v[e]) + bf16_bits_to_f32((uint16_t)(bb[e >> 1] >> ((e & 1) * 16)));
    }
    if (act)
    {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(round_bf16(v[e]));
    }
    st16(Y + (ldy == N ? i : (i / N) * ldy + i % N), u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])});
}

// the fp8 forms: W4A8 y = bf16(float(bf16(sum * *w_scale)) * x_scales[m] + bias), W8A8 (w_pc) y = bf16((sum * w_scale[n]) * x_scales[m] + bias): the epilogue of the fp8 kernels (common.h: fp8_scale_bias)
__global__ __launch_bounds__(256) void splitk_reduce_fp8_kernel(uint16_t* __restrict__ Y, const float* __restrict__ P, const uint16_t* __restrict__ bias,
                                                                const float* __restrict__ x_scales, const float* __restrict__ w_scale, int w_pc, int64_t MN, int N, int S, int ldy)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= MN) return;
    f32x4 a = *reinterpret_cast<const f32x4*>(P + i), b = *reinterpret_cast<const f32x4*>(P + i + 4);
    for (int s_ = 1; s_ < S; ++s_)
    {
        a += *reinterpret_cast<const f32x4*>(P + (size_t)s_ * MN + i);
        b += *reinterpret_cast<const f32x4*>(P + (size_t)s_ * MN + i + 4);
    }
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
    const float ts = x_scales[m];
    float ws[8];
    if (w_pc)
    {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w_scale + n), w1 = *reinterpret_cast<const f32x4*>(w_scale + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ws[e] = w0[e]; ws[4 + e] = w1[e]; }
    }
    else
    {
        const float w = *w_scale;
#pragma unroll
        for (int e = 0; e < 8; ++e) ws[e] = w;
    }
    u32x4 bb = {0u, 0u, 0u, 0u};
    if (bias) bb = *reinterpret_cast<const u32x4*>(bias + n);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fp8_scale_bias(w_pc != 0, v[e], ws[e], ts, bias != nullptr, bf16_bits_to_f32((uint16_t)(bb[e >> 1] >> ((e & 1) * 16))));
    st16(Y + (size_t)m * ldy + n, u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])});
}

int gemm_fp8_splitk_for(int M, int K, int N)      // S (>= 2), or 0: the fp8 (W4A8) 256 x 128 ring has no split-K form for this shape
{
    if (!g_gemm_splitk || g_gemm_pingpong != 5 || g_gemm_fp8_tail_form != 0) return 0;
    if (M <= 0 || N % 128 != 0 || K % 128 != 0 || !lds_dma_addressable(M, K, N)) return 0;
    const int tiles = ((M + 255) / 256) * (N / 128), nk = K / 128;
    if (tiles > kNumCU / 2) return 0;
    const int S = min(min(kNumCU / tiles, nk / 8), 16);
    return S >= 2 ? S : 0;
}

static int launch_gemm256x128_fp8_splitk(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, const uint16_t* bias, int M, int K, int N,
                                         hipStream_t s, float* partials, int S, int ldy = 0)
{
    static bool attr_set = false;
    if (!attr_set)
    {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256x128_kernel<true, false, 2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               3 * kStage3Bytes), "hipFuncSetAttribute(gemm256x128 fp8 split-K)");
        if (rc) return rc;
        attr_set = true;
    }
    note_form("fp8_gemm256x128_splitk");
    Gemm256Params p{nullptr, reinterpret_cast<const uint16_t*>(X8), reinterpret_cast<const uint16_t*>(W8), nullptr, M, K, N, (M + 255) / 256, N / 128, nullptr, nullptr, 0, 0, partials, S};
    hipLaunchKernelGGL((gemm256x128_kernel<true, false, 2, false, true>), dim3(p.tiles_m * p.tiles_n * S), dim3(512), 3 * kStage3Bytes, s, p);
    int rc = check_hip(hipGetLastError(), "gemm256x128 (fp8 split-K)");
    if (rc) return rc;
    const int64_t MN = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_fp8_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, s, Y, partials, bias, x_scales, w_scale.p, w_scale.per_channel, MN, N, S, ldy ? ldy : N);
    return check_hip(hipGetLastError(), "splitk_reduce_fp8");
}

// the second kernel on its own (the few-row form of gemm_fewrow_bf16.hip writes the same [S][M][N] partials)
int launch_splitk_reduce(uint16_t* Y, const float* partials, const uint16_t* bias, int M, int N, int S, int act, hipStream_t s)
{
    const int64_t MN = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, s, Y, partials, bias, MN, N, S, act, N);
    return check_hip(hipGetLastError(), "splitk_reduce");
}

int launch_gemm256x128_splitk(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act, float* partials, int S, int ldy)
{
    static bool attr_set = false;
    if (!attr_set)
    {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256x128_kernel<false, false, 2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               3 * kStage3Bytes), "hipFuncSetAttribute(gemm256x128 split-K)");
        if (rc) return rc;
        attr_set = true;
    }
    note_form("gemm256x128_splitk");
    Gemm256Params p{Y, X, W, nullptr, M, K, N, (M + 255) / 256, N / 128, nullptr, nullptr, 0, 0, partials, S};
    hipLaunchKernelGGL((gemm256x128_kernel<false, false, 2, false, true>), dim3(p.tiles_m * p.tiles_n * S), dim3(512), 3 * kStage3Bytes, s, p);
    int rc = check_hip(hipGetLastError(), "gemm256x128 (split-K)");
    if (rc) return rc;
    const int64_t MN = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, s, Y, partials, bias, MN, N, S, act, ldy ? ldy : N);
    return check_hip(hipGetLastError(), "splitk_reduce");
}

// one 512-thread workgroup per CU: worth it only when the tile count fills whole rounds of 256 CUs
int g_gemm256_min_fill = 80;      // tuning "gemm.tile256_min_fill": the 256 x 256 grid applies when its tiles fill at least this many percent of their rounds of CUs
MILA_TUNE("gemm.tile256_min_fill", g_gemm256_min_fill);
bool gemm256_applicable(int M, int K, int N)
{
    if (M <= 0 || N % 256 != 0 || K % 64 != 0 || !lds_dma_addressable(M, K, N)) return false;
    const int tiles = ((M + 255) / 256) * (N / 256);
    const int rounds = (tiles + kNumCU - 1) / kNumCU;
    return tiles >= 200 && tiles * 100 >= g_gemm256_min_fill * rounds * kNumCU;      // (80 %: nine tile-rows of fc_gate_up -- 1080 tiles, 4.2 rounds walked as 5 -- stay on the fused kernel)
}

// bf16 only: N of any size (ragged last tile-column, any row pitch) on the PERSISTENT 256 x 256 schedule when the grid is many rounds deep -- GPT-2's lm_head
// (N = 50257: 6 304 tiles at M = 8192, K = 768).  A tile there is 12 K-tiles, as long as what a one-tile workgroup pays around them (launch, first-load latency, the
// store drain): the persistent walk overlaps all three with the next tile's K loop
bool gemm256_ragged_n_applicable(int M, int K, int N)
{
    if (M <= 0 || K % 128 != 0 || N % 256 == 0 || !lds_dma_addressable(M, K, N)) return false;
    return (int64_t)((M + 255) / 256) * ((N + 255) / 256) >= 4 * kNumCU;
}

int g_gemm_rowwise = 1;       // tuning "gemm.rowwise_epilogue": 0 = an output whose row pitch is no multiple of 128 bytes keeps the direct epilogue stores
MILA_TUNE("gemm.rowwise_epilogue", g_gemm_rowwise);
int g_gemm_persistent = 1;    // tuning "gemm.persistent": 0 = one workgroup per tile instead of the persistent tile walk
MILA_TUNE("gemm.persistent", g_gemm_persistent);
int g_gemm_pingpong = 5;      // tuning "gemm.schedule": 0 = all eight waves in lockstep; 1 = staggered (ping-pong), four phases per
                              // K-tile in the 256 x 256 kernel; 2 = 1 + prefer the 256 x 128 ring; 3 = staggered, two phases per K-tile in the
                              // 256 x 256 kernel; 4 = 3 + fp8 x fp8 shapes take the 256 x 256 kernel wherever it applies; 5 (default) = 4 with ONE
                              // s_setprio 1 for waves 4-7 (the later-dispatched half loses every issue arbitration by age) instead of a raise around
                              // every MFMA block: 0.5-1 % on each bf16 shape, nothing on fp8; same bits
MILA_TUNE("gemm.schedule", g_gemm_pingpong);

template <int MODE, int PP>
static int launch_gemm256_tt(const Gemm256Params& p, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set)
    {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<MODE, PP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               2 * kBufBytes), "hipFuncSetAttribute(gemm256)");
        if (rc) return rc;
        attr_set = true;
    }
    // two-phase schedules run persistent (one workgroup per CU walking its tiles) when the K-tile count is even (the LDS buffer parity then runs on across tiles)
    const int tiles = p.tiles_m * p.tiles_n, nk = p.K / ((MODE == G_FP8 || MODE == G_FP8_GEGLU) ? 128 : 64);
    const int grid = (PP >= 2 && nk >= 2 && nk % 2 == 0 && g_gemm_persistent && !p.rowwise) ? (tiles < kNumCU ? tiles : kNumCU) : tiles;
#ifdef MILA_GEMM_SKIP
    static const int dbg = getenv("MILA_GEMM_SKIP") ? atoi(getenv("MILA_GEMM_SKIP")) : 0;
    Gemm256Params q = p;
    q.dbg = dbg;
    hipLaunchKernelGGL((gemm256_kernel<MODE, PP>), dim3(grid), dim3(512), 2 * kBufBytes, s, q);
#else
    hipLaunchKernelGGL((gemm256_kernel<MODE, PP>), dim3(grid), dim3(512), 2 * kBufBytes, s, p);
#endif
    MILA_LAUNCH_CHECK("gemm256");
}
template <int MODE>
static int launch_gemm256_t(const Gemm256Params& p, hipStream_t s)
{
    if (g_gemm_pingpong == 5) return launch_gemm256_tt<MODE, 3>(p, s);
    if (g_gemm_pingpong >= 3) return launch_gemm256_tt<MODE, 2>(p, s);
    return g_gemm_pingpong ? launch_gemm256_tt<MODE, 1>(p, s) : launch_gemm256_tt<MODE, 0>(p, s);
}

int launch_gemm256(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act, int ldy)
{
    // a row pitch that is no multiple of 128 bytes: one workgroup per tile and the row-wise epilogue through LDS (two-phase schedules only)
    const int rowwise = ((N & 63) != 0 && g_gemm_pingpong >= 3 && g_gemm_rowwise) ? 1 : 0;
    note_form("gemm256");
    Gemm256Params p{Y, X, W, bias, M, K, N, (M + 255) / 256, (N + 255) / 256, nullptr, nullptr, rowwise, act};
    p.ldy = ldy;
    return launch_gemm256_t<G_PLAIN>(p, s);
}

// ---- column split (round 4): an output whose 256 x 256 tile list ends in a nearly empty round -- Gemma's global qkv_proj, N = 8704 at T = 2048: 272 tiles on 256 CUs (the
// 256 x 128 ring ran it as 544 tiles in three rounds for 2.1 rounds of work, 165 us = 0.83 PFLOP/s) -- is cut at the last whole round: columns [0, n_main) as whole rounds
// of 256 x 256 tiles, the few remaining column tiles through the split-K form of the ring (their S copies cover the idle CUs), both writing their column range of Y with
// the pitch of the whole row.  Needs the caller's workspace (S M (N - n_main) floats).  n_main = 0: no split for this shape.
int g_gemm_colsplit = 1;
MILA_TUNE("gemm.colsplit", g_gemm_colsplit);
int gemm_splitk_for(int M, int K, int N);
int gemm_colsplit_main(int M, int K, int N, int* S_rest)
{
    *S_rest = 0;
    if (!g_gemm_colsplit || g_gemm_pingpong != 5 || M < 256 || N % 256 != 0 || K % 64 != 0 || !lds_dma_addressable(M, K, N)) return 0;
    const int tm = (M + 255) / 256, tn = N / 256, tiles = tm * tn;
    if (tiles <= kNumCU) return 0;
    const int rounds = (tiles + kNumCU - 1) / kNumCU;
    if (tiles >= 0.80 * rounds * kNumCU) return 0;                  // the whole list fills its rounds well enough (gemm256_applicable's rule)
    const int tn_main = ((tiles / kNumCU) * kNumCU) / tm;           // column tiles of the whole rounds
    if (tn_main <= 0 || tn_main >= tn) return 0;
    const int n_main = tn_main * 256, rest = N - n_main;
    const int S = gemm_splitk_for(M, K, rest);
    if (S < 2) return 0;
    *S_rest = S;
    return n_main;
}

// Y[M, F] = GeGLU(X W^T), W = [gate rows 0 .. F-1 | up rows F .. 2F-1]
bool gemm256_geglu_applicable(int M, int K, int F)
{
    if (M <= 0 || F % 128 != 0 || K % 64 != 0 || !lds_dma_addressable(M, K, 2 * F)) return false;
    const int tiles = ((M + 255) / 256) * (F / 128);
    const int rounds = (tiles + kNumCU - 1) / kNumCU;
    return tiles >= 200 && tiles >= 0.80 * rounds * kNumCU;
}
int launch_gemm256_geglu(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, hipStream_t s)
{
    note_form("gemm256_geglu");
    Gemm256Params p{Y, X, W, nullptr, M, K, F, (M + 255) / 256, F / 128, nullptr, nullptr};
    return launch_gemm256_t<G_GEGLU>(p, s);
}

// gemm_fp8_tail.hip: the same arithmetic for any row count (masked 128-row tiles; skinny weight streaming for <= 64 rows)
int launch_gemm_fp8_tail(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, const uint16_t* bias,
                         int M, int K, int N, hipStream_t s);
int launch_gemm_fp8_geglu_tail(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, int M, int K, int F,
                               hipStream_t s);

// Row counts of any kind (the fp4 policy's prefill is W4A8 for EVERY M > 1, CudaLinearOp.ixx:646-715):
//   (short prompts, measured with tools/experiments/short_prompt_rules.py, profiles/r03_short_prompts.txt: below 512 rows the LDS-DMA kernels still win wherever their grid
//    has >= 120 tiles -- qkv, fc_gate_up -- and lose on the N = 3840 shapes, whose 30-60 tiles leave the chip empty; fp4-policy prefill of 300 / 400 / 511 tokens
//    14.8 / 17.4 / 18.3 -> 11.8 / 13.0 / 13.6 ms, monotonic in T again)
//   M >= 512 and N % 128 == 0, K % 128 == 0: the LDS-DMA kernels over ceil(M / 256) tile-rows -- a ragged last tile-row stages row M - 1 for the rows past M
//     and masks its stores (a 208-row tail costs one tile-row, 1/8 of a T = 2048 chunk; on the masked 128-row tiles it cost 40 % of the chunk) -- except that a
//     tail of <= 64 rows goes to the skinny weight-streaming kernel instead (a 1-row tail: +11 % of the chunk instead of +12.5 %, and no MFMA work on padding);
//   everything else: the tail kernels of gemm_fp8_tail.hip alone.
// Rows are independent and the LDS-DMA kernels and the masked tiles run the same instruction chain per output element.
constexpr int kSkinnyTailRows = 64;
// Few rows (tools/experiments/few_row_rules_fp4.sh, profiles/r03_splitk.txt): one 16-row group stays with the skinny kernel -- a weight stream of one byte per weight
// (fp4-policy prefill of 8 / 16 tokens 4.83 / 4.92 ms, against 4.95 / 5.02 on the tile forms); from 17 rows on the tile grids (>= 120 tiles) and the split-K form are faster
// (17 / 24 / 32 tokens 5.39 / 5.50 / 5.69 ms skinny, 5.09 / 5.10 / 5.17 here; 64 tokens 7.15 -> 5.66).  A W4A8 form of the few-row kernel (gemm_fewrow_bf16.hip with
// e4m3 operands; parity-green) was slower than both on every length -- the skinny kernel's fused GeGLU saves fc_gate_up a reduce and an elementwise pass -- and is not kept.
int g_fp8_splitk_min_rows = 17;      // tuning "gemm_fp8.splitk_min_rows": fewer rows stay with the skinny kernel
MILA_TUNE("gemm_fp8.splitk_min_rows", g_fp8_splitk_min_rows);
int g_fp8_big_rule = 3;      // tuning "gemm_fp8.big_rule": 0 = LDS-DMA kernels from 512 rows on (round 3's first rule), 1 = from 128 rows on,
                             // 2 = from 512 rows on or wherever ceil(M / 256) x (W rows / 128) >= 120 tiles, 3 (default) = 2 without the skinny split of a short prompt's remainder
MILA_TUNE("gemm_fp8.big_rule", g_fp8_big_rule);
static int fp8_big_rows(int M, int K, int N_mult, int w_rows)      // rows the LDS-DMA kernels take (0 = none); N_mult: the column granularity the form needs (128, or 64 for 256 x 128 GeGLU)
{
    if (!lds_dma_addressable(M, K, w_rows)) return 0;
    if (g_gemm_fp8_tail_form != 0 || K % 128 != 0 || N_mult == 0) return 0;
    // below two full tile-rows the LDS-DMA kernels pay only where their grid still covers the chip (tuning variable gemm_fp8.big_rule)
    const int tiles = ((M + 255) / 256) * (w_rows / 128);
    const bool big = g_fp8_big_rule == 1 ? M >= 128 : (g_fp8_big_rule >= 2 ? (M >= 512 || (M >= g_fp8_splitk_min_rows && tiles >= 120)) : M >= 512);
    if (!big) return 0;
    if (g_fp8_big_rule == 3 && M < 512) return M;        // a short prompt's <= 64-row remainder stays in the ragged tile-row: a skinny pass re-streams every weight, which only a long main part amortises
    const int tail = M % 256;
    return (tail > 0 && tail <= kSkinnyTailRows) ? M - tail : M;
}
// which LDS-DMA kernel: 2 = 256 x 256 (enough tiles for the chip, N % 256 == 0), 1 = 256 x 128
static int fp8_pick(int rows, int N)
{
    const int tm = (rows + 255) / 256;
    if (N % 256 == 0 && tm * (N / 256) >= 200) return 2;
    return 1;
}

// Y[M, F] = GeGLU of the W4A8 Linear over W8 = [gate rows | up rows] (2F x K e4m3)
int launch_gemm_fp8_geglu(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, int M, int K, int F,
                          hipStream_t s)
{
    const bool form256 = F % 128 == 0 && ((M + 255) / 256) * (F / 128) >= 200;
    const int rows = fp8_big_rows(M, K, form256 ? 128 : (F % 64 == 0 ? 64 : 0), 2 * F);
    if (rows)
    {
        int rc;
        const int tm = (rows + 255) / 256;
        if (!form256)
        {
            Gemm256Params q{Y, reinterpret_cast<const uint16_t*>(X8), reinterpret_cast<const uint16_t*>(W8), nullptr, rows, K, F, tm, F / 64, x_scales, w_scale.p};
            q.w_pc = w_scale.per_channel;
            note_form("fp8_gemm256x128_geglu");
            rc = launch_gemm256x128_t<true, true>(q, s);
        }
        else
        {
            Gemm256Params p{Y, reinterpret_cast<const uint16_t*>(X8), reinterpret_cast<const uint16_t*>(W8), nullptr, rows, K, F, tm, F / 128, x_scales, w_scale.p};
            p.w_pc = w_scale.per_channel;
            note_form("fp8_gemm256_geglu");
            rc = launch_gemm256_t<G_FP8_GEGLU>(p, s);
        }
        if (rc || rows == M) return rc;
    }
    return launch_gemm_fp8_geglu_tail(Y + (size_t)rows * F, X8 + (size_t)rows * K, W8, x_scales + rows, w_scale, M - rows, K, F, s);
}
int launch_gemm_fp8(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, const uint16_t* bias,
                    int M, int K, int N, hipStream_t s)
{
    const int rows = fp8_big_rows(M, K, N % 128 == 0 ? 128 : 0, N);
    if (rows)
    {
        const int which = fp8_pick(rows, N), tm = (rows + 255) / 256;
        Gemm256Params p{Y, reinterpret_cast<const uint16_t*>(X8), reinterpret_cast<const uint16_t*>(W8), bias, rows, K, N, tm, which == 2 ? N / 256 : N / 128, x_scales, w_scale.p};
        p.w_pc = w_scale.per_channel;
        note_form(which == 2 ? "fp8_gemm256" : "fp8_gemm256x128");
        const int rc = which == 2 ? launch_gemm256_t<G_FP8>(p, s) : launch_gemm256x128_t<true>(p, s);
        if (rc || rows == M) return rc;
    }
    return launch_gemm_fp8_tail(Y + (size_t)rows * N, X8 + (size_t)rows * K, W8, x_scales + rows, w_scale, bias, M - rows, K, N, s);
}


// ---- the same with a caller workspace (mila_cdna4_gemm_fp8_scaled_ws; mirrors gemm.hip's bf16_ws_plan): a short prompt whose tile list covers at most half the CUs
// splits K whole; a long prompt's remainder whose ragged tile-row would open another round of the grid (T = 2303 on the N = 3840 shapes) splits K alone ----
struct Fp8WsPlan { int main_rows, S; int n_main = 0; };      // n_main > 0: the column split (see gemm_colsplit_main): columns [0, n_main) on 256 x 256 fp8 tiles, the rest split-K
static Fp8WsPlan fp8_ws_plan(int M, int K, int N)
{
    if (M < g_fp8_splitk_min_rows) return {M, 0};
    int S = gemm_fp8_splitk_for(M, K, N);
    if (S) return {0, S};
    // the 256 x 256 fp8 tile list of N = 8704 at T = 2048 is 272 tiles: one round and sixteen stragglers, walked as two (fp8_pick: 2).  Whole rounds + a split-K rest instead.
    if (g_gemm_colsplit && g_gemm_pingpong == 5 && g_gemm_fp8_tail_form == 0 && M >= 512 && N % 256 == 0 && K % 128 == 0 && lds_dma_addressable(M, K, N))
    {
        const int tm = (M + 255) / 256, tn = N / 256, tiles = tm * tn;
        const int rounds = (tiles + kNumCU - 1) / kNumCU;
        if (tiles > kNumCU && tiles < 0.80 * rounds * kNumCU)
        {
            const int tn_main = ((tiles / kNumCU) * kNumCU) / tm;
            if (tn_main > 0 && tn_main < tn)
            {
                const int n_main = tn_main * 256;
                const int Sr = gemm_fp8_splitk_for(M, K, N - n_main);
                if (Sr >= 2 && fp8_big_rows(M, K, 128, n_main) == M && fp8_pick(M, n_main) == 2) { Fp8WsPlan cs{0, Sr}; cs.n_main = n_main; return cs; }
            }
        }
    }
    const int tail = M % 256, main_rows = M - tail;
    if (M < 512 || tail < g_fp8_splitk_min_rows) return {M, 0};
    if (fp8_big_rows(main_rows, K, N % 128 == 0 ? 128 : 0, N) != main_rows) return {M, 0};
    const int per_row = fp8_pick(main_rows, N) == 2 ? N / 256 : N / 128, tm = main_rows / 256;
    const bool new_round = (tm * per_row + kNumCU - 1) / kNumCU < ((tm + 1) * per_row + kNumCU - 1) / kNumCU;
    if (!new_round) return {M, 0};
    S = gemm_fp8_splitk_for(tail, K, N);
    return S ? Fp8WsPlan{main_rows, S} : Fp8WsPlan{M, 0};
}
size_t gemm_fp8_ws_bytes(int M, int K, int N)
{
    const Fp8WsPlan pl = fp8_ws_plan(M, K, N);
    if (pl.n_main) return (size_t)pl.S * M * (N - pl.n_main) * sizeof(float);
    return pl.S ? (size_t)pl.S * (M - pl.main_rows) * N * sizeof(float) : 0;
}
int launch_gemm_fp8_ws(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, const uint16_t* bias, int M, int K, int N, hipStream_t s,
                       void* ws)
{
    const Fp8WsPlan pl = fp8_ws_plan(M, K, N);
    if (!pl.S) return launch_gemm_fp8(Y, X8, W8, x_scales, w_scale, bias, M, K, N, s);
    if (pl.n_main)
    {
        note_form("fp8_gemm256_colsplit");
        note_form("fp8_gemm256");
        Gemm256Params p{Y, reinterpret_cast<const uint16_t*>(X8), reinterpret_cast<const uint16_t*>(W8), bias, M, K, pl.n_main, (M + 255) / 256, pl.n_main / 256, x_scales, w_scale.p};
        p.w_pc = w_scale.per_channel;
        p.ldy = N;
        int rc = launch_gemm256_t<G_FP8>(p, s);
        if (rc) return rc;
        const Fp8WScale wr{w_scale.per_channel ? w_scale.p + pl.n_main : w_scale.p, w_scale.per_channel};
        return launch_gemm256x128_fp8_splitk(Y + pl.n_main, X8, W8 + (size_t)pl.n_main * K, x_scales, wr, bias ? bias + pl.n_main : nullptr, M, K, N - pl.n_main, s,
                                             static_cast<float*>(ws), pl.S, N);
    }
    if (pl.main_rows > 0)
    {
        int rc = launch_gemm_fp8(Y, X8, W8, x_scales, w_scale, bias, pl.main_rows, K, N, s);
        if (rc) return rc;
    }
    return launch_gemm256x128_fp8_splitk(Y + (size_t)pl.main_rows * N, X8 + (size_t)pl.main_rows * K, W8, x_scales + pl.main_rows, w_scale, bias, M - pl.main_rows, K, N, s,
                                         static_cast<float*>(ws), pl.S);
}
// the fused GeGLU form steps aside where the plain GEMM over [2F, K] would split K (same reason as gemm.hip's geglu_rows_applicable)
bool gemm_fp8_geglu_steps_aside(int M, int K, int F) { return fp8_ws_plan(M, K, 2 * F).S != 0; }

}  // namespace mila
