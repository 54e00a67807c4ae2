// Flash-attention prefill on the matrix cores (causal, optional sliding window, GQA/MQA, ring cache),
// also used for GPT-2's packed-QKV MHA.
//
// Replaces the reference's flash-prefill rungs (OPS/Attention/GQA/Kernels/Gqa.Flash.Wmma.cu:246,
// Gqa.Flash.Fa2.cu:177 -- mma.sync m16n8k16 + ldmatrix, Br = Bc = 16) and its cuBLASLt
// QK -> softmax -> AV pipeline (CudaGqaOp.ixx:673-794).  Mask / scale / head-map semantics as in
// attention.hip.  fp32 scores and softmax state, P rounded to bf16 for the PV product exactly as the
// reference's flash kernels do (SURVEY.md Appendix A "Intermediate precision").
//
// CDNA4 design (re-derived, not translated):
//   * v_mfma_f32_16x16x32_bf16 for both products.  One wave owns 16 query rows of one head and
//     keeps O^T (HS x 16, fp32) in HS/4 accumulator registers and its Q fragments (HS/8 VGPRs).
//   * the products are computed TRANSPOSED: S^T = K Q^T puts a query row on a lane (col = lane & 15)
//     and its keys in the lane's registers, so the row max / row sum are in-lane plus two xor
//     shuffles (16, 32); the exponentiated scores ARE the B operand of O^T += V^T P^T with no lane
//     movement (the PV contraction index is permuted consistently on both operands).
//   * K / V tiles of 32 keys are staged global -> registers -> LDS (next tile's loads in flight during
//     the current tile's math); K fragments are ds_read_b128 from an XOR-swizzled row-major image,
//     V^T fragments come from the row-major V image through ds_read_b64_tr_b16 (hardware transpose),
//     so V is stored exactly as it arrives from HBM.
//   * a workgroup = 4 waves = (query heads sharing one KV head) x (16-row query sub-tiles): all four
//     waves consume the same K/V tiles (GQA/MQA reuse in LDS); the heaviest (latest) query tiles are
//     dispatched first.
#include <cstdlib>

#include "common.h"
#include <type_traits>
#include <utility>
#include "internal.h"
#include "attention_generic.h"
#include "attention_tiles.h"

namespace mila {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in the body (assembly offsets, if constexpr)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// the V^T fragments of DT output tiles from the tile image at LDS byte address vimg: compile-time offsets (d >> 3) * 256 and + 16 rows; after(d) runs behind the
// reads of tile d (the kernels issue the next tile's LDS-DMA requests there, a few reads apart)
template <int ROWB, int BASE, typename After, int... I>
__device__ __forceinline__ void read_vt_frags(const unsigned (&addr)[8], s16x8* va, After&& after, std::integer_sequence<int, I...>)
{
    auto one = [&](auto ic) {
        constexpr int d = decltype(ic)::value;
        const s16x4 lo = lds_read_tr16<BASE + (d >> 3) * 256>(addr[d & 7]);
        const s16x4 hi = lds_read_tr16<BASE + (d >> 3) * 256 + 16 * ROWB>(addr[d & 7]);
        va[d][0] = lo[0]; va[d][1] = lo[1]; va[d][2] = lo[2]; va[d][3] = lo[3];
        va[d][4] = hi[0]; va[d][5] = hi[1]; va[d][6] = hi[2]; va[d][7] = hi[3];
        after(ic);
    };
    (one(std::integral_constant<int, I>{}), ...);
}

// One tile's online-softmax step on the 8 scores a lane holds of its query row (tv: scores x scale x log2 e, -inf where masked), shared by both kernels so that
// every form produces the same bits.  Round 4: the log2 domain (exp2 is the hardware's own function: no multiplication in front of each exponential), and the
// row sum stays PER LANE -- the four lanes of a row rescale their partial sums by the same alpha, so they are added once, in the epilogue, instead of through
// two cross-lane exchanges per tile.  Returns the bf16 P fragment; alpha = the factor of the accumulators (exactly 1 where the row's maximum did not move).
// (maxima as the bare instructions: fmaxf() quiets a possible signalling NaN with a v_max x, x in front of every operand that is not known to come from
// arithmetic -- 11 instructions for the maximum of 8 values and 4 more around the lane exchanges, where 4 + 3 do; a wave issues one instruction per 4 cycles)
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ bf16x8 softmax_tile_step(const float (&tv)[8], float& m_run, float& l_lane, float& alpha)
{
    float mt = vmax2(vmax3(vmax3(vmax3(tv[0], tv[1], tv[2]), tv[3], tv[4]), tv[5], tv[6]), tv[7]);
    {
        // the row's other keys sit on lanes l ^ 16, l ^ 32, l ^ 48
        const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
        mt = vmax2(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
        const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
        mt = vmax2(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    }
    // (A LAZY reference -- keep m until a tile exceeds it by 2^8, so that the accumulators are rescaled in a few tiles instead of 40-60 % of them -- was tried in
    // round 4: mathematically the same O / l, but P is then rounded to bf16 at another, non-power-of-two scale than the reference's flash kernels round it, and
    // 15 of 573 440 outputs of the HS 512 chunked test left the 1 ulp + 2e-3 bar (2 ulp).  The true running maximum stays.)
    const float mn = vmax2(m_run, mt);
    const float msafe = (mn == -INFINITY) ? 0.0f : mn;      // row with nothing visible yet
    alpha = __builtin_amdgcn_exp2f(m_run - msafe);          // m_run = -inf -> 0
    float pe[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) pe[r] = __builtin_amdgcn_exp2f(tv[r] - msafe);
    l_lane = l_lane * alpha + (((pe[0] + pe[1]) + (pe[2] + pe[3])) + ((pe[4] + pe[5]) + (pe[6] + pe[7])));
    m_run = mn;
    // B operand of O^T += V^T P^T: element j <-> key (j < 4 ? 4 g + j : 16 + 4 g + j - 4)
    u32x4 pb;
    pb[0] = pack_bf16x2(pe[0], pe[1]);
    pb[1] = pack_bf16x2(pe[2], pe[3]);
    pb[2] = pack_bf16x2(pe[4], pe[5]);
    pb[3] = pack_bf16x2(pe[6], pe[7]);
    return __builtin_bit_cast(bf16x8, pb);
}

struct FlashParams
{
    uint16_t* Y;              // [B*Tq, NH*HS]
    const uint16_t* Q;        // row (b*Tq+t): Q + (b*Tq+t)*q_row_stride + h*HS
    const uint16_t* K;        // K + b*kv_b_stride + kvh*kv_h_stride + row*kv_r_stride, row = pos % capacity
    const uint16_t* V;
    int64_t q_row_stride, kv_b_stride, kv_h_stride, kv_r_stride;
    int Tq, NH, NKV, capacity, pos_offset, window;
    float scale;
    int n_qtiles, n_hblk;    // set by the launcher: query tiles and head blocks (grid.x = n_qtiles * n_hblk)
};

// HB = heads per workgroup (1, 2 or 4); QB = 4 / HB query sub-tiles of 16 rows
template <int HS, int HB>
__global__ __launch_bounds__(256, 2) void flash_prefill_kernel(const FlashParams p)
{
    constexpr bool DEEP = true;              // two staging register sets (HS = 512 runs flash_prefill_kernel_s1)
    constexpr int QB = 4 / HB;
    constexpr int QROWS = 16 * QB;
    constexpr int KSTEPS = HS / 32;          // MFMA k-steps of the QK^T product
    constexpr int DT = HS / 16;              // 16-wide d tiles of O^T
    constexpr int ROWB = HS * 2;
    constexpr int TILE_BYTES = kKeysPerTile * ROWB;
    constexpr int CH_PER_THREAD = (kKeysPerTile * (ROWB / 16)) / 256;   // 16-byte chunks each thread stages per tile
    static_assert(CH_PER_THREAD >= 1, "tile too small");

    // TWO [K tile | V tile] buffers: tile t is stored into buffer t & 1 while slower waves still read tile t - 1 from the other one, so a
    // tile costs one workgroup barrier instead of two (the store of tile t follows the barrier of tile t - 1, which every wave reaches
    // only after its reads of tile t - 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int GS = p.NH / p.NKV;
    // Causal work per query tile grows with its index.  Two workgroups are resident per CU and the hardware deals workgroup ids
    // round-robin over the CUs, so ids c and c + 256 end up sharing a CU: even rounds of 256 ids take tiles from the heavy end of
    // the (tile, head-block) list, odd rounds from the light end -- every CU gets a heavy and a light workgroup instead of two heavy ones.
    const int n_items = p.n_qtiles * p.n_hblk;
    const int bid = blockIdx.x, round = bid / kNumCU, k = (round >> 1) * kNumCU + bid % kNumCU;
    const int item = (round & 1) ? n_items - 1 - k : k;         // index in the heavy -> light order
    const int qt = p.n_qtiles - 1 - item / p.n_hblk;
    const int hblk = item % p.n_hblk, b = blockIdx.z;
    const int h = hblk * HB + (wave % HB);
    const int kvh = (hblk * HB) / GS;                          // all HB heads share one KV head (HB | GS)
    const int q0 = qt * QROWS;                                 // first query row (within the chunk) of this workgroup
    const int wq0 = q0 + 16 * (wave / HB);                     // this wave's first row
    const int my_row = wq0 + l15;                              // the query row on this lane
    const bool row_valid = my_row < p.Tq;
    const int my_pos = p.pos_offset + (row_valid ? my_row : p.Tq - 1);

    // ---- Q fragments: B operand of S^T = K Q^T: lane holds Q[row l15][32 s + 8 g + j] ----
    s16x8 qf[KSTEPS];
    {
        const uint16_t* qp = p.Q + ((size_t)b * p.Tq + (row_valid ? my_row : 0)) * p.q_row_stride + (size_t)h * HS + 8 * g;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
        {
            const u32x4 v = row_valid ? ld16(qp + 32 * s) : u32x4{0u, 0u, 0u, 0u};
            qf[s] = __builtin_bit_cast(s16x8, v);
        }
    }

    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float m_run = -INFINITY, l_run = 0.0f;                       // running maximum (log2 domain) of the lane's row; the LANE's share of the row sum
    const float c2 = p.scale * 1.4426950408889634f;              // scores x scale x log2 e: p = exp2(. - m)

    // key range needed by the workgroup (union over its rows)
    const int pos_first = p.pos_offset + q0;
    const int pos_last = p.pos_offset + min(q0 + QROWS, p.Tq) - 1;
    const int kmin = (p.window > 0) ? max(0, pos_first - p.window + 1) : 0;
    const int kt0 = kmin & ~(kKeysPerTile - 1);
    const int ntiles = (pos_last - kt0) / kKeysPerTile + 1;

    const uint16_t* kbase = p.K + (size_t)b * p.kv_b_stride + (size_t)kvh * p.kv_h_stride;
    const uint16_t* vbase = p.V + (size_t)b * p.kv_b_stride + (size_t)kvh * p.kv_h_stride;

    // K/V tiles travel global -> registers -> LDS with TWO register sets: the loads of tile t + 2 are issued while tile t is
    // being multiplied, so a whole tile of math covers the L2 round trip (with one set the wait at the next stage_store cost
    // 30 % of the kernel).  Loads are branch-free (rows past the workgroup's last key are clamped to it: their scores are masked and
    // the clamped V row is a real, finite row), so the compiler keeps the other set's loads in flight across the wait.
    struct StageRegs { u32x4 k[CH_PER_THREAD], v[CH_PER_THREAD]; };
    const int kt_last = kt0 + (ntiles - 1) * kKeysPerTile;
    const bool ring = pos_last >= p.capacity;                   // uniform: only a bounded ring wraps inside one prefill
    const int wpos0 = p.pos_offset + wq0;                        // position of this wave's first row (uniform)
    const bool rows_ok = wq0 + 16 <= p.Tq;
    auto stage_load = [&](StageRegs& r, int kt) {
        if constexpr (DEEP) kt = min(kt, kt_last);
#pragma unroll
        for (int i = 0; i < CH_PER_THREAD; ++i)
        {
            const int c = tid + 256 * i;
            const int row = c / (ROWB / 16), chunk = c % (ROWB / 16);
            if constexpr (DEEP)
            {
                const int pos = min(kt + row, pos_last);
                // (round 4: the modulo only where a bounded ring can wrap inside this prefill -- 10 vector instructions per request otherwise spent on pos % capacity)
                const size_t off = (size_t)(ring ? pos % p.capacity : pos) * p.kv_r_stride + chunk * 8;
                r.k[i] = ld16(kbase + off);
                r.v[i] = ld16(vbase + off);
            }
            else
            {
                // HS = 512: every register counts; the guarded form keeps the loads late and the pressure at 244
                const int pos = kt + row;
                if (pos <= pos_last)     // rows beyond the last key any row of this workgroup may see stay zero
                {
                    const size_t off = (size_t)(pos % p.capacity) * p.kv_r_stride + chunk * 8;
                    r.k[i] = ld16(kbase + off);
                    r.v[i] = ld16(vbase + off);
                }
                else
                {
                    r.k[i] = u32x4{0u, 0u, 0u, 0u};
                    r.v[i] = u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
    };
    auto stage_store = [&](const StageRegs& r, unsigned char* ldsK, unsigned char* ldsV) {
#pragma unroll
        for (int i = 0; i < CH_PER_THREAD; ++i)
        {
            const int c = tid + 256 * i;
            const int row = c / (ROWB / 16), chunk = c % (ROWB / 16);
            *reinterpret_cast<u32x4*>(ldsK + k_off<HS>(row, chunk)) = r.k[i];
            *reinterpret_cast<u32x4*>(ldsV + v_off<HS>(row, chunk)) = r.v[i];
        }
    };

    StageRegs ra, rb;
    stage_load(ra, kt0);
    if constexpr (DEEP) stage_load(rb, kt0 + kKeysPerTile);
    auto tile = [&](int t, StageRegs& regs) {
        const int kt = kt0 + t * kKeysPerTile;
        unsigned char* ldsK = smem + (t & 1) * 2 * TILE_BYTES;
        unsigned char* ldsV = ldsK + TILE_BYTES;
        stage_store(regs, ldsK, ldsV);
        __syncthreads();
        // DEEP: two tiles ahead, in flight during this tile's and the next one's math; else the next tile
        if (DEEP || t + 1 < ntiles) stage_load(regs, kt + (DEEP ? 2 : 1) * kKeysPerTile);

        // ---- S^T = K Q^T : two 16-key groups ----
        f32x4 s0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
        {
            // A operand: K[key l15 (+16)][32 s + 8 g .. +7] -> chunk 4 s + g
            const s16x8 ka = *reinterpret_cast<const s16x8*>(ldsK + k_off<HS>(l15, 4 * s + g));
            const s16x8 kb = *reinterpret_cast<const s16x8*>(ldsK + k_off<HS>(16 + l15, 4 * s + g));
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka), __builtin_bit_cast(bf16x8, qf[s]), s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb), __builtin_bit_cast(bf16x8, qf[s]), s1, 0, 0, 0);
        }
        // lane holds keys kt + 4 g + r (s0) and kt + 16 + 4 g + r (s1) of query row l15
        float tv[8];
        // a tile every row of this wave sees whole (the interior of the band: most tiles) needs no mask (round 4: this kernel masked every tile -- 67 of its 214
        // instructions per tile for 8 MFMAs; same values either way)
        const bool whole = rows_ok && kt + kKeysPerTile - 1 <= wpos0 && (p.window == 0 || kt > wpos0 + 15 - p.window);
        if (whole)
        {
#pragma unroll
            for (int r = 0; r < 8; ++r) tv[r] = ((r < 4) ? s0[r] : s1[r - 4]) * c2;
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                const int key = kt + ((r < 4) ? (4 * g + r) : (16 + 4 * g + (r - 4)));
                const float raw = (r < 4) ? s0[r] : s1[r - 4];
                const bool vis = row_valid && key <= my_pos && (p.window == 0 || key > my_pos - p.window);
                tv[r] = vis ? raw * c2 : -INFINITY;
            }
        }
        float alpha;
        const bf16x8 pfrag = softmax_tile_step(tv, m_run, l_run, alpha);
        // few output tiles (HS = 64: DT = 4): the rescale is multiplied in unconditionally -- alpha is exactly 1 where the row's maximum did not move, so the bits are
        // the same, and 16 multiplies cost less than the vote plus the 16 selects the compiler made of the guarded form (round 4: 54 v_cndmask per pair of tiles)
        const bool rescale = DT <= 4 || __any(alpha != 1.0f);
#pragma unroll
        for (int d = 0; d < DT; ++d)
        {
            // A operand: V^T[dim 16 d + l15][keys as above] via the transposing read:
            // lane 4 q + pp of a 16-lane group supplies row q, columns 4 pp .. 4 pp + 3 of the block
            const int q4 = l15 >> 2, pp = l15 & 3;
            const int col = 16 * d + 4 * pp;                   // first of 4 columns (8 bytes)
            const int r_lo = 4 * g + q4, r_hi = 16 + 4 * g + q4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(ldsV + v_off<HS>(r_lo, col >> 3) + ((col & 7) << 1)));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(ldsV + v_off<HS>(r_hi, col >> 3) + ((col & 7) << 1)));
            s16x8 va;
            va[0] = lo[0]; va[1] = lo[1]; va[2] = lo[2]; va[3] = lo[3];
            va[4] = hi[0]; va[5] = hi[1]; va[6] = hi[2]; va[7] = hi[3];
            if (rescale)
            {
                o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha;
            }
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va), pfrag, o[d], 0, 0, 0);
        }
    };
    if constexpr (DEEP)
    {
        for (int t = 0; t < ntiles; t += 2)
        {
            tile(t, ra);
            if (t + 1 < ntiles) tile(t + 1, rb);
        }
    }
    else
    {
        for (int t = 0; t < ntiles; ++t) tile(t, ra);
    }

    // ---- epilogue: O^T[dim 16 d + 4 g + r][row l15] -> Y[row][h*HS + dim] ----
    const float l_row = quad_rows_sum(l_run);                   // the four lanes of a row
    if (row_valid)
    {
        const float inv = (l_row > 0.0f) ? 1.0f / l_row : 0.0f;
        uint16_t* y = p.Y + (((size_t)b * p.Tq + my_row) * p.NH + h) * HS + 4 * g;
#pragma unroll
        for (int d = 0; d < DT; ++d)
            *reinterpret_cast<u32x2*>(y + 16 * d) = u32x2{pack_bf16x2(o[d][0] * inv, o[d][1] * inv), pack_bf16x2(o[d][2] * inv, o[d][3] * inv)};
    }
}

// Diagnostic build only (-DMILA_FLASH_STAMPS, never the product library): wave 0 of workgroup 0 of flash_prefill_kernel_s1 accumulates the shader cycles of each
// segment of its tile loop (wait + barrier | staging issue | QK^T | softmax | PV) into g_flash_stamps; read back with mila_dbg_flash_stamps().
#ifdef MILA_FLASH_STAMPS
__device__ unsigned long long g_flash_stamps[16];
#define FLASH_STAMP(i) do { if (stamping) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); seg[i] += now_ - last_; last_ = now_; } } while (0)
#else
#define FLASH_STAMP(i) do { } while (0)
#endif

// Single-staging form (one register set, next tile's loads in flight during the current tile's math): HS = 512, where the
// accumulators (128 VGPRs) and Q fragments (64) leave no room for a second staging set.
// HB = heads per workgroup (1, 2 or 4); QB = 4 / HB query sub-tiles of 16 rows
// DS = 2: a head's output dimensions are SPLIT over two waves (each computes the whole S^T and softmax of its 16 rows -- identical in both -- and half of
// O^T): 64 accumulator registers instead of 128, so the kernel fits 256 registers, two workgroups share a CU (two waves per SIMD cover each other's LDS
// latency) and the compiler has registers left to batch its fragment reads.  QK^T is computed twice per head (1.5x the MFMA work at 8 % utilisation);
// every output element is produced by the same instruction sequence as with DS = 1, so the results are bit-identical.
// NW = 8 (HB = 4, DS = 2): ONE 8-wave workgroup per CU -- four heads x two d-halves share every K / V tile (half the L2 -> LDS traffic per FLOP of the 4-wave
// form), the tiles are double-buffered in 128 KB of LDS, so tile t + 1 is requested at the top of tile t and lands under a whole tile of arithmetic, and
// one barrier per tile suffices (it says both "tile t is here" and "everybody is done with tile t - 1").
template <int HS, int HB, int DS, int NW = 4, bool PIPE = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : (HS <= 256 ? 2 : DS)) void flash_prefill_kernel_s1(const FlashParams p)
{
    static_assert(!PIPE || (NW == 8) || (HS <= 256), "the software-pipelined loop is a form of the double-buffered kernels");
    static_assert(DS == 1 || DS == 2, "d-split");
    static_assert(NW == 4 || (NW == 8 && ((HB == 4 && DS == 2) || (HB == 2 && DS == 1))), "workgroup shapes");
    // double-buffered tiles, one barrier per tile: the 8-wave form, and every HS <= 256 form (two [K | V] pairs are 64 KB there: two workgroups still share a CU)
    constexpr bool DB = (NW == 8) || (HS <= 256);
    constexpr bool ASM_TR = !(HS >= 512 && DS == 1);         // V^T fragments by inline-assembly reads (every form but the spilling one, see the tile body)
    // HS = 512 (round 4): S^T is the SUM OF TWO half-dimension products in every form, so that all forms give the same bits.  With DS = 2 the two d-share waves
    // of a (head, row block) each multiply ONE half -- half the K fragment reads, half the QK^T MFMAs, half the Q fragment registers -- and exchange their
    // partial scores through 2 KB of LDS per wave (both then add own + partner: the same two numbers); before, each of them computed the whole product.
    constexpr bool HALVES = HS >= 512;
    constexpr bool XCH = HALVES && DS == 2;
    constexpr int QB = NW / (HB * DS);
    static_assert(QB >= 1, "at most NW (head, d-half) pairs per workgroup");
    constexpr int QROWS = 16 * QB;
    constexpr int KSTEPS = HS / 32;          // MFMA k-steps of the QK^T product
    constexpr int KOWN = XCH ? KSTEPS / 2 : KSTEPS;      // ... of which this wave multiplies (its half under the exchange)
    constexpr int DT = HS / 16 / DS;         // 16-wide d tiles of O^T this wave owns
    constexpr int ROWB = HS * 2;
    constexpr int TILE_BYTES = kKeysPerTile * ROWB;
    constexpr int XCH_OFF = ((NW == 8) || (HS <= 256) ? 4 : 2) * TILE_BYTES;      // the exchange area behind the tile buffers: [wave][lane][8 floats] (PIPE: two of them, by tile parity)
    constexpr int XCH_BYTES = NW * 2048;
    constexpr int RPI = 1024 / ROWB;                     // K / V rows one LDS-DMA wave-instruction (64 lanes x 16 bytes) covers
    static_assert(RPI == 1 || RPI == 2, "HS = 512 or 256");
    constexpr int CPR = ROWB / 16;                       // 16-byte chunks per row
    constexpr int DMAS = kKeysPerTile / NW / RPI;        // LDS-DMA instructions each wave issues per tile and matrix
    static_assert(DMAS >= 1, "tile too small for the wave count");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [K tile | V tile] (NW = 8: two of them)
#ifdef MILA_FLASH_STAMPS
    const unsigned long long entry_ = __builtin_amdgcn_s_memtime();
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int GS = p.NH / p.NKV;
    // Causal work per query tile grows with its index.  Two workgroups are resident per CU and the hardware deals workgroup ids
    // round-robin over the CUs, so ids c and c + 256 end up sharing a CU: even rounds of 256 ids take tiles from the heavy end of
    // the (tile, head-block) list, odd rounds from the light end -- every CU gets a heavy and a light workgroup instead of two heavy ones.
    const int n_items = p.n_qtiles * p.n_hblk;
    const int bid = blockIdx.x, round = bid / kNumCU, k = (round >> 1) * kNumCU + bid % kNumCU;
    // (NW = 8: one workgroup per CU at a time; the dispatcher hands the next id to the CU that frees up first, so plain heavy -> light order balances)
    const int item = (NW == 8) ? bid : ((round & 1) ? n_items - 1 - k : k);         // index in the heavy -> light order
    const int qt = p.n_qtiles - 1 - item / p.n_hblk;
    const int hblk = item % p.n_hblk, b = blockIdx.z;
    const int h = hblk * HB + (wave % HB);
    const int dsel = (wave / HB) % DS;                         // which share of the head's output dimensions this wave accumulates
    const int kvh = (hblk * HB) / GS;                          // all HB heads share one KV head (HB | GS)
    const int q0 = qt * QROWS;                                 // first query row (within the chunk) of this workgroup
    const int wq0 = q0 + 16 * (wave / (HB * DS));              // this wave's first row
    const int my_row = wq0 + l15;                              // the query row on this lane
    const bool row_valid = my_row < p.Tq;
    const int my_pos = p.pos_offset + (row_valid ? my_row : p.Tq - 1);

    // ---- Q fragments: B operand of S^T = K Q^T: lane holds Q[row l15][32 s + 8 g + j]; under the exchange only this wave's half of the dimensions ----
    s16x8 qf[KOWN];
    {
        const uint16_t* qp = p.Q + ((size_t)b * p.Tq + (row_valid ? my_row : 0)) * p.q_row_stride + (size_t)h * HS + 8 * g + (XCH ? dsel * (HS / 2) : 0);
#pragma unroll
        for (int s = 0; s < KOWN; ++s)
        {
            const u32x4 v = row_valid ? ld16(qp + 32 * s) : u32x4{0u, 0u, 0u, 0u};
            qf[s] = __builtin_bit_cast(s16x8, v);
        }
    }

    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float m_run = -INFINITY, l_run = 0.0f;                       // running maximum (log2 domain) of the lane's row; the LANE's share of the row sum
    const float c2 = p.scale * 1.4426950408889634f;              // scores x scale x log2 e: p = exp2(. - m)

    // key range needed by the workgroup (union over its rows)
    const int pos_first = p.pos_offset + q0;
    const int pos_last = p.pos_offset + min(q0 + QROWS, p.Tq) - 1;
    const int kmin = (p.window > 0) ? max(0, pos_first - p.window + 1) : 0;
    const int kt0 = kmin & ~(kKeysPerTile - 1);
    const int ntiles = (pos_last - kt0) / kKeysPerTile + 1;

    const uint16_t* kbase = p.K + (size_t)b * p.kv_b_stride + (size_t)kvh * p.kv_h_stride;
    const uint16_t* vbase = p.V + (size_t)b * p.kv_b_stride + (size_t)kvh * p.kv_h_stride;

    const bool ring = pos_last >= p.capacity;                   // uniform: only a bounded ring wraps inside one prefill
    // K / V tiles go global -> LDS by LDS-DMA, no staging registers: wave w requests rows w, w + NW, ... of a tile, RPI rows (1 KiB) per instruction; a lane's LDS
    // slot is linear (M0 base + 16 lane), so the swizzle of k_off / v_off -- an involution on the chunk index -- is applied to the SOURCE chunk it fetches.
    // Every tile requests all its rows, so the request counts the waits rely on are constant: a row beyond the last key any row of this workgroup may see
    // re-reads that last key (a finite in-band row; the mask works on key positions, not contents).
    // (round 4) The requests are buffer loads (buffer_load_dwordx4 ... offen lds) on one resource per matrix, based at this (batch, KV head): a lane's 32-bit
    // byte offset -- its row inside a 16-row group and its swizzled chunk, fixed for the whole kernel -- rides in the vector offset, the tile's rows in the
    // SCALAR offset: a request of a whole in-band tile costs one scalar add where the global_load_lds form rebuilt a 64-bit vector address per request
    // (v_lshl_add_u64 pairs, eight long-lived address registers -- at HS = 512 the kernel spilled for them).  A wave's consecutive instructions are STEP rows
    // apart and the swizzles repeat every 16 rows, so instructions i and i + 16 / STEP differ by exactly 16 rows: one or two lane offsets per matrix cover them.
    // The last one or two tiles of a band and a wrapped ring take the general form: per-lane clamp / modulo in 32-bit arithmetic, scalar offset 0.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kbase), 0, 0x7fffffff, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vbase), 0, 0x7fffffff, 0x00020000);
    const int rstride = (int)(p.kv_r_stride * 2);                // row pitch in bytes (the launcher checks that a cache's rows fit 31 bits of byte offset)
    constexpr int STEP = RPI * NW, NB = 16 / STEP;
    static_assert(!DB || NB == 1 || NB == 2, "the double-buffered forms issue whole 16-row groups per one or two instructions");
    int ks0 = 0, ks1 = 0, vs0 = 0, vs1 = 0;
    {
        const int slot = lane % CPR;
        const int r0 = RPI * wave + lane / CPR, r1 = r0 + STEP;
        ks0 = r0 * rstride + (k_off<HS>(r0, slot) - r0 * ROWB);
        vs0 = r0 * rstride + (v_off<HS>(r0, slot) - r0 * ROWB);
        ks1 = r1 * rstride + (k_off<HS>(r1, slot) - r1 * ROWB);
        vs1 = r1 * rstride + (v_off<HS>(r1, slot) - r1 * ROWB);
    }
    auto stage_k = [&](int kt, unsigned char* ldsK) {
#pragma unroll
        for (int i = 0; i < DMAS; ++i)
        {
            // (an opaque copy of the lane id: this address arithmetic -- first tile, the last tiles of a band -- is redone in place; hoisted out of the tile
            // loops, its per-request invariants cost the registers the fragments need)
            int lz = lane;
            asm volatile("" : "+v"(lz));
            const int row0 = RPI * (NW * i + wave);                       // the instruction's first row: RPI consecutive rows = 1 KiB of LDS
            const int row = row0 + lz / CPR, slot = lz % CPR, pos = min(kt + row, pos_last);
            const int voff = (ring ? pos % p.capacity : pos) * rstride + (k_off<HS>(row, slot) - row * ROWB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr_t)(ldsK + row0 * ROWB), 16, voff, 0, 0, 0);
        }
    };
    auto stage_v = [&](int kt, unsigned char* ldsV) {
#pragma unroll
        for (int i = 0; i < DMAS; ++i)
        {
            int lz = lane;
            asm volatile("" : "+v"(lz));
            const int row0 = RPI * (NW * i + wave);
            const int row = row0 + lz / CPR, slot = lz % CPR, pos = min(kt + row, pos_last);
            const int voff = (ring ? pos % p.capacity : pos) * rstride + (v_off<HS>(row, slot) - row * ROWB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr_t)(ldsV + row0 * ROWB), 16, voff, 0, 0, 0);
        }
    };
    // Per-lane LDS offsets of the fragment reads, once: the swizzles touch only the low four bits of the chunk index, so the offsets repeat every 16 chunks
    // (256 bytes) and the second 16-key group of a tile is 16 rows further: four K and eight V offsets per lane, everything else is an immediate.
    int kaddr[4], vaddr[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) kaddr[i] = k_off<HS>(l15, 4 * i + g) + (XCH ? dsel * (KOWN / 4) * 256 : 0);      // (under the exchange: this wave's half of the k-steps)
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        const int col = 16 * (dsel * DT + i) + 4 * (l15 & 3);          // lane 4 q + pp of a 16-lane group supplies row q, columns 4 pp .. 4 pp + 3 of the block
        vaddr[i] = v_off<HS>(4 * g + (l15 >> 2), col >> 3) + ((col & 7) << 1);
    }
    unsigned vaddr_lds[8];                                      // the same as LDS byte addresses (the assembly reads take addresses, not pointers)
    const unsigned smem_lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#pragma unroll
    for (int i = 0; i < 8; ++i) vaddr_lds[i] = smem_lds0 + (unsigned)vaddr[i];
    const int wpos0 = p.pos_offset + wq0;                        // position of this wave's first row (uniform)
    const bool rows_ok = wq0 + 16 <= p.Tq;

    // Two barriers per tile.  B1: K(t) has landed everywhere and every wave is done with V(t - 1) -> request V(t), multiply K(t) Q^T under it.
    // B2: V(t) has landed everywhere and every wave is done with K(t) -> request K(t + 1), softmax and the PV product under it.
    // Double-buffered forms: one barrier per tile, tile t + 1 requested into the other buffer right behind it.
    // The tile body is instantiated per buffer (tiles alternate), so every LDS address is a per-lane offset + an immediate.
    stage_k(kt0, smem);
    if constexpr (DB) stage_v(kt0, smem + TILE_BYTES);
#ifdef MILA_FLASH_STAMPS
    const bool stamping = blockIdx.x == 0 && wave == 0;
    unsigned long long seg[5] = {0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
    const unsigned long long first_ = last_;
#endif
    // MODE 0 (double-buffered forms, main loop): the buffer is a template constant and the tile requests a WHOLE in-band successor of an unwrapped cache -- no
    //        branch, no address arithmetic: scalar tile offset + the lane's fixed offsets, one request at a time between the fragment reads.  (Round 4: the
    //        per-request choice of form cost ~ 56 of a tile's 74 scalar instructions -- a wave issues one instruction per four cycles, whatever its kind.)
    // MODE 1 (double-buffered forms, the last two or three tiles of a band, every tile of a wrapped ring): buffer by the tile's parity at run time; the
    //        successor, if any, is requested in the general form (per-row clamp / modulo) at the top of the tile.
    // MODE 2: the single-buffered forms (two barriers per tile).
    auto tile_body = [&](int t, auto buf_c, auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr int BUF = (MODE == 0) ? decltype(buf_c)::value : 0;
        const int kt = kt0 + t * kKeysPerTile;
        const int rbuf = (MODE == 1) ? (t & 1) * 2 * TILE_BYTES : BUF * 2 * TILE_BYTES;      // this tile's [K | V] pair
        unsigned char* ldsK = smem + rbuf;
        unsigned char* ldsV = ldsK + TILE_BYTES;
        unsigned char* nxt = smem + ((MODE == 1) ? 2 * TILE_BYTES - rbuf : (BUF ^ 1) * 2 * TILE_BYTES);
        const int ktn = kt + kKeysPerTile;
        const int adv = ktn * rstride;                                    // bytes, wave-uniform
        // MODE 0: request i of tile t + 1's K (V) rows, issued ONE AT A TIME between the fragment reads of this tile -- K requests among the K fragment reads, V
        // requests among the V^T fragment reads -- instead of all eight in a row behind the K reads: a workgroup's waves reach this point together, their 32
        // requests take the CU's address path (64 B / clk) ~ 500 cycles during which every wave stood in the issue queue with nothing else running; spread out,
        // the LDS reads and the requests feed two pipes at once.
        auto next_k = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (MODE == 0 && i < DMAS)
            {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr_t)(nxt + RPI * (NW * i + wave) * ROWB), 16, (NB == 2 && (i & 1)) ? ks1 : ks0,
                                                         adv + (i / NB) * 16 * rstride, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto next_v = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (MODE == 0 && i < DMAS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr_t)(nxt + TILE_BYTES + RPI * (NW * i + wave) * ROWB), 16, (NB == 2 && (i & 1)) ? vs1 : vs0,
                                                         adv + (i / NB) * 16 * rstride, 0, 0);
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows of tile t (NW = 4: its K rows; nothing else is in flight here)
        __syncthreads();
        FLASH_STAMP(0);
        if constexpr (MODE == 1)
        {
            if (t + 1 < ntiles) { stage_k(ktn, nxt); stage_v(ktn, nxt + TILE_BYTES); }
        }

        // ---- S^T = K Q^T : two 16-key groups; the fragments of KB k-steps are requested together, then multiplied ----
        f32x4 s0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
        f32x4 h0 = s0, h1 = s0;                                  // HALVES without the exchange (one wave per head): the upper half's product, added below
        constexpr int KB = (HS >= 512) ? 4 : 8;                  // k-steps whose fragments are fetched together (registers: HS = 512 holds 64 of Q)
        constexpr int KPER = (KOWN >= DMAS) ? KOWN / DMAS : 1;   // one K request behind every KPER k-steps of fragment reads
        static_assert((KOWN % DMAS == 0 || DMAS % KOWN == 0) && DT % DMAS == 0, "requests spread evenly over the fragment reads");
        constexpr int RPK = (DMAS > KOWN) ? DMAS / KOWN : 1;     // ... or RPK requests behind every k-step where a wave has more requests than k-steps
        auto k_group = [&](auto s8c) {
            constexpr int s8 = decltype(s8c)::value;
            s16x8 ka[KB], kb[KB];
            auto rd = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                // A operand: K[key l15 (+16)][32 s + 8 g .. +7] -> chunk 4 s + g
                constexpr int s_ = s8 + j;
                ka[j] = *reinterpret_cast<const s16x8*>(ldsK + kaddr[s_ & 3] + (s_ >> 2) * 256);
                kb[j] = *reinterpret_cast<const s16x8*>(ldsK + kaddr[s_ & 3] + (s_ >> 2) * 256 + 16 * ROWB);
                if constexpr ((s_ + 1) % KPER == 0)
                    static_for<RPK>([&](auto rc) { next_k(std::integral_constant<int, ((s_ + 1) / KPER - 1) * RPK + decltype(rc)::value>{}); });
            };
            static_for<KB>(rd);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE == 2 && s8 == 0) { stage_v(kt, ldsV); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (s8 == 0) FLASH_STAMP(1);
#pragma unroll
            for (int j = 0; j < KB; ++j)
            {
                if (HALVES && !XCH && s8 + j >= KSTEPS / 2)
                {
                    h0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), h0, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), h1, 0, 0, 0);
                }
                else
                {
                    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), s1, 0, 0, 0);
                }
            }
        };
        static_for<KOWN / KB>([&](auto gc) { k_group(std::integral_constant<int, decltype(gc)::value * KB>{}); });
        if constexpr (HALVES && !XCH) { s0 += h0; s1 += h1; }
        if constexpr (XCH)
        {
            // this wave's half of the scores out to LDS; the partner's (wave ^ HB: the other d-share of the same head and rows) back after the barrier
            f32x4* xw = reinterpret_cast<f32x4*>(smem + XCH_OFF) + (wave * 64 + lane) * 2;
            xw[0] = s0; xw[1] = s1;
        }
        if constexpr (MODE == 2)
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's V rows of tile t
            __syncthreads();                                     // every wave is done with K(t), and V(t) is complete
            if (t + 1 < ntiles) stage_k(kt + kKeysPerTile, ldsK);      // in flight during the softmax and the PV product
        }
        else if constexpr (XCH) __syncthreads();                 // (double-buffered forms: a second barrier per tile, for the exchange only)
        // ---- V^T fragments of the wave's d tiles, requested here so that they land under the softmax ----
        // A operand: V^T[dim 16 d + l15][keys]: two transposing reads (keys 4 g .. and 16 + 4 g ..)
        // (inline-assembly reads: the intrinsic form would wait here for the NEXT tile's LDS-DMA, see attention_tiles.h.  NOT in the form that keeps a whole
        // HS = 512 head in one wave -- it spills, and a register the compiler spills between an assembly read and its wait is stored before the data has landed;
        // tests/test_capi_cpu.py holds every form that uses the assembly reads to zero spills)
        s16x8 va[DT];
        if constexpr (!ASM_TR)
        {
#pragma unroll
            for (int d = 0; d < DT; ++d)
            {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ldsV + vaddr[d & 7] + (d >> 3) * 256));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ldsV + vaddr[d & 7] + (d >> 3) * 256 + 16 * ROWB));
                va[d][0] = lo[0]; va[d][1] = lo[1]; va[d][2] = lo[2]; va[d][3] = lo[3];
                va[d][4] = hi[0]; va[d][5] = hi[1]; va[d][6] = hi[2]; va[d][7] = hi[3];
            }
        }
        else
        {
            constexpr int VOFF = BUF * 2 * TILE_BYTES + TILE_BYTES;           // this buffer's V image
            constexpr int SPAN = ((DT - 1) >> 3) * 256 + 16 * ROWB;
            constexpr int DPER = DT / DMAS;                                   // one V request behind every DPER d tiles of fragment reads
            auto after = [&](auto dc) {
                constexpr int d = decltype(dc)::value;
                if constexpr ((d + 1) % DPER == 0) next_v(std::integral_constant<int, (d + 1) / DPER - 1>{});
            };
            if constexpr (MODE != 1 && VOFF + SPAN < 65536)
                read_vt_frags<ROWB, VOFF>(vaddr_lds, va, after, std::make_integer_sequence<int, DT>{});
            else
            {
                unsigned vb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) vb[i] = vaddr_lds[i] + (unsigned)(rbuf + TILE_BYTES);
                read_vt_frags<ROWB, 0>(vb, va, after, std::make_integer_sequence<int, DT>{});
            }
        }
        if constexpr (XCH)
        {
            const f32x4* xr = reinterpret_cast<const f32x4*>(smem + XCH_OFF) + ((wave ^ HB) * 64 + lane) * 2;
            s0 += xr[0]; s1 += xr[1];
        }
#ifdef MILA_FLASH_STAMPS
        if (stamping && s0[0] == 12345.678f) seg[4] += 1;          // the stamp waits for the products
#endif
        FLASH_STAMP(2);
        // lane holds keys kt + 4 g + r (s0) and kt + 16 + 4 g + r (s1) of query row l15
        float tv[8];
        // a tile every row of this wave sees whole (the interior of the band: most tiles) needs no mask
        const bool whole = rows_ok && kt + kKeysPerTile - 1 <= wpos0 && (p.window == 0 || kt > wpos0 + 15 - p.window);
        if (whole)
        {
#pragma unroll
            for (int r = 0; r < 8; ++r) tv[r] = ((r < 4) ? s0[r] : s1[r - 4]) * c2;
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                const int key = kt + ((r < 4) ? (4 * g + r) : (16 + 4 * g + (r - 4)));
                const float raw = (r < 4) ? s0[r] : s1[r - 4];
                const bool vis = row_valid && key <= my_pos && (p.window == 0 || key > my_pos - p.window);
                tv[r] = vis ? raw * c2 : -INFINITY;
            }
        }
        float alpha;
        const bf16x8 pfrag = softmax_tile_step(tv, m_run, l_run, alpha);
        // the running maximum of a row moves in its first tiles and then rarely: the accumulators are rescaled only in a tile where some row's did
        // (a real branch: alpha is exactly 1 everywhere otherwise, so skipping the multiplications changes no bit)
        if (__any(alpha != 1.0f))
        {
#pragma unroll
            for (int d = 0; d < DT; ++d) { o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha; }
            asm volatile("" ::: "memory");                     // keeps the block a branch target (no if-conversion into 4 DT selects)
        }
        FLASH_STAMP(3);
        if constexpr (ASM_TR) lds_tr_wait(va);
#pragma unroll
        for (int d = 0; d < DT; ++d)
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va[d]), pfrag, o[d], 0, 0, 0);
#ifdef MILA_FLASH_STAMPS
        if (stamping && o[DT - 1][0] == 12345.678f) seg[0] += 1;
#endif
        FLASH_STAMP(4);
    };
    // ---- PIPE: the software-pipelined loop of the double-buffered forms.  K runs ONE TILE AHEAD of V: iteration t multiplies K(t + 1) Q^T (matrix cores) while the
    // vector ALU runs the softmax step of tile t -- the two are independent, where the plain loop's tile is a chain K reads -> QK^T -> softmax -> PV with idle
    // matrix cores under the softmax -- and, under the exchange, the partner's half of S^T(t) was written a whole iteration earlier, so the exchange needs no
    // barrier of its own (one barrier per tile again; the exchange area is double-buffered by tile parity).
    //   iteration t:   wait + barrier | K(t + 2) -> K slot of buffer t & 1 (K(t) was multiplied in iteration t - 1), V(t + 1) -> V slot of buffer (t + 1) & 1
    //                  S(t + 1) = K(t + 1) Q^T from buffer (t + 1) & 1  ||  softmax(S(t))  |  O += V(t)^T P from buffer t & 1
    //   landed: K(t + 1) and V(t) were requested in iteration t - 1 and waited for at the top of iteration t.
    f32x4 sc0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, sc1 = sc0;         // S^T of the tile whose softmax comes next (this wave's half under the exchange)
    // S^T(tq) from the K image at `img` into (a0, a1); MODE 0 interleaves the requests of K(tq + 1)
    auto qk_tile = [&](unsigned char* img, f32x4& a0, f32x4& a1, auto reqs) {
        a0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; a1 = a0;
        f32x4 h0 = a0, h1 = a0;
        constexpr int KB = (HS >= 512) ? 4 : 8;
        constexpr int KPER = (KOWN >= DMAS) ? KOWN / DMAS : 1;
        constexpr int RPK = (DMAS > KOWN) ? DMAS / KOWN : 1;
        static_for<KOWN / KB>([&](auto gc) {
            constexpr int s8 = decltype(gc)::value * KB;
            s16x8 ka[KB], kb[KB];
            static_for<KB>([&](auto jc) {
                constexpr int s_ = s8 + decltype(jc)::value;
                ka[s_ - s8] = *reinterpret_cast<const s16x8*>(img + kaddr[s_ & 3] + (s_ >> 2) * 256);
                kb[s_ - s8] = *reinterpret_cast<const s16x8*>(img + kaddr[s_ & 3] + (s_ >> 2) * 256 + 16 * ROWB);
                if constexpr ((s_ + 1) % KPER == 0)
                    static_for<RPK>([&](auto rc) { reqs(std::integral_constant<int, ((s_ + 1) / KPER - 1) * RPK + decltype(rc)::value>{}); });
            });
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < KB; ++j)
            {
                if (HALVES && !XCH && s8 + j >= KSTEPS / 2)
                {
                    h0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), h0, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), h1, 0, 0, 0);
                }
                else
                {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), a1, 0, 0, 0);
                }
            }
        });
        if constexpr (HALVES && !XCH) { a0 += h0; a1 += h1; }
    };
    auto pipe_body = [&](int t, auto buf_c, auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;              // 0: lean (template buffer, branch-free requests); 1: buffer by parity at run time, general requests at the top
        constexpr int BUF = (MODE == 0) ? decltype(buf_c)::value : 0;
        const int kt = kt0 + t * kKeysPerTile;
        const int vbuf = (MODE == 1) ? (t & 1) * 2 * TILE_BYTES : BUF * 2 * TILE_BYTES;      // V(t) sits in the V slot of this buffer; its K slot takes K(t + 2)
        const int kbuf = 2 * TILE_BYTES - vbuf;                                               // K(t + 1) sits in the K slot of the other one; its V slot takes V(t + 1)
        unsigned char* ldsKn = smem + ((MODE == 1) ? kbuf : (BUF ^ 1) * 2 * TILE_BYTES);
        const bool more = t + 1 < ntiles;                          // (MODE 0: always)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        FLASH_STAMP(0);
        if constexpr (MODE == 1)
        {
            // requests at the top of the tile; a whole tile of an unwrapped cache by lane offsets + a scalar (one uniform branch per matrix), the band's last tile
            // and a wrapped ring in the general form
            auto top_requests = [&](auto isk, int tn, unsigned char* img) {
                constexpr bool ISK = decltype(isk)::value;
                const int ktn = kt0 + tn * kKeysPerTile;
                if (!ring && ktn + kKeysPerTile - 1 <= pos_last)
                {
#pragma unroll
                    for (int i = 0; i < DMAS; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(ISK ? rsK : rsV, (lds_ptr_t)(img + RPI * (NW * i + wave) * ROWB), 16,
                                                                 ISK ? ((NB == 2 && (i & 1)) ? ks1 : ks0) : ((NB == 2 && (i & 1)) ? vs1 : vs0),
                                                                 (ktn + (i / NB) * 16) * rstride, 0, 0);
                }
                else if constexpr (ISK) stage_k(ktn, img);
                else stage_v(ktn, img);
            };
            if (t + 2 < ntiles) top_requests(std::true_type{}, t + 2, smem + vbuf);
            if (more) top_requests(std::false_type{}, t + 1, smem + kbuf + TILE_BYTES);
        }
        if constexpr (XCH)
        {
            const f32x4* xr = reinterpret_cast<const f32x4*>(smem + XCH_OFF + (t & 1) * XCH_BYTES) + ((wave ^ HB) * 64 + lane) * 2;
            sc0 += xr[0]; sc1 += xr[1];
        }
        f32x4 n0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, n1 = n0;
        auto req_k = [&](auto ic) {                                // K(t + 2) -> K slot of V(t)'s buffer
            constexpr int i = decltype(ic)::value;
            if constexpr (MODE == 0 && i < DMAS)
            {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr_t)(smem + BUF * 2 * TILE_BYTES + RPI * (NW * i + wave) * ROWB), 16, (NB == 2 && (i & 1)) ? ks1 : ks0,
                                                         (kt + 2 * kKeysPerTile + (i / NB) * 16) * rstride, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto req_v = [&](auto ic) {                                // V(t + 1) -> V slot of K(t + 1)'s buffer
            constexpr int i = decltype(ic)::value;
            if constexpr (MODE == 0 && i < DMAS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr_t)(smem + (BUF ^ 1) * 2 * TILE_BYTES + TILE_BYTES + RPI * (NW * i + wave) * ROWB), 16,
                                                         (NB == 2 && (i & 1)) ? vs1 : vs0, (kt + kKeysPerTile + (i / NB) * 16) * rstride, 0, 0);
        };
        if (MODE == 0 || more)
        {
            qk_tile(ldsKn, n0, n1, req_k);
            if constexpr (XCH)
            {
                f32x4* xw = reinterpret_cast<f32x4*>(smem + XCH_OFF + ((t + 1) & 1) * XCH_BYTES) + (wave * 64 + lane) * 2;
                xw[0] = n0; xw[1] = n1;
            }
        }
        FLASH_STAMP(1);
        s16x8 va[DT];
        {
            constexpr int VOFF = BUF * 2 * TILE_BYTES + TILE_BYTES;
            constexpr int SPAN = ((DT - 1) >> 3) * 256 + 16 * ROWB;
            constexpr int DPER = DT / DMAS;
            auto after = [&](auto dc) {
                constexpr int d = decltype(dc)::value;
                if constexpr ((d + 1) % DPER == 0) req_v(std::integral_constant<int, (d + 1) / DPER - 1>{});
            };
            if constexpr (MODE != 1 && VOFF + SPAN < 65536)
                read_vt_frags<ROWB, VOFF>(vaddr_lds, va, after, std::make_integer_sequence<int, DT>{});
            else
            {
                unsigned vb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) vb[i] = vaddr_lds[i] + (unsigned)(vbuf + TILE_BYTES);
                read_vt_frags<ROWB, 0>(vb, va, after, std::make_integer_sequence<int, DT>{});
            }
        }
        FLASH_STAMP(2);
        float tv[8];
        // MODE 0 runs only tiles that every row of the workgroup sees whole (the driver below keeps the band's first two and last three tiles, and a workgroup
        // with rows past the chunk, in MODE 1): no mask, no branch -- the scores' scaling sits in ONE block with the K(t + 1) Q^T products in front of it
        const bool whole = MODE == 0 || (rows_ok && kt + kKeysPerTile - 1 <= wpos0 && (p.window == 0 || kt > wpos0 + 15 - p.window));
        if (whole)
        {
#pragma unroll
            for (int r = 0; r < 8; ++r) tv[r] = ((r < 4) ? sc0[r] : sc1[r - 4]) * c2;
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                const int key = kt + ((r < 4) ? (4 * g + r) : (16 + 4 * g + (r - 4)));
                const float raw = (r < 4) ? sc0[r] : sc1[r - 4];
                const bool vis = row_valid && key <= my_pos && (p.window == 0 || key > my_pos - p.window);
                tv[r] = vis ? raw * c2 : -INFINITY;
            }
        }
        float alpha;
        const bf16x8 pfrag = softmax_tile_step(tv, m_run, l_run, alpha);
        if (__any(alpha != 1.0f))
        {
#pragma unroll
            for (int d = 0; d < DT; ++d) { o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha; }
            asm volatile("" ::: "memory");
        }
        FLASH_STAMP(3);
        lds_tr_wait(va);
#pragma unroll
        for (int d = 0; d < DT; ++d)
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va[d]), pfrag, o[d], 0, 0, 0);
        sc0 = n0; sc1 = n1;
#ifdef MILA_FLASH_STAMPS
        if (stamping && o[DT - 1][0] == 12345.678f) seg[0] += 1;
#endif
        FLASH_STAMP(4);
    };
    if constexpr (PIPE)
    {
        static_assert(ASM_TR, "the pipelined forms read V^T by inline assembly");
        // prologue: K(1) on its way, S^T(0) from K(0) (both requested above in the general form; V(0) lands under the first product)
        if (ntiles > 1) stage_k(kt0 + kKeysPerTile, smem + 2 * TILE_BYTES);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        qk_tile(smem, sc0, sc1, [](auto) {});
        if constexpr (XCH)
        {
            f32x4* xw = reinterpret_cast<f32x4*>(smem + XCH_OFF) + (wave * 64 + lane) * 2;
            xw[0] = sc0; xw[1] = sc1;
        }
        // lean pairs (MODE 0): the tiles they request -- K(t + 2), K(t + 3), V(t + 1), V(t + 2) -- are whole tiles of an unwrapped cache (every tile but a band's
        // last one is), and the tiles they MASK are seen whole by every row of the workgroup: with <= 32 rows per workgroup that holds from the band's third
        // tile (a window's lower edge crosses at most the first two: 32 t >= QROWS + 30) to its fourth from last (the causal edge crosses at most the last two)
        static_assert(QROWS <= 32, "the whole-tile range of the lean loop is derived for <= 32 query rows per workgroup");
        int t = 0;
        if (!ring && q0 + QROWS <= p.Tq)
        {
            if (p.window > 0)
                for (; t < 2 && t < ntiles; ++t) pipe_body(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            for (; t + 5 <= ntiles; t += 2)
            {
                pipe_body(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                pipe_body(t + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            }
        }
        for (; t < ntiles; ++t) pipe_body(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    }
    else if constexpr (DB)
    {
        // every tile but the last one of a band is whole and in band; pairs whose requested successors (t + 1, t + 2) are such tiles run in the lean loop
        int t = 0;
        if (!ring)
            for (; t + 4 <= ntiles; t += 2)
            {
                tile_body(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                tile_body(t + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            }
        for (; t < ntiles; ++t) tile_body(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    }
    else
    {
        for (int t = 0; t < ntiles; ++t) tile_body(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    }
#ifdef MILA_FLASH_STAMPS
    if (stamping && lane == 0)
    {
        for (int i = 0; i < 5; ++i) g_flash_stamps[i] = seg[i];
        g_flash_stamps[5] = (unsigned long long)ntiles;
        g_flash_stamps[6] = __builtin_amdgcn_s_memtime() - first_;
    }
#endif

    // ---- epilogue: O^T[dim 16 d + 4 g + r][row l15] -> Y[row][h*HS + dim] ----
    const float l_row = quad_rows_sum(l_run);                   // the four lanes of a row
    if (row_valid)
    {
        const float inv = (l_row > 0.0f) ? 1.0f / l_row : 0.0f;
        uint16_t* y = p.Y + (((size_t)b * p.Tq + my_row) * p.NH + h) * HS + 16 * dsel * DT + 4 * g;
#pragma unroll
        for (int d = 0; d < DT; ++d)
            *reinterpret_cast<u32x2*>(y + 16 * d) = u32x2{pack_bf16x2(o[d][0] * inv, o[d][1] * inv), pack_bf16x2(o[d][2] * inv, o[d][3] * inv)};
    }
#ifdef MILA_FLASH_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamping && lane == 0) g_flash_stamps[7] = __builtin_amdgcn_s_memtime() - entry_;        // the stamped wave, entry to its last store retired
#endif
}

// PING-PONG form (round 4): one 8-wave workgroup per CU whose two wave groups (waves 0-3 / 4-7: the two waves of every SIMD) run HALF A TILE APART.
//   phase A(t): K fragments of tile t (ds_read_b128) -> S^T = K Q^T on the matrix cores
//   phase B(t): V^T fragments of tile t (transposing reads, requested first) -> softmax step on the vector ALU -> O^T += V^T P^T
//   slot 2t    : group 0 runs A(t),  group 1 runs B(t - 1)        slot 2t + 1 : group 0 runs B(t),  group 1 runs A(t)        one s_barrier between slots
// In the lockstep forms all waves of a workgroup read their fragments together (an LDS burst with idle matrix cores), multiply together, and run the softmax's
// dependent vector chain together (idle matrix cores again); two independent workgroups per CU only overlap by accident, and the kernel's time was the time of
// its heaviest workgroup, one wave per SIMD, 2 600 cycles per tile for 512 cycles of matrix work (SQ counters: 31 % of the wave cycles parked at a barrier or a
// waitcnt, 30 % issue stalls).  Here every SIMD always has one wave in a matrix phase while its partner reads, exponentiates or waits for LDS, and BOTH waves
// of a SIMD work on the heaviest item.  The same instructions per output element as the lockstep forms: identical bits.
// Tiles are double-buffered ([K | V] x 2); all eight waves share the request work: K(t + 1) is requested during slot 2t, V(t + 1) during slot 2t + 1,
// `s_waitcnt vmcnt(requests of the newest batch)` + the barrier at the end of a slot retire the batch before it.
//   WAR: K(t + 1) overwrites K(t - 1), last read in slot 2t - 1 (group 1's A(t - 1));  V(t + 1) overwrites V(t - 1), last read in slot 2t (group 1's B(t - 1)).
//   RAW: K(t + 1) is first read in slot 2t + 2, V(t + 1) in slot 2t + 3: each batch has a whole slot in flight behind the one that requests it.
// Workgroup = HB heads x DS d-shares x QB = 8 / (HB DS) blocks of 16 query rows: HS 256: 2 heads x 4 row blocks (group 1 owns the later rows);
// HS 512: 4 heads x 2 d-halves x 1 row block (group 1 owns the upper half of the output dimensions).
template <int HS, int HB, int DS>
__global__ __launch_bounds__(512, 1) void flash_prefill_pp_kernel(const FlashParams p)
{
    constexpr int NW = 8;
    constexpr int QB = NW / (HB * DS);
    static_assert((HB * DS) % 4 == 0 || (HB * DS) == 2, "a wave group holds whole (head, d-share) sets or whole row blocks");
    constexpr int QROWS = 16 * QB;
    constexpr int KSTEPS = HS / 32, DT = HS / 16 / DS, ROWB = HS * 2, TILE_BYTES = kKeysPerTile * ROWB;
    constexpr int RPI = 1024 / ROWB, CPR = ROWB / 16, DMAS = kKeysPerTile / NW / RPI;
    static_assert(RPI == 1 || RPI == 2, "HS = 512 or 256");
    constexpr int STEP = RPI * NW, NB = 16 / STEP;
    static_assert(NB == 1 || NB == 2, "whole 16-row groups per one or two requests");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x [K tile | V tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                 // waves w and w + 4 share a SIMD
    const int l15 = lane & 15, g = lane >> 4;
    const int GS = p.NH / p.NKV;
    const int item = blockIdx.x;                               // heavy -> light: the dispatcher hands the next item to the CU that frees up first
    const int qt = p.n_qtiles - 1 - item / p.n_hblk;
    const int hblk = item % p.n_hblk, b = blockIdx.z;
    const int h = hblk * HB + (wave % HB);
    const int dsel = (wave / HB) % DS;
    const int kvh = (hblk * HB) / GS;
    const int q0 = qt * QROWS;
    const int wq0 = q0 + 16 * (wave / (HB * DS));
    const int my_row = wq0 + l15;
    const bool row_valid = my_row < p.Tq;
    const int my_pos = p.pos_offset + (row_valid ? my_row : p.Tq - 1);

    s16x8 qf[KSTEPS];
    {
        const uint16_t* qp = p.Q + ((size_t)b * p.Tq + (row_valid ? my_row : 0)) * p.q_row_stride + (size_t)h * HS + 8 * g;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
        {
            const u32x4 v = row_valid ? ld16(qp + 32 * s) : u32x4{0u, 0u, 0u, 0u};
            qf[s] = __builtin_bit_cast(s16x8, v);
        }
    }
    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float m_run = -INFINITY, l_run = 0.0f;
    const float c2 = p.scale * 1.4426950408889634f;

    const int pos_first = p.pos_offset + q0;
    const int pos_last = p.pos_offset + min(q0 + QROWS, p.Tq) - 1;
    const int kmin = (p.window > 0) ? max(0, pos_first - p.window + 1) : 0;
    const int kt0 = kmin & ~(kKeysPerTile - 1);
    const int ntiles = (pos_last - kt0) / kKeysPerTile + 1;
    const uint16_t* kbase = p.K + (size_t)b * p.kv_b_stride + (size_t)kvh * p.kv_h_stride;
    const uint16_t* vbase = p.V + (size_t)b * p.kv_b_stride + (size_t)kvh * p.kv_h_stride;
    const bool ring = pos_last >= p.capacity;

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kbase), 0, 0x7fffffff, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vbase), 0, 0x7fffffff, 0x00020000);
    const int rstride = (int)(p.kv_r_stride * 2);
    int ks0, ks1, vs0, vs1;                                    // lane offsets of a request's rows inside a 16-row group (see flash_prefill_kernel_s1)
    {
        const int slot = lane % CPR;
        const int r0 = RPI * wave + lane / CPR, r1 = r0 + STEP;
        ks0 = r0 * rstride + (k_off<HS>(r0, slot) - r0 * ROWB);
        vs0 = r0 * rstride + (v_off<HS>(r0, slot) - r0 * ROWB);
        ks1 = r1 * rstride + (k_off<HS>(r1 & 15, slot) - (r1 & 15) * ROWB);
        vs1 = r1 * rstride + (v_off<HS>(r1 & 15, slot) - (r1 & 15) * ROWB);
    }
    // request i (0 .. DMAS - 1) of this wave's share of tile tn's K (V) rows into buffer tn & 1; nothing behind the last tile
    auto request = [&](auto isk, auto ic, int tn) {
        constexpr bool ISK = decltype(isk)::value;
        constexpr int i = decltype(ic)::value;
        if (tn >= ntiles) return;
        const int ktn = kt0 + tn * kKeysPerTile;
        unsigned char* img = smem + (tn & 1) * 2 * TILE_BYTES + (ISK ? 0 : TILE_BYTES);
        const int row0 = RPI * (NW * i + wave);
        if (!ring && ktn + kKeysPerTile - 1 <= pos_last)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ISK ? rsK : rsV, (lds_ptr_t)(img + row0 * ROWB), 16,
                                                     ISK ? ((NB == 2 && (i & 1)) ? ks1 : ks0) : ((NB == 2 && (i & 1)) ? vs1 : vs0),
                                                     (ktn + (i / NB) * 16) * rstride, 0, 0);
        else
        {
            int lz = lane;
            asm volatile("" : "+v"(lz));                       // (the last tiles' address arithmetic stays in place, see flash_prefill_kernel_s1)
            const int row = row0 + lz / CPR, slot = lz % CPR, pos = min(ktn + row, pos_last);
            const int voff = (ring ? pos % p.capacity : pos) * rstride + ((ISK ? k_off<HS>(row, slot) : v_off<HS>(row, slot)) - row * ROWB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ISK ? rsK : rsV, (lds_ptr_t)(img + row0 * ROWB), 16, voff, 0, 0, 0);
        }
    };
    int kaddr[4];
    unsigned vaddr_lds[8];
    {
        const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#pragma unroll
        for (int i = 0; i < 4; ++i) kaddr[i] = k_off<HS>(l15, 4 * i + g);
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
            const int col = 16 * (dsel * DT + i) + 4 * (l15 & 3);
            vaddr_lds[i] = smem_lds + (unsigned)(v_off<HS>(4 * g + (l15 >> 2), col >> 3) + ((col & 7) << 1));
        }
    }
    const int wpos0 = p.pos_offset + wq0;
    const bool rows_ok = wq0 + 16 <= p.Tq;

#ifdef MILA_FLASH_STAMPS
    const bool stamping = blockIdx.x == 0 && (wave & 3) == 0;    // wave 0 (group 0) and wave 4 (group 1) of the heaviest item: [A body, A slot end, B body, B slot end]
    unsigned long long seg[5] = {0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
    f32x4 s0, s1;                                              // S^T of the tile between its phase A and its phase B
    s16x8 va[DT];                                              // ... and its V^T fragments: requested at the end of phase A, they land across the barrier
    // ---- phase A(t): S^T = K Q^T, then the V^T fragment reads; requests of tile tn (REQ) issued between the fragment reads ----
    auto phase_a = [&](int t, auto buf_c, auto req_c, int tn) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr bool REQ = decltype(req_c)::value;
        unsigned char* ldsK = smem + BUF * 2 * TILE_BYTES;
        s0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; s1 = s0;
        f32x4 h0 = s0, h1 = s0;
        constexpr int KB = (HS >= 512) ? 4 : 8;
        constexpr int KPER = KSTEPS / DMAS;
        static_for<KSTEPS / KB>([&](auto gc) {
            constexpr int s8 = decltype(gc)::value * KB;
            s16x8 ka[KB], kb[KB];
            static_for<KB>([&](auto jc) {
                constexpr int s_ = s8 + decltype(jc)::value;
                ka[s_ - s8] = *reinterpret_cast<const s16x8*>(ldsK + kaddr[s_ & 3] + (s_ >> 2) * 256);
                kb[s_ - s8] = *reinterpret_cast<const s16x8*>(ldsK + kaddr[s_ & 3] + (s_ >> 2) * 256 + 16 * ROWB);
                if constexpr (REQ && (s_ + 1) % KPER == 0)
                {
                    __builtin_amdgcn_sched_barrier(0);
                    request(std::true_type{}, std::integral_constant<int, (s_ + 1) / KPER - 1>{}, tn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < KB; ++j)
            {
                if constexpr (HS >= 512 && s8 >= KSTEPS / 2)
                {
                    h0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), h0, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), h1, 0, 0, 0);
                }
                else
                {
                    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb[j]), __builtin_bit_cast(bf16x8, qf[s8 + j]), s1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        });
        if constexpr (HS >= 512) { s0 += h0; s1 += h1; }       // HS = 512: S^T is the sum of the two half-dimension products in every form (same bits as the exchanging forms)
        {
            constexpr int VOFF = BUF * 2 * TILE_BYTES + TILE_BYTES;
            constexpr int SPAN = ((DT - 1) >> 3) * 256 + 16 * ROWB;
            constexpr int DPER = DT / DMAS;
            auto after = [&](auto dc) {
                constexpr int d = decltype(dc)::value;
                if constexpr (REQ && (d + 1) % DPER == 0) request(std::false_type{}, std::integral_constant<int, (d + 1) / DPER - 1>{}, tn);
            };
            if constexpr (VOFF + SPAN < 65536)
                read_vt_frags<ROWB, VOFF>(vaddr_lds, va, after, std::make_integer_sequence<int, DT>{});
            else
            {
                unsigned vb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) vb[i] = vaddr_lds[i] + VOFF;
                read_vt_frags<ROWB, 0>(vb, va, after, std::make_integer_sequence<int, DT>{});
            }
        }
        // the fragments are in registers before the barrier: the next slot's requests overwrite this V image's partner, and the slot after it this one
        lds_tr_wait(va);
#ifdef MILA_FLASH_STAMPS
        if (stamping && s0[0] == 12345.678f && s1[0] == 12345.678f) seg[4] += 1;          // the stamp waits for the products
#endif
    };
    // ---- phase B(t): the softmax step, then O^T += V^T P^T; group 1 issues its share of tile tn's requests here (REQ) ----
    auto phase_b = [&](int t, auto req_c, int tn) {
        constexpr bool REQ = decltype(req_c)::value;
        const int kt = kt0 + t * kKeysPerTile;
        if constexpr (REQ)
        {
            static_for<DMAS>([&](auto ic) { request(std::true_type{}, ic, tn); });
            static_for<DMAS>([&](auto ic) { request(std::false_type{}, ic, tn); });
        }
        float tv[8];
        const bool whole = rows_ok && kt + kKeysPerTile - 1 <= wpos0 && (p.window == 0 || kt > wpos0 + 15 - p.window);
        if (whole)
        {
#pragma unroll
            for (int r = 0; r < 8; ++r) tv[r] = ((r < 4) ? s0[r] : s1[r - 4]) * c2;
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                const int key = kt + ((r < 4) ? (4 * g + r) : (16 + 4 * g + (r - 4)));
                const float raw = (r < 4) ? s0[r] : s1[r - 4];
                const bool vis = row_valid && key <= my_pos && (p.window == 0 || key > my_pos - p.window);
                tv[r] = vis ? raw * c2 : -INFINITY;
            }
        }
        float alpha;
        const bf16x8 pfrag = softmax_tile_step(tv, m_run, l_run, alpha);
        if (__any(alpha != 1.0f))
        {
#pragma unroll
            for (int d = 0; d < DT; ++d) { o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha; }
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int d = 0; d < DT; ++d)
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va[d]), pfrag, o[d], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
#ifdef MILA_FLASH_STAMPS
        if (stamping && o[DT - 1][0] == 12345.678f) seg[4] += 1;
#endif
    };
    // end of a slot.  EVEN slots carry the requests (K and V of the next tile: group 0 among its phase-A reads, group 1 at the top of its phase B), ODD slots
    // retire them: a batch has the rest of its slot and the whole next one in flight.
    auto slot_end = [&](bool retire) {
#ifndef MILA_FLASH_EXP_NOWAIT
        if (retire) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_s_barrier();
    };

    // prologue: K(0) and V(0) have landed everywhere before slot 0
    static_for<DMAS>([&](auto ic) { request(std::true_type{}, ic, 0); });
    static_for<DMAS>([&](auto ic) { request(std::false_type{}, ic, 0); });
    slot_end(true);
    constexpr std::integral_constant<int, 0> B0{};
    constexpr std::integral_constant<int, 1> B1{};
#ifdef MILA_FLASH_STAMPS
    last_ = __builtin_amdgcn_s_memtime();
#endif
    if (grp == 0)
    {
        // slot 2t: A(t) + the requests of tile t + 1; slot 2t + 1: B(t)
        for (int t = 0; t < ntiles; t += 2)
        {
            phase_a(t, B0, std::true_type{}, t + 1);       FLASH_STAMP(0); slot_end(false); FLASH_STAMP(1);
            phase_b(t, std::false_type{}, 0);              FLASH_STAMP(2); slot_end(true);  FLASH_STAMP(3);
            if (t + 1 < ntiles)
            {
                phase_a(t + 1, B1, std::true_type{}, t + 2);   FLASH_STAMP(0); slot_end(false); FLASH_STAMP(1);
                phase_b(t + 1, std::false_type{}, 0);          FLASH_STAMP(2); slot_end(true);  FLASH_STAMP(3);
            }
        }
        slot_end(false);                                       // slot 2 ntiles: group 1's last phase B
    }
    else
    {
        // slot 0: the requests of tile 1 only; slot 2t + 1: A(t); slot 2t + 2: B(t) + the requests of tile t + 2
        static_for<DMAS>([&](auto ic) { request(std::true_type{}, ic, 1); });
        static_for<DMAS>([&](auto ic) { request(std::false_type{}, ic, 1); });
        slot_end(false);
#ifdef MILA_FLASH_STAMPS
        last_ = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0; t < ntiles; t += 2)
        {
            phase_a(t, B0, std::false_type{}, 0);          FLASH_STAMP(0); slot_end(true);  FLASH_STAMP(1);
            phase_b(t, std::true_type{}, t + 2);           FLASH_STAMP(2); slot_end(false); FLASH_STAMP(3);
            if (t + 1 < ntiles)
            {
                phase_a(t + 1, B1, std::false_type{}, 0);      FLASH_STAMP(0); slot_end(true);  FLASH_STAMP(1);
                phase_b(t + 1, std::true_type{}, t + 3);       FLASH_STAMP(2); slot_end(false); FLASH_STAMP(3);
            }
        }
    }
#ifdef MILA_FLASH_STAMPS
    if (stamping && lane == 0)
    {
        for (int i = 0; i < 4; ++i) g_flash_stamps[8 + 4 * grp + i] = seg[i];
        g_flash_stamps[5] = (unsigned long long)ntiles;
    }
#endif

    const float l_row = quad_rows_sum(l_run);
    if (row_valid)
    {
        const float inv = (l_row > 0.0f) ? 1.0f / l_row : 0.0f;
        uint16_t* y = p.Y + (((size_t)b * p.Tq + my_row) * p.NH + h) * HS + 16 * dsel * DT + 4 * g;
#pragma unroll
        for (int d = 0; d < DT; ++d)
            *reinterpret_cast<u32x2*>(y + 16 * d) = u32x2{pack_bf16x2(o[d][0] * inv, o[d][1] * inv), pack_bf16x2(o[d][2] * inv, o[d][3] * inv)};
    }
}

static int g_tune_flash_form = 8;      // tuning "flash.form": 8 (default) = the LDS-DMA forms (HS 512: 8-wave workgroups, four heads x two d-halves; HS 256: double-buffered 4-wave
                                       // workgroups, two per CU); 9 = 8 with lockstep 8-wave workgroups at HS 256 too; 10 = the ping-pong 8-wave form at both head sizes; 11 = the
                                       // software-pipelined loop (K one tile ahead of V; HS 512: 2 % faster at the whole 160 KB of LDS, HS 256: 6 % slower); 2 = HS 512 as 4-wave
                                       // d-split workgroups; 1 = the register-staged kernels.  All give the same bits.
MILA_TUNE("flash.form", g_tune_flash_form);

// the register-staged kernels (HS <= 256; every head size under form 1)
template <int HS, int HB>
static int launch_flash(const FlashParams& p, int B, hipStream_t s)
{
    constexpr int QROWS = 16 * (4 / HB);
    const size_t lds = (size_t)(HS >= 512 ? 2 : 4) * kKeysPerTile * HS * 2;      // HS <= 256: two [K | V] buffers (HS = 512 here: one wave per head, no exchange area)
    FlashParams q = p;
    q.n_qtiles = (p.Tq + QROWS - 1) / QROWS;
    q.n_hblk = p.NH / HB;
    const dim3 grid(q.n_qtiles * q.n_hblk, 1, B);
    if constexpr (HS >= 512) hipLaunchKernelGGL((flash_prefill_kernel_s1<HS, HB, 1, 4>), grid, dim3(256), lds, s, q);
    else hipLaunchKernelGGL((flash_prefill_kernel<HS, HB>), grid, dim3(256), lds, s, q);
    MILA_LAUNCH_CHECK("flash_prefill");
}

// the LDS-DMA kernel (HS = 256 or 512): HB heads x DS d-shares x (NW / (HB DS)) row blocks per workgroup of NW waves
template <int HS, int HB, int DS, int NW, bool PIPE = false>
static int launch_flash_dma(const FlashParams& p, int B, hipStream_t s)
{
    constexpr int QROWS = 16 * (NW / (HB * DS));
    constexpr bool DB = (NW == 8) || (HS <= 256);
    const size_t lds = (size_t)(DB ? 4 : 2) * kKeysPerTile * HS * 2 + ((HS >= 512 && DS == 2) ? (size_t)NW * 2048 * (PIPE ? 2 : 1) : 0);      // + the score-exchange area(s)
    FlashParams q = p;
    q.n_qtiles = (p.Tq + QROWS - 1) / QROWS;
    q.n_hblk = p.NH / HB;
    const dim3 grid(q.n_qtiles * q.n_hblk, 1, B);
    if (lds > 65536)
    {
        // more than 64 KB of dynamic LDS must be allowed once per process (never inside a stream capture: the first prefill of a model is eager)
        static const hipError_t allowed = hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_prefill_kernel_s1<HS, HB, DS, NW, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (allowed != hipSuccess) return check_hip(allowed, "flash_prefill: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    hipLaunchKernelGGL((flash_prefill_kernel_s1<HS, HB, DS, NW, PIPE>), grid, dim3(64 * NW), lds, s, q);
    MILA_LAUNCH_CHECK("flash_prefill");
}

template <int HS, int HB, int DS>
static int launch_flash_pp(const FlashParams& p, int B, hipStream_t s)
{
    constexpr int QROWS = 16 * (8 / (HB * DS));
    const size_t lds = (size_t)4 * kKeysPerTile * HS * 2;
    FlashParams q = p;
    q.n_qtiles = (p.Tq + QROWS - 1) / QROWS;
    q.n_hblk = p.NH / HB;
    const dim3 grid(q.n_qtiles * q.n_hblk, 1, B);
    if (lds > 65536)
    {
        static const hipError_t allowed = hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_prefill_pp_kernel<HS, HB, DS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (allowed != hipSuccess) return check_hip(allowed, "flash_prefill: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    hipLaunchKernelGGL((flash_prefill_pp_kernel<HS, HB, DS>), grid, dim3(512), lds, s, q);
    MILA_LAUNCH_CHECK("flash_prefill");
}

template <int HS>
static int dispatch_hb(const FlashParams& p, int B, hipStream_t s)
{
    const int GS = p.NH / p.NKV;
    if constexpr (HS == 512) { if (g_tune_flash_form == 10 && GS % 4 == 0) return launch_flash_pp<HS, 4, 2>(p, B, s); }
    if constexpr (HS == 256) { if (g_tune_flash_form == 10 && GS % 2 == 0) return launch_flash_pp<HS, 2, 1>(p, B, s); }
    if constexpr (HS == 512) { if (g_tune_flash_form == 11 && GS % 4 == 0) return launch_flash_dma<HS, 4, 2, 8, true>(p, B, s); }
    if constexpr (HS == 256) { if (g_tune_flash_form == 11 && GS % 2 == 0) return launch_flash_dma<HS, 2, 1, 4, true>(p, B, s); }
    if constexpr (HS == 512)
    {
        if (g_tune_flash_form >= 8 && GS % 4 == 0) return launch_flash_dma<HS, 4, 2, 8>(p, B, s);      // four heads x two d-halves, double-buffered tiles
        if (g_tune_flash_form >= 2) return GS % 2 == 0 ? launch_flash_dma<HS, 2, 2, 4>(p, B, s) : launch_flash_dma<HS, 1, 2, 4>(p, B, s);
    }
    if constexpr (HS == 256)
    {
        // Double-buffered 4-wave workgroups, two per CU (their phases drift apart, so one's softmax runs under the other's products): 58 us on Gemma's
        // sliding-window shape at T = 2048.  The 8-wave form (2 heads x 4 row blocks on one tile stream, one workgroup per CU: 1.4x the work per tile load, but
        // its eight waves move in lockstep) was the faster one -- 76 vs 82 us -- while the tile body was bound by its vector ALU work; with the lean body it
        // is the slower one (62 us) and stays behind tuning form 9 where at least half the chunk's rows see a full window.  (A d-split does not pay at this
        // head size: 92.7 us.)
        if (g_tune_flash_form == 9 && GS % 2 == 0 && p.window > 0 && p.pos_offset + p.Tq / 2 >= p.window) return launch_flash_dma<HS, 2, 1, 8>(p, B, s);
        if (g_tune_flash_form >= 8)
        {
            if (GS % 4 == 0) return launch_flash_dma<HS, 4, 1, 4>(p, B, s);
            if (GS % 2 == 0) return launch_flash_dma<HS, 2, 1, 4>(p, B, s);
            return launch_flash_dma<HS, 1, 1, 4>(p, B, s);
        }
    }
    if (GS % 4 == 0) return launch_flash<HS, 4>(p, B, s);
    if (GS % 2 == 0) return launch_flash<HS, 2>(p, B, s);
    return launch_flash<HS, 1>(p, B, s);
}

int flash_dispatch(int HS, const FlashParams& p, int B, hipStream_t s)
{
    switch (HS)
    {
        case 64: return dispatch_hb<64>(p, B, s);
        case 128: return dispatch_hb<128>(p, B, s);
        case 256: return dispatch_hb<256>(p, B, s);
        case 512: return dispatch_hb<512>(p, B, s);
        default:
        {
            // any other head size (the reference tests' own HS = 4 / 8 geometries): the one-wave-per-row kernel of attention_generic.hip
            GenericAttnParams g{p.Y, p.Q, p.K, p.V, (int64_t)p.Tq * p.q_row_stride, p.q_row_stride, p.kv_b_stride, p.kv_h_stride, p.kv_r_stride,
                                B, p.Tq, p.NH, p.NKV, HS, p.capacity, p.pos_offset, p.window, p.scale};
            return launch_attn_generic(g, s);
        }
    }
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_attn_prefill_bf16(uint16_t* Y, const uint16_t* Q, const uint16_t* Kc, const uint16_t* Vc, int B, int chunk,
                                 int NH, int NKV, int HS, int capacity, int pos_offset, int window, float scale,
                                 mila_stream_t stream)
{
    MILA_REQUIRE(Y && Q && Kc && Vc, "attn_prefill_bf16: null pointer");
    MILA_REQUIRE(B > 0 && chunk > 0 && NH > 0 && NKV > 0 && NH % NKV == 0, "attn_prefill_bf16: bad sizes");
    MILA_REQUIRE(pos_offset >= 0 && capacity > 0 && window >= 0, "attn_prefill_bf16: bad positions");
    {
        // every key a query of this chunk may see must still be resident in the ring
        const int last = pos_offset + chunk - 1;
        const int oldest_needed = (window > 0) ? max(0, pos_offset - window + 1) : 0;
        MILA_REQUIRE(last - oldest_needed + 1 <= capacity,
                     "attn_prefill_bf16: keys [%d,%d] do not fit the cache capacity %d", oldest_needed, last, capacity);
    }
    FlashParams p;
    p.Y = Y; p.Q = Q; p.K = Kc; p.V = Vc;
    p.q_row_stride = (int64_t)NH * HS;
    p.kv_b_stride = (int64_t)NKV * capacity * HS;
    p.kv_h_stride = (int64_t)capacity * HS;
    p.kv_r_stride = HS;
    p.Tq = chunk; p.NH = NH; p.NKV = NKV; p.capacity = capacity; p.pos_offset = pos_offset; p.window = window;
    p.scale = scale;
    return flash_dispatch(HS, p, B, as_stream(stream));
}

int mila_cdna4_mha_bf16(uint16_t* Y, const uint16_t* QKV, int B, int T, int C, int NH, mila_stream_t stream)
{
    MILA_REQUIRE(Y && QKV, "mha_bf16: null pointer");
    MILA_REQUIRE(B > 0 && T > 0 && C > 0 && NH > 0 && C % NH == 0, "mha_bf16: bad sizes (B=%d T=%d C=%d NH=%d)", B, T, C, NH);
    const int HS = C / NH;
    FlashParams p;
    p.Y = Y; p.Q = QKV; p.K = QKV + C; p.V = QKV + 2 * C;
    p.q_row_stride = 3 * (int64_t)C;
    p.kv_b_stride = (int64_t)T * 3 * C;
    p.kv_h_stride = HS;
    p.kv_r_stride = 3 * (int64_t)C;
    p.Tq = T; p.NH = NH; p.NKV = NH; p.capacity = T; p.pos_offset = 0; p.window = 0;
    p.scale = 1.0f / sqrtf((float)HS);
    return flash_dispatch(HS, p, B, as_stream(stream));
}

}  // extern "C"

#ifdef MILA_FLASH_STAMPS
extern "C" MILA_API int mila_dbg_flash_stamps(unsigned long long* out8)
{
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(mila::g_flash_stamps), 16 * sizeof(unsigned long long));
}
#endif
