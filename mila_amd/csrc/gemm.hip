// Prefill (M > 1) Linear on the matrix cores:  Y[M,N] = X[M,K] * W[N,K]^T (+ bias), bf16 in/out,
// fp32 MFMA accumulate, for bf16 / fp8-per-channel / fp4-per-group weights.
//
// Replaces the cuBLASLt NT plans (OPS/Linear/CudaLinearOp.ixx:798-824) and, for quantized weights,
// the reference's 2-phase path -- dequantize the whole matrix to a bf16 scratch
// (Fp8Prefill/CudaFp8Prefill.cu:64-84, W4A16Gemm/CudaW4A16Gemm.cu:210-235), then GEMM -- by
// dequantizing each weight tile in registers on its way to LDS: the same arithmetic
// (w = bf16(decode(q) * scale), fp32 accumulate) without writing and re-reading N*K*2 bytes.
// Bias follows the reference's prefill order: the GEMM result is rounded to bf16 first, then
// bias is added in fp32 and rounded again (cuda_add_bias, CudaFp8Prefill.cu:239-256).
//
// Tile: 128 x 128 x 64 per 256-thread workgroup, 2 x 2 waves, each wave 64 x 64 as 2 x 2
// v_mfma_f32_32x32x16_bf16 accumulators.  The product is computed transposed (A operand = W tile,
// B operand = X tile) so that a lane's 4 consecutive accumulator registers are 4 consecutive
// output columns n of one row m -> 8-byte stores.  LDS rows are 128 B (64 bf16 of K) with the
// 16-byte slot index XOR-swizzled by (row >> 1) & 7, which makes every ds_read_b128 fragment
// read conflict-free (MI355X LDS: 64 banks x 4 B, 16-lane groups for b128).  Global->LDS staging
// goes through registers (the quantized formats must pass through VALU anyway), software
// pipelined one K-tile ahead with two LDS buffers and one barrier per K-tile.  Workgroup ids are
// remapped so that the workgroups sharing an XCD (id % 8) walk neighbouring tiles and share
// their W / X panels in that XCD's L2.
#include "common.h"
#include "internal.h"

namespace mila {

enum { G_BF16 = 0, G_FP8 = 1, G_FP4 = 2 };

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = 128 * BK * 2;   // one operand tile in LDS (bf16)

struct GemmParams
{
    uint16_t* Y;
    const uint16_t* X;
    const uint8_t* W;
    const float* scales;
    const uint16_t* bias;
    int M, K, N, group;
    int tiles_m, tiles_n;
    int act = 0;          // 1 = tanh-GELU on the stored Linear output (see Gemm256Params::act)
};

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// registers holding one thread's share of a K-tile on its way to LDS
template <int FMT> struct StageRegs;
template <> struct StageRegs<G_BF16> { u32x4 x[4]; u32x4 w[4]; };
template <> struct StageRegs<G_FP8> { u32x4 x[4]; u32x4 w[2]; };
template <> struct StageRegs<G_FP4> { u32x4 x[4]; u32x4 w[1]; float sc; };

template <int FMT>
__device__ __forceinline__ void stage_load(StageRegs<FMT>& r, const GemmParams& p, int m0, int n0, int k0)
{
    const int tid = threadIdx.x;
    // X tile: 128 rows x 8 slots(16 B); thread -> slots tid + 256*i
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int s = tid + 256 * i, row = s >> 3, slot = s & 7;
        const int m = m0 + row, k = k0 + slot * 8;
        r.x[i] = (m < p.M && k < p.K) ? ld16(p.X + (size_t)m * p.K + k) : u32x4{0u, 0u, 0u, 0u};
    }
    if constexpr (FMT == G_BF16)
    {
        const uint16_t* W = reinterpret_cast<const uint16_t*>(p.W);
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int s = tid + 256 * i, row = s >> 3, slot = s & 7;
            const int n = n0 + row, k = k0 + slot * 8;
            r.w[i] = (n < p.N && k < p.K) ? ld16(W + (size_t)n * p.K + k) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    else if constexpr (FMT == G_FP8)
    {
        // 128 rows x 4 segments of 16 fp8
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
            const int s = tid + 256 * i, row = s >> 2, seg = s & 3;
            const int n = n0 + row, k = k0 + seg * 16;
            r.w[i] = (n < p.N && k < p.K) ? ld16(p.W + (size_t)n * p.K + k) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    else
    {
        // 128 rows x 2 segments of 32 fp4 (16 bytes)
        const int row = tid >> 1, seg = tid & 1;
        const int n = n0 + row, k = k0 + seg * 32;
        const bool in = n < p.N && k < p.K;
        r.w[0] = in ? ld16(p.W + ((size_t)n * p.K + k) / 2) : u32x4{0u, 0u, 0u, 0u};
        r.sc = in ? p.scales[(size_t)n * (p.K / p.group) + k / p.group] : 0.0f;
    }
}

template <int FMT>
__device__ __forceinline__ void stage_store(const StageRegs<FMT>& r, const GemmParams& p, unsigned char* ldsX,
                                            unsigned char* ldsW, int n0)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int s = tid + 256 * i, row = s >> 3, slot = s & 7;
        *reinterpret_cast<u32x4*>(ldsX + swz(row, slot)) = r.x[i];
    }
    if constexpr (FMT == G_BF16)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int s = tid + 256 * i, row = s >> 3, slot = s & 7;
            *reinterpret_cast<u32x4*>(ldsW + swz(row, slot)) = r.w[i];
        }
    }
    else if constexpr (FMT == G_FP8)
    {
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
            const int s = tid + 256 * i, row = s >> 2, seg = s & 3;
            const int n = n0 + row;
            const float sc = (n < p.N) ? p.scales[n] : 0.0f;
            u32x4 lo, hi;
#pragma unroll
            for (int d = 0; d < 2; ++d)
            {
                const f32x2 a = fp8x2_to_f32x2(r.w[i][d], false), b = fp8x2_to_f32x2(r.w[i][d], true);
                lo[2 * d] = pack_bf16x2(a[0] * sc, a[1] * sc);
                lo[2 * d + 1] = pack_bf16x2(b[0] * sc, b[1] * sc);
                const f32x2 c = fp8x2_to_f32x2(r.w[i][d + 2], false), e = fp8x2_to_f32x2(r.w[i][d + 2], true);
                hi[2 * d] = pack_bf16x2(c[0] * sc, c[1] * sc);
                hi[2 * d + 1] = pack_bf16x2(e[0] * sc, e[1] * sc);
            }
            *reinterpret_cast<u32x4*>(ldsW + swz(row, seg * 2)) = lo;
            *reinterpret_cast<u32x4*>(ldsW + swz(row, seg * 2 + 1)) = hi;
        }
    }
    else
    {
        const int row = tid >> 1, seg = tid & 1;
        const float sc = r.sc;
#pragma unroll
        for (int d = 0; d < 4; ++d)
        {
            const uint32_t w = r.w[0][d];
            u32x4 o;
            const bf16x2 v0 = fp4x2_to_bf16x2<0>(w), v1 = fp4x2_to_bf16x2<1>(w), v2 = fp4x2_to_bf16x2<2>(w),
                         v3 = fp4x2_to_bf16x2<3>(w);
            o[0] = pack_bf16x2((float)v0[0] * sc, (float)v0[1] * sc);
            o[1] = pack_bf16x2((float)v1[0] * sc, (float)v1[1] * sc);
            o[2] = pack_bf16x2((float)v2[0] * sc, (float)v2[1] * sc);
            o[3] = pack_bf16x2((float)v3[0] * sc, (float)v3[1] * sc);
            *reinterpret_cast<u32x4*>(ldsW + swz(row, seg * 4 + d)) = o;
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 bufs][X tile | W tile]

    // XCD-aware tile order: workgroups with equal (id % 8) share an XCD/L2 -> give them
    // consecutive tiles along N within one M panel (bijective for any grid size)
    const int nwg = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
    const int tile = ((xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave >> 1, wm = wave & 1;        // wave's 64x64 sub-tile: n half, m half
    const int r32 = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];                               // [n tile][m tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int nk = (p.K + BK - 1) / BK;
    StageRegs<FMT> regs;
    stage_load<FMT>(regs, p, m0, n0, 0);
    stage_store<FMT>(regs, p, smem, smem + kTileBytes, n0);
    __syncthreads();

    for (int t = 0; t < nk; ++t)
    {
        unsigned char* cur = smem + (t & 1) * 2 * kTileBytes;
        unsigned char* nxt = smem + ((t + 1) & 1) * 2 * kTileBytes;
        const bool more = t + 1 < nk;
        if (more) stage_load<FMT>(regs, p, m0, n0, (t + 1) * BK);

        const unsigned char* lx = cur;
        const unsigned char* lw = cur + kTileBytes;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
        {
            // fragment: 8 bf16 of K starting at ks*16 + h*8  -> slot ks*2 + h
            s16x8 fw[2], fx[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
            {
                const int rw = wn * 64 + a * 32 + r32;
                fw[a] = *reinterpret_cast<const s16x8*>(lw + swz(rw, ks * 2 + h));
                const int rx = wm * 64 + a * 32 + r32;
                fx[a] = *reinterpret_cast<const s16x8*>(lx + swz(rx, ks * 2 + h));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, fw[a]), __builtin_bit_cast(bf16x8, fx[b]), acc[a][b], 0, 0, 0);
        }
        if (more) stage_store<FMT>(regs, p, nxt, nxt + kTileBytes, n0);
        __syncthreads();
    }

    // D[n][m]: lane -> column m = r32 of the (b) m-tile; registers -> rows n = (e&3) + 8*(e>>2) + 4*h
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
        {
            const int m = m0 + wm * 64 + b * 32 + r32;
            if (m >= p.M) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g)
            {
                const int n = n0 + wn * 64 + a * 32 + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
                if (p.bias)
                {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) v[e] = round_bf16(v[e]) + bf16_bits_to_f32(p.bias[n + e]);
                }
                if (p.act)
                {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(round_bf16(v[e]));
                }
                uint16_t* dst = p.Y + (size_t)m * p.N + n;
                if (n + 3 < p.N && (p.N & 3) == 0)
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                else
                {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) dst[e] = f32_to_bf16_bits(v[e]);
                }
            }
        }
}

template <int FMT>
static int launch_gemm(GemmParams p, hipStream_t s)
{
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int nwg = p.tiles_m * p.tiles_n;
    note_form(FMT == G_BF16 ? "gemm128" : (FMT == G_FP8 ? "gemm128_w8a16" : "gemm128_w4a16"));
    hipLaunchKernelGGL(gemm_kernel<FMT>, dim3(nwg), dim3(256), 4 * kTileBytes, s, p);
    MILA_LAUNCH_CHECK("gemm");
}

bool gemm256_applicable(int M, int K, int N);
int launch_gemm256(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act = 0, int ldy = 0);
int gemm_colsplit_main(int M, int K, int N, int* S_rest);      // gemm256.hip: columns the whole rounds of 256 x 256 tiles take (0 = no column split), split count of the rest
bool gemm256x128_applicable(int M, int K, int N);
bool gemm256x128_ragged_n_applicable(int M, int K, int N);
bool gemm256_ragged_n_applicable(int M, int K, int N);
int launch_gemm256x128(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act = 0);
bool gemm256_geglu_applicable(int M, int K, int F);
// gemm_skinny_bf16.hip: weight streaming for <= 64 rows per launch (any M as 64-row pieces)
int launch_gemm_bf16_skinny(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, int act, hipStream_t s);
int launch_gemm_bf16_skinny_geglu(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, hipStream_t s);
constexpr int kBf16SkinnyRows = 64;      // a remainder (or a whole prompt) of up to this many rows is a weight stream: the skinny kernel
int launch_gemm256_geglu(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, hipStream_t s);
static int g_gemm_force128 = 0;      // tuning "gemm.force128": 1 = always the 128 x 128 register-staged GEMM (A/B against the LDS-DMA kernels)
MILA_TUNE("gemm.force128", g_gemm_force128);
static int g_bf16_skinny = 1;      // tuning "gemm.bf16_skinny": the bf16 skinny kernel for <= 64-row prompts and remainders
MILA_TUNE("gemm.bf16_skinny", g_bf16_skinny);
extern int g_gemm_pingpong;     // gemm256.hip
extern int g_gemm_persistent;
extern int g_gemm_rowwise;
extern int g_gemm_fp8_tail_form;      // gemm_fp8_tail.hip
extern int g_skinny_whole_x;
extern int g_fp8_big_rule;           // gemm256.hip
extern int g_ldsdma_loose_tiles;     // gemm256.hip
extern int g_gemm_splitk;            // gemm256.hip
extern int g_fp8_splitk_min_rows;    // gemm256.hip
int gemm_splitk_for(int M, int K, int N);
int launch_gemm256x128_splitk(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act, float* partials, int S, int ldy = 0);
// Few rows (tools/experiments/few_row_rules.sh, profiles/r03_splitk.txt): bf16-policy prefill of 2 / 4 / 8 / 16 tokens with the skinny kernels ahead of the tile grids up
// to 16 rows 7.46 / 7.39 / 7.75 / 7.94 ms, never ahead 7.41 / 7.05 / 7.13 / 7.28 -- a one-round tile grid and the split-K form stream the weights at 3-4 TB/s from two
// rows on, the skinny kernel's 16-row groups at that rate only for one group.  The skinny kernels keep what has no such grid: a 1-row remainder, narrow outputs, calls
// without a workspace on the N = 3840 shapes.
int g_skinny_ahead_rows = 1;         // tuning "gemm.skinny_ahead_rows": up to this many rows the skinny kernels go ahead of an applicable tile grid (plain and GeGLU)
MILA_TUNE("gemm.skinny_ahead_rows", g_skinny_ahead_rows);
int gemm_fewrow_splits(int M, int K, int N);      // gemm_fewrow_bf16.hip
int launch_gemm_bf16_fewrow(float* partials, const uint16_t* X, const uint16_t* W, int M, int K, int N, int S, hipStream_t s);
int launch_splitk_reduce(uint16_t* Y, const float* partials, const uint16_t* bias, int M, int N, int S, int act, hipStream_t s);      // gemm256.hip
int g_fewrow = 1;                    // tuning "gemm.fewrow": the few-row weight-streaming form for <= 32 rows with a workspace
MILA_TUNE("gemm.fewrow", g_fewrow);
int g_splitk_min_rows = 2;           // tuning "gemm.splitk_min_rows": row counts below this stay off the split-K form even with a workspace
MILA_TUNE("gemm.splitk_min_rows", g_splitk_min_rows);

// which direct-to-LDS kernel serves a bf16-weight GEMM of this shape: 2 = 256 x 256, 1 = 256 x 128, 0 = none (128 x 128 register-staged)
static int glds_kernel_for(int M, int K, int N)
{
    if (g_gemm_force128) return 0;
    if (g_gemm_pingpong == 2 && gemm256x128_applicable(M, K, N)) return 1;      // tuning: the 256 x 128 ring wherever it applies
    if (gemm256_applicable(M, K, N)) return 2;
    if (gemm256x128_applicable(M, K, N)) return 1;
    if (g_gemm_pingpong >= 3 && gemm256_ragged_n_applicable(M, K, N)) return 2;
    if (gemm256x128_ragged_n_applicable(M, K, N)) return 1;
    return 0;
}
static int launch_glds(int which, uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act = 0)
{
    return which == 2 ? launch_gemm256(Y, X, W, bias, M, K, N, s, act) : launch_gemm256x128(Y, X, W, bias, M, K, N, s, act);
}
// Ragged prompt lengths (round 3): the LDS-DMA kernels take ANY M -- a ragged last tile-row stages row M - 1 for the rows past M and masks its stores (the fp8 forms'
// mechanism) -- so a 2049- or 2000-token prompt stays on them (bf16 policy: 60.6 / 62.5 ms -> one tile-row more / the 2048 time; the register-staged 128-tile
// kernel streamed a 1-row remainder's weights at 1 TB/s).  Where the whole M does not make an LDS-DMA grid, the leading multiple of 256 rows may still, and the
// rest goes to the 128-tile kernel (rows are independent; no padding, nothing read past the tensors).
// Returns the number of leading rows the LDS-DMA kernel serves (0 = none) and which kernel.
static int glds_rows_for(int M, int K, int N, int* which)
{
    *which = glds_kernel_for(M, K, N);
    if (*which) return M;
    const int main_rows = M - M % 256;
    if (main_rows >= 512 && (*which = glds_kernel_for(main_rows, K, N)) != 0) return main_rows;
    return 0;
}
static int launch_bf16_rows(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act = 0);

// ---- 2-phase staging (the reference's own prefill structure for quantized weights,
// OPS/Linear/CudaLinearOp.ixx:597-644, :716-764): dequantize the whole matrix to bf16 scratch, then the bf16 GEMM.
// Used when the 256 x 256 direct-to-LDS kernel applies: it reads its tiles with LDS-DMA and cannot dequantize.
// w = bf16(float(e4m3) * scale[n])   (Fp8Prefill/CudaFp8Prefill.cu:64-84)
__global__ __launch_bounds__(256) void dequant_fp8_kernel(uint16_t* __restrict__ out, const uint8_t* __restrict__ W,
                                                          const float* __restrict__ scales, int vec_per_row)
{
    // one workgroup per output channel (no index division): 16 weights per thread and pass
    const size_t row = blockIdx.x;
    const float sc = scales[row];
    const uint8_t* wrow = W + row * (size_t)vec_per_row * 16;
    uint16_t* orow = out + row * (size_t)vec_per_row * 16;
    for (int i = threadIdx.x; i < vec_per_row; i += 256)
    {
        const u32x4 w = ld16_nt(wrow + (size_t)i * 16);
        u32x4 lo, hi;
#pragma unroll
        for (int d = 0; d < 2; ++d)
        {
            const f32x2 a = fp8x2_to_f32x2(w[d], false), b = fp8x2_to_f32x2(w[d], true);
            lo[2 * d] = pack_bf16x2(a[0] * sc, a[1] * sc);
            lo[2 * d + 1] = pack_bf16x2(b[0] * sc, b[1] * sc);
            const f32x2 c = fp8x2_to_f32x2(w[d + 2], false), e = fp8x2_to_f32x2(w[d + 2], true);
            hi[2 * d] = pack_bf16x2(c[0] * sc, c[1] * sc);
            hi[2 * d + 1] = pack_bf16x2(e[0] * sc, e[1] * sc);
        }
        st16(orow + (size_t)i * 16, lo);
        st16(orow + (size_t)i * 16 + 8, hi);
    }
}
// w = bf16(lut[nibble] * scale[n, k / G])   (W4A16Gemm/CudaW4A16Gemm.cu:210-235); one 16-byte load = 32 elements
__global__ __launch_bounds__(256) void dequant_fp4_kernel(uint16_t* __restrict__ out, const uint8_t* __restrict__ W,
                                                          const float* __restrict__ scales, int vec_per_row, int vec_per_group_shift)
{
    // one workgroup per output channel; one 16-byte load = 32 elements; group index = vec >> shift (G / 32 = 2 or 4 vectors per group)
    const size_t row = blockIdx.x;
    const uint8_t* wrow = W + row * (size_t)vec_per_row * 16;
    const float* srow = scales + row * (size_t)(vec_per_row >> vec_per_group_shift);
    uint16_t* orow = out + row * (size_t)vec_per_row * 32;
    for (int i = threadIdx.x; i < vec_per_row; i += 256)
    {
        const float sc = srow[i >> vec_per_group_shift];
        const u32x4 w = ld16_nt(wrow + (size_t)i * 16);
#pragma unroll
        for (int d = 0; d < 4; ++d)
        {
            const bf16x2 v0 = fp4x2_to_bf16x2<0>(w[d]), v1 = fp4x2_to_bf16x2<1>(w[d]), v2 = fp4x2_to_bf16x2<2>(w[d]),
                         v3 = fp4x2_to_bf16x2<3>(w[d]);
            u32x4 o;
            o[0] = pack_bf16x2((float)v0[0] * sc, (float)v0[1] * sc);
            o[1] = pack_bf16x2((float)v1[0] * sc, (float)v1[1] * sc);
            o[2] = pack_bf16x2((float)v2[0] * sc, (float)v2[1] * sc);
            o[3] = pack_bf16x2((float)v3[0] * sc, (float)v3[1] * sc);
            st16(orow + (size_t)i * 32 + d * 8, o);
        }
    }
}

// dequantize a whole [N, K] weight to bf16 (the staging pass of the 2-phase prefill)
static int launch_dequant(int fmt, uint16_t* out, const uint8_t* W, const float* scales, int N, int K, int group, hipStream_t s)
{
    if (fmt == 1) hipLaunchKernelGGL(dequant_fp8_kernel, dim3(N), dim3(256), 0, s, out, W, scales, K / 16);
    else hipLaunchKernelGGL(dequant_fp4_kernel, dim3(N), dim3(256), 0, s, out, W, scales, K / 32, group == 128 ? 2 : 1);
    return check_hip(hipGetLastError(), fmt == 1 ? "dequant_fp8" : "dequant_fp4");
}

static int validate_gemm(const char* who, const void* Y, const void* X, const void* W, int M, int K, int N)
{
    MILA_REQUIRE(Y && X && W, "%s: null pointer", who);
    MILA_REQUIRE(M > 0 && K > 0 && N > 0, "%s: M, K, N must be positive (%d,%d,%d)", who, M, K, N);
    MILA_REQUIRE(K % 8 == 0, "%s: K=%d must be a multiple of 8 (16-byte rows)", who, K);
    return MILA_OK;
}

// bf16-weight GEMM over M rows: LDS-DMA kernel on the leading multiple of 256 rows when one applies, 128-tile kernel on the rest
static int launch_bf16_rows(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act)
{
    int which;
    if (!g_gemm_force128 && g_bf16_skinny)
    {
        // few rows: a weight stream (the 128-tile kernel pushed the bf16 model's weights past a 16-row prompt at 1 TB/s)
        // (past g_skinny_ahead_rows, and above all past one 16-row group -- the stream then re-reads X per 16 W rows and runs at 1.2-2 TB/s -- a tile grid that covers half
        // the chip or more is the faster form --
        // fc_gate_up at 64 rows 196 us against ~55 as 240 tiles of 256 x 128; on the N = 3840 shapes' 30 tiles the stream still wins)
        if (M <= kBf16SkinnyRows && (M <= g_skinny_ahead_rows || (int64_t)((N + 127) / 128) < kNumCU / 2 || !glds_kernel_for(M, K, N)))
            return launch_gemm_bf16_skinny(Y, X, W, bias, M, K, N, act, s);
        // a long prompt's <= 64-row remainder: the LDS-DMA kernels on the leading tile-rows, the skinny kernel on the rest (a ragged tile-row of the N = 3840 shapes
        // would open a second round of full-length tiles; the 128-tile kernel cost a 1-row remainder +18 ms per prefill)
        const int tail = M % 256;
        if (M >= 512 && tail > 0 && tail <= kBf16SkinnyRows && (which = glds_kernel_for(M - tail, K, N)) != 0)
        {
            int rc = launch_glds(which, Y, X, W, bias, M - tail, K, N, s, act);
            if (rc) return rc;
            return launch_gemm_bf16_skinny(Y + (size_t)(M - tail) * N, X + (size_t)(M - tail) * K, W, bias, tail, K, N, act, s);
        }
        // (a longer remainder, 65 .. 255 rows, rides in a ragged tile-row.  Where that row opens a new round of the grid -- the N = 3840 shapes: 240 tiles fill the chip, 270
        // need a second round of full-length tiles -- 64-row skinny passes were tried instead and are slower: the staged 64-row form streams at 1.2 TB/s, fc_down 4 x 90 us
        // against 180 for the extra round; profiles/r03_bf16_ragged.txt.  What that case wants is a split-K tail.)
    }
    const int main_rows = glds_rows_for(M, K, N, &which);
    if (main_rows > 0)
    {
        int rc = launch_glds(which, Y, X, W, bias, main_rows, K, N, s, act);
        if (rc || main_rows == M) return rc;
    }
    GemmParams p{Y + (size_t)main_rows * N, X + (size_t)main_rows * K, reinterpret_cast<const uint8_t*>(W), nullptr, bias, M - main_rows, K, N, 0, 0, 0, act};
    return launch_gemm<G_BF16>(p, s);
}

// ---- the same GEMM with a caller workspace (mila_cdna4_gemm_bf16_ws): what the split-K form changes ----
// plan: rows [0, main) as launch_bf16_rows serves them, rows [main, M) split-K with S copies (S = 0: no split-K part, everything as launch_bf16_rows)
struct Bf16WsPlan { int main_rows, S; bool fewrow; int n_main = 0, cs_rows = 0; };      // n_main > 0: the column split over rows [0, cs_rows): columns [0, n_main) on 256 x 256 tiles, the
                                                                                        // rest split-K with S copies; rows [cs_rows, M) (a <= 64-row remainder) as launch_bf16_rows serves them
// the split-K form for `rows` rows: up to 32 rows the few-row weight stream (gemm_fewrow_bf16.hip), else the 256 x 128 ring over S copies of the tile list
static Bf16WsPlan splitk_form(int main_rows, int rows, int K, int N)
{
    // (a remainder takes it only where the ring form could split too -- at most half a round of tiles: a wide output's remainder stays with the call's other forms,
    // so that a long prompt's fc_gate_up keeps its fused GeGLU kernel)
    if (g_fewrow && rows <= 32 && (main_rows == 0 || (N + 127) / 128 <= kNumCU / 2))
    {
        const int Sf = gemm_fewrow_splits(rows, K, N);
        if (Sf) return {main_rows, Sf, true};
    }
    return {main_rows, gemm_splitk_for(rows, K, N), false};
}
static Bf16WsPlan bf16_ws_plan(int M, int K, int N)
{
    if (g_gemm_force128 || M < g_splitk_min_rows) return {M, 0, false};
    // a short prompt: the whole tile list covers at most half the CUs (or the few-row form serves it)
    Bf16WsPlan pl = splitk_form(0, M, K, N);
    if (pl.S) return pl;
    // a tile list that ends in a nearly empty round (N = 8704 at T = 2048): whole rounds of 256 x 256 tiles + the remaining columns split-K (gemm256.hip: gemm_colsplit_main)
    // (a long prompt's <= 64-row remainder stays with the skinny kernel -- the dispatch ladder showed T = 2049 turning its ONE extra row into a ninth tile-row of the split)
    {
        const int tail_ = M % 256, rows = (M >= 512 && tail_ > 0 && tail_ <= kBf16SkinnyRows && g_bf16_skinny) ? M - tail_ : M;
        int S_rest = 0;
        const int n_main = gemm_colsplit_main(rows, K, N, &S_rest);
        if (n_main > 0 && gemm256_applicable(rows, K, n_main)) { Bf16WsPlan cs{0, S_rest, false}; cs.n_main = n_main; cs.cs_rows = rows; return cs; }
    }
    // a long prompt's remainder whose ragged tile-row would open another round of the grid (T = 2303 on the N = 3840 shapes: 240 tiles fill the chip, 270 run two
    // rounds of full-length tiles -- fc_down 200 -> 400 us): the whole tile-rows as before, the remainder split-K
    const int tail = M % 256, main_rows = M - tail;
    if (M < 512 || tail < g_splitk_min_rows) return {M, 0, false};
    const int which = glds_kernel_for(main_rows, K, N);
    if (!which) return {M, 0, false};
    const int per_row = which == 2 ? (N + 255) / 256 : (N + 127) / 128, tm = main_rows / 256;
    const bool new_round = (tm * per_row + kNumCU - 1) / kNumCU < ((tm + 1) * per_row + kNumCU - 1) / kNumCU;
    if (!new_round) return {M, 0, false};
    pl = splitk_form(main_rows, tail, K, N);
    return pl.S ? pl : Bf16WsPlan{M, 0, false};
}
static size_t bf16_ws_bytes(int M, int K, int N)
{
    const Bf16WsPlan pl = bf16_ws_plan(M, K, N);
    if (pl.n_main) return (size_t)pl.S * pl.cs_rows * (N - pl.n_main) * sizeof(float);
    return pl.S ? (size_t)pl.S * (M - pl.main_rows) * N * sizeof(float) : 0;
}
static int launch_bf16_rows_ws(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, hipStream_t s, int act, void* ws)
{
    const Bf16WsPlan pl = bf16_ws_plan(M, K, N);
    if (!pl.S) return launch_bf16_rows(Y, X, W, bias, M, K, N, s, act);
    if (pl.n_main)
    {
        note_form("gemm256_colsplit");
        const int R = pl.cs_rows;
        int rc = launch_gemm256(Y, X, W, bias, R, K, pl.n_main, s, act, N);
        if (rc) return rc;
        rc = launch_gemm256x128_splitk(Y + pl.n_main, X, W + (size_t)pl.n_main * K, bias ? bias + pl.n_main : nullptr, R, K, N - pl.n_main, s, act, static_cast<float*>(ws), pl.S, N);
        if (rc || R == M) return rc;
        return launch_bf16_rows(Y + (size_t)R * N, X + (size_t)R * K, W, bias, M - R, K, N, s, act);
    }
    if (pl.main_rows > 0)
    {
        int rc = launch_bf16_rows(Y, X, W, bias, pl.main_rows, K, N, s, act);
        if (rc) return rc;
    }
    uint16_t* Yt = Y + (size_t)pl.main_rows * N;
    const uint16_t* Xt = X + (size_t)pl.main_rows * K;
    const int rows = M - pl.main_rows;
    if (pl.fewrow)
    {
        note_form("fewrow_bf16");
        int rc = launch_gemm_bf16_fewrow(static_cast<float*>(ws), Xt, W, rows, K, N, pl.S, s);
        if (rc) return rc;
        return launch_splitk_reduce(Yt, static_cast<const float*>(ws), bias, rows, N, pl.S, act, s);
    }
    return launch_gemm256x128_splitk(Yt, Xt, W, bias, rows, K, N, s, act, static_cast<float*>(ws), pl.S);
}

// Linear + GeGLU over any row count the fused forms serve: the LDS-DMA GeGLU kernel on whole / ragged tile-rows, the skinny GeGLU kernel on <= 64 rows (a short prompt,
// or the remainder of a long one)
static bool geglu_rows_applicable(int M, int K, int F)      // what the fused kernels can run
{
    if (gemm256_geglu_applicable(M, K, F)) return true;
    if (!g_bf16_skinny) return false;
    if (M <= kBf16SkinnyRows) return true;                  // few rows: the skinny GeGLU kernel
    const int tail = M % 256;
    return M >= 512 && tail > 0 && tail <= kBf16SkinnyRows && gemm256_geglu_applicable(M - tail, K, F);
}
// ... and where they are also the faster choice for a caller that holds the gemm_bf16_ws workspace (RocmLinearOp / GemmaBlock): not where the plain GEMM over the
// [2F, K] weight would split K (that Linear + GeGLU pair is faster, and fused and unfused prefill keep identical bits: the fused kernels sum K in one order only), and,
// past g_skinny_ahead_rows, not where the plain Linear has an LDS-DMA grid (fc_gate_up at 64 rows: 196 us skinny, ~55 as a one-round tile grid + the elementwise pass)
static bool geglu_rows_preferred(int M, int K, int F)
{
    if (!geglu_rows_applicable(M, K, F) || bf16_ws_plan(M, K, 2 * F).S) return false;
    if (gemm256_geglu_applicable(M, K, F)) return true;
    if (M <= kBf16SkinnyRows) return M <= g_skinny_ahead_rows || !glds_kernel_for(M, K, 2 * F);
    return true;
}
static int launch_geglu_rows(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, hipStream_t s)
{
    const int tail = M % 256;
    const bool whole = gemm256_geglu_applicable(M, K, F);
    if (g_bf16_skinny && M <= kBf16SkinnyRows && (M <= g_skinny_ahead_rows || !whole)) return launch_gemm_bf16_skinny_geglu(Y, X, W, M, K, F, s);
    if (g_bf16_skinny && M >= 512 && tail > 0 && tail <= kBf16SkinnyRows && (tail <= g_skinny_ahead_rows || !whole) && gemm256_geglu_applicable(M - tail, K, F))
    {
        int rc = launch_gemm256_geglu(Y, X, W, M - tail, K, F, s);
        if (rc) return rc;
        return launch_gemm_bf16_skinny_geglu(Y + (size_t)(M - tail) * F, X + (size_t)(M - tail) * K, W, tail, K, F, s);
    }
    return launch_gemm256_geglu(Y, X, W, M, K, F, s);
}

}  // namespace mila

using namespace mila;

extern "C" {

int mila_cdna4_gemm_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N,
                         mila_stream_t stream)
{
    int rc = validate_gemm("gemm_bf16", Y, X, W, M, K, N);
    if (rc) return rc;
    return launch_bf16_rows(Y, X, W, bias, M, K, N, as_stream(stream));
}

int mila_cdna4_gemm_gelu_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, mila_stream_t stream)
{
    int rc = validate_gemm("gemm_gelu_bf16", Y, X, W, M, K, N);
    if (rc) return rc;
    return launch_bf16_rows(Y, X, W, bias, M, K, N, as_stream(stream), 1);
}

size_t mila_cdna4_gemm_workspace_bytes(int M, int K, int N)
{
    return (M > 0 && K > 0 && N > 0 && K % 8 == 0) ? bf16_ws_bytes(M, K, N) : 0;
}

int mila_cdna4_gemm_bf16_ws(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, int act, void* workspace, size_t workspace_bytes,
                            mila_stream_t stream)
{
    int rc = validate_gemm("gemm_bf16_ws", Y, X, W, M, K, N);
    if (rc) return rc;
    MILA_REQUIRE(act == 0 || act == 1, "gemm_bf16_ws: act must be 0 (none) or 1 (tanh-GELU), got %d", act);
    const size_t need = bf16_ws_bytes(M, K, N);
    if (need && (!workspace || workspace_bytes < need))
        return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_bf16_ws: workspace %zu bytes < required %zu (ask gemm_workspace_bytes)", workspace_bytes, need);
    MILA_REQUIRE(!need || (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "gemm_bf16_ws: the workspace must be 16-byte aligned");
    return launch_bf16_rows_ws(Y, X, W, bias, M, K, N, as_stream(stream), act, workspace);
}

int mila_cdna4_gemm_bf16_w8a16(uint16_t* Y, const uint16_t* X, const uint8_t* W, const float* scales,
                               const uint16_t* bias, int M, int K, int N, mila_stream_t stream)
{
    int rc = validate_gemm("gemm_bf16_w8a16", Y, X, W, M, K, N);
    if (rc) return rc;
    MILA_REQUIRE(scales != nullptr, "gemm_bf16_w8a16: per-channel scales are required");
    MILA_REQUIRE(K % 16 == 0, "gemm_bf16_w8a16: K=%d must be a multiple of 16", K);
    GemmParams p{Y, X, W, scales, bias, M, K, N, 0, 0, 0};
    return launch_gemm<G_FP8>(p, as_stream(stream));
}

int mila_cdna4_gemm_bf16_w4a16(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed, const float* scales,
                               const uint16_t* bias, int M, int K, int N, int group, mila_stream_t stream)
{
    int rc = validate_gemm("gemm_bf16_w4a16", Y, X, W_packed, M, K, N);
    if (rc) return rc;
    MILA_REQUIRE(scales != nullptr, "gemm_bf16_w4a16: per-group scales are required");
    MILA_REQUIRE(group == 64 || group == 128, "gemm_bf16_w4a16: group size must be 64 or 128 (got %d)", group);
    MILA_REQUIRE(K % group == 0, "gemm_bf16_w4a16: K=%d must be a multiple of the group size %d", K, group);
    MILA_REQUIRE(K % 32 == 0, "gemm_bf16_w4a16: K=%d must be a multiple of 32", K);
    GemmParams p{Y, X, W_packed, scales, bias, M, K, N, group, 0, 0};
    return launch_gemm<G_FP4>(p, as_stream(stream));
}

int mila_cdna4_dequantize_to_bf16(uint16_t* out, const void* W, const float* scales, int fmt, int N, int K, int group, mila_stream_t stream)
{
    MILA_REQUIRE(out && W && scales, "dequantize_to_bf16: null pointer");
    MILA_REQUIRE(fmt == 1 || fmt == 2, "dequantize_to_bf16: fmt must be 1 (fp8 per channel) or 2 (fp4 per group), got %d", fmt);
    MILA_REQUIRE(N > 0 && K > 0 && K % 32 == 0, "dequantize_to_bf16: bad sizes (N=%d K=%d)", N, K);
    if (fmt == 2) MILA_REQUIRE((group == 64 || group == 128) && K % group == 0, "dequantize_to_bf16: bad group size %d for K=%d", group, K);
    return launch_dequant(fmt, out, static_cast<const uint8_t*>(W), scales, N, K, group, as_stream(stream));
}

// the staged forms' scratch: the dequantized [N, K] bf16 weights, then (16-byte aligned: K % 8 == 0) the split-K workspace of the bf16 GEMM over them -- the staged call
// and gemm_bf16_ws on weights dequantized ahead of time (the host's resident prefill weights) then run the same kernels and give the same bits
size_t mila_cdna4_gemm_staging_bytes(int M, int K, int N)
{
    if (M <= 0 || K <= 0 || N <= 0 || K % 8 != 0) return 0;
    int which;
    const size_t ws = bf16_ws_bytes(M, K, N);
    if (ws || glds_rows_for(M, K, N, &which)) return (size_t)N * K * 2 + ws;
    // few rows: the staged forms dequantize once and stream the bf16 weights through the skinny kernel (the in-register-dequantizing 128-tile kernel pushed the fp8
    // policy's weights past a 16-row prompt at 1 TB/s)
    return (!g_gemm_force128 && g_bf16_skinny && M > 1 && M <= kBf16SkinnyRows) ? (size_t)N * K * 2 : 0;
}

int mila_cdna4_gemm_bf16_w8a16_staged(uint16_t* Y, const uint16_t* X, const uint8_t* W, const float* scales, const uint16_t* bias,
                                      int M, int K, int N, void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    const size_t need = mila_cdna4_gemm_staging_bytes(M, K, N);
    if (need == 0) return mila_cdna4_gemm_bf16_w8a16(Y, X, W, scales, bias, M, K, N, stream);
    int rc = validate_gemm("gemm_bf16_w8a16_staged", Y, X, W, M, K, N);
    if (rc) return rc;
    MILA_REQUIRE(scales != nullptr, "gemm_bf16_w8a16_staged: per-channel scales are required");
    if (!scratch || scratch_bytes < need)
        return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_bf16_w8a16_staged: scratch %zu bytes < required %zu", scratch_bytes, need);
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "gemm_bf16_w8a16_staged: the scratch must be 16-byte aligned");
    rc = launch_dequant(1, reinterpret_cast<uint16_t*>(scratch), W, scales, N, K, 0, as_stream(stream));
    if (rc) return rc;
    return launch_bf16_rows_ws(Y, X, reinterpret_cast<const uint16_t*>(scratch), bias, M, K, N, as_stream(stream), 0, static_cast<unsigned char*>(scratch) + (size_t)N * K * 2);
}

int mila_cdna4_gemm_bf16_w4a16_staged(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed, const float* scales,
                                      const uint16_t* bias, int M, int K, int N, int group, void* scratch, size_t scratch_bytes,
                                      mila_stream_t stream)
{
    const size_t need = mila_cdna4_gemm_staging_bytes(M, K, N);
    if (need == 0) return mila_cdna4_gemm_bf16_w4a16(Y, X, W_packed, scales, bias, M, K, N, group, stream);
    int rc = validate_gemm("gemm_bf16_w4a16_staged", Y, X, W_packed, M, K, N);
    if (rc) return rc;
    MILA_REQUIRE(scales != nullptr, "gemm_bf16_w4a16_staged: per-group scales are required");
    MILA_REQUIRE(group == 64 || group == 128, "gemm_bf16_w4a16_staged: group size must be 64 or 128 (got %d)", group);
    MILA_REQUIRE(K % group == 0, "gemm_bf16_w4a16_staged: K=%d must be a multiple of the group size %d", K, group);
    if (!scratch || scratch_bytes < need)
        return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_bf16_w4a16_staged: scratch %zu bytes < required %zu", scratch_bytes, need);
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "gemm_bf16_w4a16_staged: the scratch must be 16-byte aligned");
    rc = launch_dequant(2, reinterpret_cast<uint16_t*>(scratch), W_packed, scales, N, K, group, as_stream(stream));
    if (rc) return rc;
    return launch_bf16_rows_ws(Y, X, reinterpret_cast<const uint16_t*>(scratch), bias, M, K, N, as_stream(stream), 0, static_cast<unsigned char*>(scratch) + (size_t)N * K * 2);
}

/* ---- Linear + GeGLU in one kernel (prefill fc_gate_up): Y[M, F] = GeGLU(X W^T), W = [gate | up] rows ---- */
int mila_cdna4_gemm_geglu_applicable(int M, int K, int F)
{
    return (M > 0 && K > 0 && F > 0 && K % 8 == 0 && !g_gemm_force128 && geglu_rows_applicable(M, K, F)) ? 1 : 0;
}
// for a caller that HOLDS the gemm_bf16_ws workspace: where the plain GEMM over the [2F, K] weight would split K (few-row prompts, short tile lists), its
// Linear (gemm_bf16_ws) + GeGLU pair is the faster one and the fused form steps aside -- a host decision (it used to hide inside gemm_geglu_applicable,
// so a caller WITHOUT a workspace lost the fused kernels on those shapes for nothing)
int mila_cdna4_gemm_geglu_preferred(int M, int K, int F)
{
    return (mila_cdna4_gemm_geglu_applicable(M, K, F) && geglu_rows_preferred(M, K, F)) ? 1 : 0;
}

int mila_cdna4_gemm_geglu_bf16(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, mila_stream_t stream)
{
    int rc = validate_gemm("gemm_geglu_bf16", Y, X, W, M, K, F);
    if (rc) return rc;
    MILA_REQUIRE(geglu_rows_applicable(M, K, F), "gemm_geglu_bf16: shape (M=%d, K=%d, F=%d) is outside the fused kernel (ask gemm_geglu_applicable)", M, K, F);
    return launch_geglu_rows(Y, X, W, M, K, F, as_stream(stream));
}

int mila_cdna4_gemm_geglu_bf16_w8a16_staged(uint16_t* Y, const uint16_t* X, const uint8_t* W, const float* scales, int M, int K, int F,
                                            void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    int rc = validate_gemm("gemm_geglu_bf16_w8a16_staged", Y, X, W, M, K, F);
    if (rc) return rc;
    MILA_REQUIRE(geglu_rows_applicable(M, K, F), "gemm_geglu_bf16_w8a16_staged: shape (M=%d, K=%d, F=%d) is outside the fused kernel", M, K, F);
    MILA_REQUIRE(scales != nullptr, "gemm_geglu_bf16_w8a16_staged: per-channel scales are required");
    const size_t need = (size_t)2 * F * K * 2;
    if (!scratch || scratch_bytes < need)
        return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_geglu_bf16_w8a16_staged: scratch %zu bytes < required %zu", scratch_bytes, need);
    rc = launch_dequant(1, reinterpret_cast<uint16_t*>(scratch), W, scales, 2 * F, K, 0, as_stream(stream));
    if (rc) return rc;
    return launch_geglu_rows(Y, X, reinterpret_cast<const uint16_t*>(scratch), M, K, F, as_stream(stream));
}

int mila_cdna4_gemm_geglu_bf16_w4a16_staged(uint16_t* Y, const uint16_t* X, const uint8_t* W_packed, const float* scales, int M, int K,
                                            int F, int group, void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    int rc = validate_gemm("gemm_geglu_bf16_w4a16_staged", Y, X, W_packed, M, K, F);
    if (rc) return rc;
    MILA_REQUIRE(geglu_rows_applicable(M, K, F), "gemm_geglu_bf16_w4a16_staged: shape (M=%d, K=%d, F=%d) is outside the fused kernel", M, K, F);
    MILA_REQUIRE(scales != nullptr, "gemm_geglu_bf16_w4a16_staged: per-group scales are required");
    MILA_REQUIRE(group == 64 || group == 128, "gemm_geglu_bf16_w4a16_staged: group size must be 64 or 128 (got %d)", group);
    MILA_REQUIRE(K % group == 0, "gemm_geglu_bf16_w4a16_staged: K=%d must be a multiple of the group size %d", K, group);
    const size_t need = (size_t)2 * F * K * 2;
    if (!scratch || scratch_bytes < need)
        return set_error(MILA_E_SCRATCH_TOO_SMALL, "gemm_geglu_bf16_w4a16_staged: scratch %zu bytes < required %zu", scratch_bytes, need);
    rc = launch_dequant(2, reinterpret_cast<uint16_t*>(scratch), W_packed, scales, 2 * F, K, group, as_stream(stream));
    if (rc) return rc;
    return launch_geglu_rows(Y, X, reinterpret_cast<const uint16_t*>(scratch), M, K, F, as_stream(stream));
}

}  // extern "C"
