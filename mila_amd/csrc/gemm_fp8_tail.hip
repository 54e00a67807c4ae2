// fp8 x fp8 (e4m3) GEMM for the rows the LDS-DMA kernels of gemm256.hip do not take: ragged prompt lengths (M % 256 != 0), short prompts,
// the last chunk of a chunked prefill, and shapes whose N / K are outside the big tiles.  With it the fp4 policy's prefill runs W4A8 for
// EVERY M > 1, as the reference does (CudaLinearOp.ixx:646-715: kUseFp8ActivationPrefillPath is unconditional), instead of changing
// arithmetic with the prompt length.
//
//   Y[M, N] = bf16( float( bf16( (X8 W8^T)[m, n] * *w_scale ) ) * x_scales[m] + bias[n] )        (CudaFp8Prefill.cu:191-211)
//
// Bit-identical to gemm256_kernel<G_FP8> / gemm256x128_kernel<true> on the same rows: one v_mfma_scale_f32_16x16x128_f8f6f4 (unit block
// scales) per 16 x 16 sub-tile and 128-byte K-tile, K-tiles in ascending order, A operand = W rows, B operand = X rows, and a lane
// (l15, g) supplies the same 32 bytes k = 32 g .. 32 g + 31 of its row -- so an output element is the same chain of the same
// instruction on the same operand bytes whichever kernel computes it (tests/test_linear_gpu.py holds the two against each other).
//
// Tile: 128 X rows x BN W rows x 128 bytes of K, 256 threads = 4 waves as WN (W side) x 4 / WN (X side).  Rows past M / N and K
// chunks past K are staged as zeros (e4m3 0x00 = +0: the products vanish) and never stored.  Register staging (masked loads cannot be
// LDS-DMA), two LDS buffers, one barrier per K-tile; the kernel is bound by reading W once per 128 rows of M, so BN is picked by the
// launcher to give the chip enough workgroups (N = 3840 at BN = 128 would be 30).
// LDS image: 128-byte rows, source chunk c (16 bytes) at logical slot ((c & 1) << 2) | (c >> 1) -- the two halves of a lane's 32-byte
// operand at slots g and 4 + g, the conflict-free pattern of gemm256.hip -- XOR-swizzled by (row >> 1) & 7.
#include <algorithm>

#include "common.h"

namespace mila {

struct Fp8TailParams
{
    uint16_t* Y;
    const uint8_t* X;         // [M, K] e4m3
    const uint8_t* W;         // [N, K] e4m3 (GEGLU: [2 F, K], gate rows then up rows, N = F)
    const uint16_t* bias;
    const float* x_scales;    // [M]
    const float* w_scale;     // device scalar (W4A8), or with w_pc the per-channel vector over the W rows (W8A8; GEGLU: [gate rows | up rows])
    int M, K, N, tiles_m, tiles_n;
    int w_pc = 0;
};
typedef int i32x8t __attribute__((ext_vector_type(8)));

template <int WN, int PT, int QT, bool GEGLU>
__global__ __launch_bounds__(256) void gemm_fp8_tail_kernel(const Fp8TailParams p)
{
    constexpr int WM = 4 / WN;
    constexpr int BN = WN * PT * 16;                  // W rows per tile
    constexpr int BM = WM * QT * 16;                  // X rows per tile
    static_assert(BM == 128, "the X tile is 128 rows");
    constexpr int BO = GEGLU ? BN / 2 : BN;           // output columns per tile
    constexpr int kXBytes = BM * 128, kWBytes = BN * 128, kBuf = kXBytes + kWBytes;
    constexpr int NX = BM * 8 / 256, NW = BN * 8 / 256;      // 16-byte chunks per thread and K-tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kBuf];

    const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;      // the M-tiles of one W panel run side by side
    const int m0 = tm * BM, n0 = tn * BO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const int l15 = lane & 15, g = lane >> 4;
    const int K = p.K, nk = (K + 127) / 128;

    // W tile row -> global W row and whether it exists
    auto w_row = [&](int row, bool& ok) -> size_t {
        if constexpr (GEGLU)
        {
            constexpr int HALF = PT * 8;                                      // a wave's rows: HALF gate rows, then the matching HALF up rows
            const int w = row / (PT * 16), in = row % (PT * 16);
            const int col = n0 + w * HALF + (in % HALF);
            ok = col < p.N;
            return (size_t)(in < HALF ? col : p.N + col);
        }
        else
        {
            ok = n0 + row < p.N;
            return (size_t)(n0 + row);
        }
    };
    auto lds_off = [](int row, int c) { return row * 128 + (((((c & 1) << 2) | (c >> 1)) ^ ((row >> 1) & 7)) << 4); };

    // two named register sets: the K-tile staged to LDS at the end of iteration t was requested at iteration t - 2 (two K-tiles of latency cover; with one
    // set -- requested at t, stored at t -- a 208-row tail ran the tile at ~1 us per K-tile, the global latency laid bare)
    struct Regs { u32x4 x[NX], w[NW]; };
    Regs ra, rb;
    auto stage_load = [&](Regs& r, int kt) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
        {
            const int s = tid + 256 * i, row = s >> 3, c = s & 7;
            const int m = m0 + row, k = kt * 128 + c * 16;
            r.x[i] = (m < p.M && k < K) ? ld16(p.X + (size_t)m * K + k) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NW; ++i)
        {
            const int s = tid + 256 * i, row = s >> 3, c = s & 7;
            const int k = kt * 128 + c * 16;
            bool ok;
            const size_t n = w_row(row, ok);
            r.w[i] = (ok && k < K) ? ld16(p.W + n * K + k) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto stage_store = [&](const Regs& r, unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
        {
            const int s = tid + 256 * i;
            *reinterpret_cast<u32x4*>(buf + lds_off(s >> 3, s & 7)) = r.x[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i)
        {
            const int s = tid + 256 * i;
            *reinterpret_cast<u32x4*>(buf + kXBytes + lds_off(s >> 3, s & 7)) = r.w[i];
        }
    };

    f32x4 acc[PT][QT];
#pragma unroll
    for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int b = 0; b < QT; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    auto compute = [&](const unsigned char* cur) {
        struct Pair { s16x8 lo, hi; };
        i32x8t fa[PT], fb[QT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
        {
            const int r = wn * PT * 16 + pt * 16 + l15;
            const unsigned char* rowp = cur + kXBytes + r * 128;
            const int sw = (r >> 1) & 7;
            fa[pt] = __builtin_bit_cast(i32x8t, (Pair{*reinterpret_cast<const s16x8*>(rowp + ((g ^ sw) << 4)), *reinterpret_cast<const s16x8*>(rowp + (((4 + g) ^ sw) << 4))}));
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
        {
            const int r = wm * QT * 16 + qt * 16 + l15;
            const unsigned char* rowp = cur + r * 128;
            const int sw = (r >> 1) & 7;
            fb[qt] = __builtin_bit_cast(i32x8t, (Pair{*reinterpret_cast<const s16x8*>(rowp + ((g ^ sw) << 4)), *reinterpret_cast<const s16x8*>(rowp + (((4 + g) ^ sw) << 4))}));
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
                acc[pt][qt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[pt], fb[qt], acc[pt][qt], 0, 0, 0, 127, 0, 127);
    };

    // prologue: K-tile 0 in LDS buffer 0; set A holds K-tile 1, set B K-tile 2 (both in flight)
    stage_load(ra, 0);
    stage_store(ra, smem);
    if (nk > 1) stage_load(ra, 1);
    if (nk > 2) stage_load(rb, 2);
    __syncthreads();
    for (int t = 0; t < nk; t += 2)
    {
        // even K-tile t (buffer 0): A -> buffer 1 is K-tile t + 1, then A requests t + 3
        compute(smem);
        if (t + 1 < nk) stage_store(ra, smem + kBuf);
        if (t + 3 < nk) stage_load(ra, t + 3);
        __syncthreads();
        if (t + 1 >= nk) break;
        // odd K-tile t + 1 (buffer 1): B -> buffer 0 is K-tile t + 2, then B requests t + 4
        compute(smem + kBuf);
        if (t + 2 < nk) stage_store(rb, smem);
        if (t + 4 < nk) stage_load(rb, t + 4);
        __syncthreads();
    }

    // ---- epilogue: D[p = 4 g + e][q = l15] -> Y[m0 + .. + q][n0 + .. + p] ----
    const bool pc = p.w_pc != 0;
    const float ws_scalar = pc ? 1.0f : *p.w_scale;
    // this lane's weight scales for columns n .. n + 3 (`up`: the GeGLU up rows, F + n); columns past N are never stored
    auto wscale4 = [&](int n, bool up, float (&w)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pc ? p.w_scale[(up ? p.N : 0) + min(n + e, p.N - 1)] : ws_scalar;
    };
    const bool vec_ok = (p.N & 3) == 0;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
    {
        const int m = m0 + wm * QT * 16 + qt * 16 + l15;
        if (m >= p.M) continue;
        const float ts = p.x_scales[m];
        uint16_t* yrow = p.Y + (size_t)m * p.N;
#pragma unroll
        for (int pt = 0; pt < (GEGLU ? PT / 2 : PT); ++pt)
        {
            const int n = n0 + wn * (GEGLU ? PT * 8 : PT * 16) + pt * 16 + 4 * g;
            if (n >= p.N) continue;
            float v[4], wg[4];
            wscale4(n, false, wg);
            if constexpr (GEGLU)
            {
                float wu[4];
                wscale4(n, true, wu);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = gelu_tanh(fp8_linear_out(pc, acc[pt][qt][e], wg[e], ts)) * fp8_linear_out(pc, acc[pt + PT / 2][qt][e], wu[e], ts);
            }
            else
            {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    v[e] = fp8_scale_bias(pc, acc[pt][qt][e], wg[e], ts, p.bias != nullptr, (p.bias && n + e < p.N) ? bf16_bits_to_f32(p.bias[n + e]) : 0.0f);
                }
            }
            if (vec_ok) *reinterpret_cast<u32x2*>(yrow + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            else
            {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.N) yrow[n + e] = f32_to_bf16_bits(v[e]);
            }
        }
    }
}

// ---- the SKINNY form: M <= 64 rows (a prompt of 2049 tokens leaves a 1-row tail; short prompts; small chunk tails) --------------------------------------
// At these row counts the GEMM is a weight stream: every byte of W is read once for 2 M FLOP, and the 128-row LDS tile above -- load, LDS round trip,
// barrier, one K-tile in flight -- moves W at ~1 TB/s (a T = 2049 prefill cost 1.31x a T = 2048 one).  Here W never touches LDS:
//   * a workgroup owns 16 W rows (GEGLU: 16 gate + the matching 16 up rows) and ALL of K; its 8 waves take the K-tiles 8 s + w of step s, so a step of the
//     workgroup covers 1 KiB of every row and the eight partial sums meet in LDS once, at the end, in wave order (a fixed order: the result does not depend
//     on M or on the grid);
//   * a lane's MFMA A operand IS its two 16-byte global loads (row l15, bytes 32 g .. 32 g + 31 of the K-tile), non-temporal, requested PF steps ahead
//     (8 waves x PF x 2 KiB in flight per workgroup, several workgroups per CU);
//   * the X rows of a step (MG x 16 rows x 1 KiB) are staged once per workgroup through LDS (same image as above) and feed one MFMA per 16-row group.
// Same instruction, same operand bytes per K-tile as the LDS-DMA kernels; the K-tiles are summed in another order (eight interleaved chains instead of one),
// so results agree with them to fp32 rounding of exact e4m3 products -- within the 2 bf16 ulp the restated reference is held to -- not bit for bit.
struct Fp8SkinnyParams
{
    uint16_t* Y;
    const uint8_t* X;
    const uint8_t* W;
    const uint16_t* bias;
    const float* x_scales;
    const float* w_scale;     // as in Fp8TailParams
    int M, K, N;              // M <= 16 MG
    int w_pc = 0;
};

// XALL (MG == 1: at most 16 rows, and their e4m3 image fits the 32 KB of LDS -- the 1-row tail of a prefill chunk, 8 rows at K = 3840): ALL of X sits in LDS before the first product, so the K loop
// has no barrier and no X staging -- eight waves stream their K-tiles of W independently (a decode matvec with an MFMA in it); with the per-step X exchange and its
// barrier fc_down's tail streamed at 2.5 TB/s.  Same products, same per-wave K order, same wave-order reduction: the bits of the staged form.
template <int MG, bool GEGLU, int NR, bool XALL = false>       // NR: 16-row groups of W per workgroup (2 where N gives the chip enough workgroups anyway: half the prologues / reductions per byte)
__global__ __launch_bounds__(512) void gemm_fp8_skinny_kernel(const Fp8SkinnyParams p)
{
    static_assert(!XALL || MG == 1, "the whole-X form is a one-row-group case");
    constexpr int PF = 3;                                   // W fragments requested this many steps ahead (6 / 8 measured SLOWER: fewer resident workgroups per CU and more pipeline moves -- gate_up tail 29.7 -> 38.9 us, T = 2049 prefill 31.1 -> 31.5 ms)
    constexpr int NG = GEGLU ? 2 : 1;                       // gate / up
    constexpr int NA = NG * NR;                             // A fragments per wave and K-tile
    constexpr int kRows = MG * 16;
    constexpr int kStepBytes = 8 * kRows * 128;             // X image of one step: [8 K-tiles][rows][128 B]
    constexpr int NXC = kRows * 64 / 512;                   // 16-byte X chunks per thread and step (2 MG)
    constexpr int kRedBytes = 8 * NA * MG * 1024;
    constexpr int kSmem = 2 * kStepBytes > kRedBytes ? 2 * kStepBytes : kRedBytes;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kSmem];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int K = p.K, nk = (K + 127) / 128, steps = (nk + 7) / 8;
    const int n0 = blockIdx.x * (16 * NR);
    bool row_ok[NR];
    const uint8_t* wrow[NA];
#pragma unroll
    for (int r = 0; r < NR; ++r)
    {
        row_ok[r] = n0 + r * 16 + l15 < p.N;
        const int n = row_ok[r] ? n0 + r * 16 + l15 : 0;
        wrow[r * NG] = p.W + (size_t)n * K;
        if constexpr (GEGLU) wrow[r * NG + 1] = p.W + (size_t)(p.N + n) * K;
    }

    auto load_w = [&](u32x4 (&dst)[NA][2], int s) {
        const int k = (8 * s + wave) * 128 + 32 * g;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                dst[a][h] = (row_ok[a / NG] && k + 16 * h < K) ? ld16_nt(wrow[a] + k + 16 * h) : u32x4{0u, 0u, 0u, 0u};
    };
    u32x4 xr[NXC];
    auto load_x = [&](int s) {
#pragma unroll
        for (int i = 0; i < NXC; ++i)
        {
            const int c = tid + 512 * i, row = c >> 6, cc = c & 63;       // chunk cc of the row's 1 KiB of this step
            const int k = s * 1024 + cc * 16;
            xr[i] = (row < p.M && k < K) ? ld16(p.X + (size_t)row * K + k) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_x = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < NXC; ++i)
        {
            const int c = tid + 512 * i, row = c >> 6, cc = c & 63;
            const int ktl = cc >> 3, ch = cc & 7;
            *reinterpret_cast<u32x4*>(buf + (ktl * kRows + row) * 128 + (((((ch & 1) << 2) | (ch >> 1)) ^ ((row >> 1) & 7)) << 4)) = xr[i];
        }
    };

    f32x4 acc[NA][MG];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int m = 0; m < MG; ++m) acc[a][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // the W pipeline: wq[j] holds step s + j
    u32x4 wq[PF][NA][2];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_w(wq[j], j);          // steps past the end load zeros (masked by k < K)
    if constexpr (XALL)
    {
        // image [K-tile][row < M][128 B], the chunk swizzle of the staged form
        const int M = p.M, nch = M * nk * 8;
        for (int c = tid; c < nch; c += 512)
        {
            const int ch = c & 7, kt = (c >> 3) % nk, row = (c >> 3) / nk;
            const int k = kt * 128 + ch * 16;
            const u32x4 v = (k < K) ? ld16(p.X + (size_t)row * K + k) : u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(smem + (kt * M + row) * 128 + (((((ch & 1) << 2) | (ch >> 1)) ^ ((row >> 1) & 7)) << 4)) = v;
        }
        __syncthreads();
        for (int s = 0; s < steps; ++s)
        {
            const int kt = 8 * s + wave;
            struct Pair { u32x4 lo, hi; };
            i32x8t fa[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) fa[a] = __builtin_bit_cast(i32x8t, (Pair{wq[0][a][0], wq[0][a][1]}));
            u32x4 xlo{0u, 0u, 0u, 0u}, xhi{0u, 0u, 0u, 0u};
            if (l15 < M && kt < nk)
            {
                const int sw = (l15 >> 1) & 7;
                const unsigned char* rowp = smem + (kt * M + l15) * 128;
                xlo = *reinterpret_cast<const u32x4*>(rowp + ((g ^ sw) << 4));
                xhi = *reinterpret_cast<const u32x4*>(rowp + (((4 + g) ^ sw) << 4));
            }
            const i32x8t fb = __builtin_bit_cast(i32x8t, (Pair{xlo, xhi}));
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[a], fb, acc[a][0], 0, 0, 0, 127, 0, 127);
#pragma unroll
            for (int j = 0; j + 1 < PF; ++j)
#pragma unroll
                for (int a = 0; a < NA; ++a) { wq[j][a][0] = wq[j + 1][a][0]; wq[j][a][1] = wq[j + 1][a][1]; }
            load_w(wq[PF - 1], s + PF);
        }
        __syncthreads();                                    // every wave is done with the image before the reduction reuses the buffer
    }
    else
    {
    load_x(0);
    store_x(smem);
    __syncthreads();
    for (int s = 0; s < steps; ++s)
    {
        const bool more = s + 1 < steps;
        if (more) load_x(s + 1);
        const unsigned char* xb = smem + (s & 1) * kStepBytes + wave * (kRows * 128);
        struct Pair { u32x4 lo, hi; };
        i32x8t fa[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) fa[a] = __builtin_bit_cast(i32x8t, (Pair{wq[0][a][0], wq[0][a][1]}));
#pragma unroll
        for (int m = 0; m < MG; ++m)
        {
            const int r = m * 16 + l15, sw = (r >> 1) & 7;
            const unsigned char* rowp = xb + r * 128;
            const i32x8t fb = __builtin_bit_cast(i32x8t, (Pair{*reinterpret_cast<const u32x4*>(rowp + ((g ^ sw) << 4)), *reinterpret_cast<const u32x4*>(rowp + (((4 + g) ^ sw) << 4))}));
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a][m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[a], fb, acc[a][m], 0, 0, 0, 127, 0, 127);
        }
        // shift the W pipeline and request step s + PF
#pragma unroll
        for (int j = 0; j + 1 < PF; ++j)
#pragma unroll
            for (int a = 0; a < NA; ++a) { wq[j][a][0] = wq[j + 1][a][0]; wq[j][a][1] = wq[j + 1][a][1]; }
        load_w(wq[PF - 1], s + PF);
        if (more) store_x(smem + ((s + 1) & 1) * kStepBytes);
        __syncthreads();
    }
    }

    // ---- the eight K-interleaved partial sums meet in LDS, in wave order ----
    float* red = reinterpret_cast<float*>(smem);            // [8 waves][NA][MG][64 lanes] f32x4
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int m = 0; m < MG; ++m) *reinterpret_cast<f32x4*>(red + ((((wave * NA + a) * MG + m) * 64 + lane) << 2)) = acc[a][m];
    __syncthreads();
    // (row group of X, row group of W) pairs are dealt to the waves
    for (int job = wave; job < MG * NR; job += 8)
    {
        const int m = job % MG, r = job / MG;
        f32x4 sum[NG];
#pragma unroll
        for (int q = 0; q < NG; ++q)
        {
            const int a = r * NG + q;
            sum[q] = *reinterpret_cast<const f32x4*>(red + ((((0 * NA + a) * MG + m) * 64 + lane) << 2));
#pragma unroll
            for (int w = 1; w < 8; ++w)
            {
                const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((((w * NA + a) * MG + m) * 64 + lane) << 2));
                sum[q] = f32x4{sum[q][0] + v[0], sum[q][1] + v[1], sum[q][2] + v[2], sum[q][3] + v[3]};
            }
        }
        const int row = m * 16 + l15, n = n0 + r * 16 + 4 * g;
        if (row >= p.M || n >= p.N) continue;
        const bool pc = p.w_pc != 0;
        const float ts = p.x_scales[row];
        float v[4], wg[4], wu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            const int ne = min(n + e, p.N - 1);       // columns past N are never stored
            wg[e] = pc ? p.w_scale[ne] : *p.w_scale;
            wu[e] = (GEGLU && pc) ? p.w_scale[p.N + ne] : wg[e];
        }
        if constexpr (GEGLU)
        {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(fp8_linear_out(pc, sum[0][e], wg[e], ts)) * fp8_linear_out(pc, sum[1][e], wu[e], ts);
        }
        else
        {
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                v[e] = fp8_scale_bias(pc, sum[0][e], wg[e], ts, p.bias != nullptr, (p.bias && n + e < p.N) ? bf16_bits_to_f32(p.bias[n + e]) : 0.0f);
            }
        }
        uint16_t* y = p.Y + (size_t)row * p.N + n;
        if ((p.N & 3) == 0) *reinterpret_cast<u32x2*>(y) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        else
        {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.N) y[e] = f32_to_bf16_bits(v[e]);
        }
    }
}

int g_skinny_whole_x = 1;      // tuning "gemm_fp8.skinny_whole_x": the barrier-free <= 4-row form of the skinny kernel
MILA_TUNE("gemm_fp8.skinny_whole_x", g_skinny_whole_x);

template <int MG, bool GEGLU>
static void launch_skinny_mg(const Fp8SkinnyParams& p, hipStream_t s)
{
    // two W row groups per workgroup where that still leaves >= 2 workgroups per CU
    constexpr bool kTwoFits = !(GEGLU && MG == 4);       // 4 A fragments x 4 row groups of accumulators + the W pipeline do not fit 256 registers
    if constexpr (MG == 1)
    {
        // rows whose whole e4m3 image fits the kernel's 32 KB of LDS: the barrier-free form
        const int nk = (p.K + 127) / 128;
        if (g_skinny_whole_x && p.M * nk * 128 <= 32768)      // (M <= 16 here: MG == 1)
        {
            if (kTwoFits && (p.N + 31) / 32 >= 2 * kNumCU) hipLaunchKernelGGL((gemm_fp8_skinny_kernel<1, GEGLU, 2, true>), dim3((p.N + 31) / 32), dim3(512), 0, s, p);
            else hipLaunchKernelGGL((gemm_fp8_skinny_kernel<1, GEGLU, 1, true>), dim3((p.N + 15) / 16), dim3(512), 0, s, p);
            return;
        }
    }
    if (kTwoFits && (p.N + 31) / 32 >= 2 * kNumCU) hipLaunchKernelGGL((gemm_fp8_skinny_kernel<MG, GEGLU, 2>), dim3((p.N + 31) / 32), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((gemm_fp8_skinny_kernel<MG, GEGLU, 1>), dim3((p.N + 15) / 16), dim3(512), 0, s, p);
}

template <bool GEGLU>
static int launch_skinny(const Fp8SkinnyParams& p, hipStream_t s)
{
    const int groups = (p.M + 15) / 16;
    note_form(GEGLU ? "fp8_skinny_geglu" : "fp8_skinny");
    if (groups <= 1) launch_skinny_mg<1, GEGLU>(p, s);
    else if (groups == 2) launch_skinny_mg<2, GEGLU>(p, s);
    else launch_skinny_mg<4, GEGLU>(p, s);
    MILA_LAUNCH_CHECK("gemm_fp8_skinny");
}

template <int WN, int PT, int QT, bool GEGLU>
static int launch_tail_t(Fp8TailParams p, hipStream_t s)
{
    constexpr int BO = (GEGLU ? WN * PT * 8 : WN * PT * 16);
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = (p.N + BO - 1) / BO;
    note_form(GEGLU ? "fp8_tail_geglu" : "fp8_tail");
    hipLaunchKernelGGL((gemm_fp8_tail_kernel<WN, PT, QT, GEGLU>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
    MILA_LAUNCH_CHECK("gemm_fp8_tail");
}

// the widest W tile that still gives the chip about one workgroup per CU
template <bool GEGLU>
static int launch_tail(const Fp8TailParams& p, hipStream_t s)
{
    const int tiles_m = (p.M + 127) / 128;
    auto wgs = [&](int bo) { return tiles_m * ((p.N + bo - 1) / bo); };
    constexpr int D = GEGLU ? 2 : 1;
    if (wgs(128 / D) >= 192) return launch_tail_t<2, 4, 4, GEGLU>(p, s);
    if (wgs(64 / D) >= 192) return launch_tail_t<1, 4, 2, GEGLU>(p, s);
    return launch_tail_t<1, 2, 2, GEGLU>(p, s);
}

int g_gemm_fp8_tail_form = 0;       // tuning "gemm_fp8.tail_form": 0 = LDS-DMA kernels on the leading rows, the tail kernels by row count on the rest; 1 = EVERY row on the masked 128-row
                                    // LDS tiles (bit-identical to the LDS-DMA kernels: the test of that statement); 2 = every row as skinny pieces
MILA_TUNE("gemm_fp8.tail_form", g_gemm_fp8_tail_form);
constexpr int kSkinnyRows = 64;     // rows one skinny launch takes
constexpr int kSkinnyMaxTail = 64;  // tails up to here run as ONE skinny launch; longer ones on the 128-row LDS tiles (measured: four skinny pieces of a 208-row tail cost more than two LDS tile rows)

static bool use_skinny(int M) { return g_gemm_fp8_tail_form == 2 || (g_gemm_fp8_tail_form == 0 && M <= kSkinnyMaxTail); }

int launch_gemm_fp8_tail(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, const uint16_t* bias,
                         int M, int K, int N, hipStream_t s)
{
    if (use_skinny(M))
    {
        for (int r0 = 0; r0 < M; r0 += kSkinnyRows)
        {
            Fp8SkinnyParams q{Y + (size_t)r0 * N, X8 + (size_t)r0 * K, W8, bias, x_scales + r0, w_scale.p, std::min(kSkinnyRows, M - r0), K, N, w_scale.per_channel};
            const int rc = launch_skinny<false>(q, s);
            if (rc) return rc;
        }
        return MILA_OK;
    }
    Fp8TailParams p{Y, X8, W8, bias, x_scales, w_scale.p, M, K, N, 0, 0, w_scale.per_channel};
    return launch_tail<false>(p, s);
}
int launch_gemm_fp8_geglu_tail(uint16_t* Y, const uint8_t* X8, const uint8_t* W8, const float* x_scales, Fp8WScale w_scale, int M, int K, int F,
                               hipStream_t s)
{
    if (use_skinny(M))
    {
        for (int r0 = 0; r0 < M; r0 += kSkinnyRows)
        {
            Fp8SkinnyParams q{Y + (size_t)r0 * F, X8 + (size_t)r0 * K, W8, nullptr, x_scales + r0, w_scale.p, std::min(kSkinnyRows, M - r0), K, F, w_scale.per_channel};
            const int rc = launch_skinny<true>(q, s);
            if (rc) return rc;
        }
        return MILA_OK;
    }
    Fp8TailParams p{Y, X8, W8, nullptr, x_scales, w_scale.p, M, K, F, 0, 0, w_scale.per_channel};
    return launch_tail<true>(p, s);
}

}  // namespace mila
