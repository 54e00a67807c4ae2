// bf16 x bf16 GEMM for FEW rows (<= 64 per launch): the remainder of a ragged prompt (T = 2049: one row), the last chunk of a chunked prefill, a 16-token prompt.
// For so few rows a Linear is a weight STREAM, not a tile problem: the register-staged 128-tile kernel took 23 ms to push the bf16 model's 24 GB of weights past a
// 16-row prompt (1 TB/s), a 1-row remainder cost the bf16 policy +18 ms per prefill.  This is the bf16 sibling of gemm_fp8_skinny_kernel (csrc/gemm_fp8_tail.hip):
//   * a workgroup owns 16 W rows (GEGLU: 16 gate + the matching 16 up rows) and ALL of K; its 8 waves take the 128-byte K-tiles 8 s + w of step s, and the eight
//     partial sums meet in LDS once, at the end, in wave order (a fixed order: the result does not depend on M or on the grid);
//   * a lane's MFMA A operands ARE its two 16-byte global loads (row l15, elements 8 g .. 8 g + 7 and 32 + 8 g .. of the 64-element K-tile), non-temporal, three
//     steps ahead;
//   * X: MG x 16 rows per step through a double-buffered LDS image, or -- XALL: one row group whose whole image fits 32 KB -- all of X in LDS before the first
//     product and no barrier in the K loop.
// y = bf16(bf16(acc) + bias) [Linear]; act: bf16(gelu(that)) [Linear + Gelu]; GEGLU: bf16(gelu(bf16(gate)) * bf16(up)) -- the roundings of the tile kernels.  The
// K-tiles are summed in eight interleaved chains instead of one: results agree with the tile kernels to fp32 rounding (within the 1-2 bf16 ulp both are held to against
// the float64 oracle), not bit for bit.
#include <algorithm>

#include "common.h"

namespace mila {

struct Bf16SkinnyParams
{
    uint16_t* Y;
    const uint16_t* X;
    const uint16_t* W;
    const uint16_t* bias;
    int M, K, N;              // M <= 16 MG; GEGLU: N = F output columns, W has 2 F rows [gate | up]
    int act;                  // 1: tanh-GELU on the stored Linear output
};

template <int MG, bool GEGLU, int NR, bool XALL>
__global__ __launch_bounds__(512) void gemm_bf16_skinny_kernel(const Bf16SkinnyParams p)
{
    static_assert(!XALL || MG == 1, "the whole-X form is a one-row-group case");
    constexpr int PF = 3;                                   // W fragments requested this many steps ahead
    constexpr int NG = GEGLU ? 2 : 1;                       // gate / up
    constexpr int NA = NG * NR;                             // A fragment pairs per wave and K-tile
    constexpr int kRows = MG * 16;
    constexpr int kStepBytes = 8 * kRows * 128;             // X image of one step: [8 K-tiles][rows][128 B]
    constexpr int NXC = kRows * 64 / 512;                   // 16-byte X chunks per thread and step
    constexpr int kRedBytes = 8 * NA * MG * 1024;
    constexpr int kSmem = 2 * kStepBytes > kRedBytes ? 2 * kStepBytes : kRedBytes;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kSmem];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int KB = p.K * 2, nk = (KB + 127) / 128, steps = (nk + 7) / 8;      // bytes of a row, 128-byte K-tiles
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(p.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
    const int n0 = blockIdx.x * (16 * NR);
    bool row_ok[NR];
    const unsigned char* wrow[NA];
#pragma unroll
    for (int r = 0; r < NR; ++r)
    {
        row_ok[r] = n0 + r * 16 + l15 < p.N;
        const int n = row_ok[r] ? n0 + r * 16 + l15 : 0;
        wrow[r * NG] = Wb + (size_t)n * KB;
        if constexpr (GEGLU) wrow[r * NG + 1] = Wb + (size_t)(p.N + n) * KB;
    }
    // h = 0: elements 8 g .. 8 g + 7 (the first v_mfma_f32_16x16x32_bf16 of the K-tile), h = 1: 32 + 8 g ..
    auto load_w = [&](u32x4 (&dst)[NA][2], int s) {
        const int kb = (8 * s + wave) * 128 + 16 * g;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                dst[a][h] = (row_ok[a / NG] && kb + 64 * h < KB) ? ld16_nt(wrow[a] + kb + 64 * h) : u32x4{0u, 0u, 0u, 0u};
    };
    u32x4 xr[NXC];
    auto load_x = [&](int s) {
#pragma unroll
        for (int i = 0; i < NXC; ++i)
        {
            const int c = tid + 512 * i, row = c >> 6, cc = c & 63;       // chunk cc of the row's 1 KiB of this step
            const int kb = s * 1024 + cc * 16;
            xr[i] = (row < p.M && kb < KB) ? ld16(Xb + (size_t)row * KB + kb) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_x = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < NXC; ++i)
        {
            const int c = tid + 512 * i, row = c >> 6, cc = c & 63;
            const int ktl = cc >> 3, ch = cc & 7;
            *reinterpret_cast<u32x4*>(buf + (ktl * kRows + row) * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = xr[i];
        }
    };

    f32x4 acc[NA][MG];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int m = 0; m < MG; ++m) acc[a][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    u32x4 wq[PF][NA][2];      // the W pipeline: wq[j] holds step s + j
#pragma unroll
    for (int j = 0; j < PF; ++j) load_w(wq[j], j);          // steps past the end load zeros
    auto products = [&](const unsigned char* rows_base, int pitch_rows, bool have_tile, int m_limit) {
        // rows_base: the K-tile's [rows][128 B] image; a lane reads row m * 16 + l15 (zero past m_limit)
#pragma unroll
        for (int m = 0; m < MG; ++m)
        {
            const int r = m * 16 + l15, sw = (r >> 1) & 7;
            u32x4 x0{0u, 0u, 0u, 0u}, x1{0u, 0u, 0u, 0u};
            if (have_tile && r < m_limit)
            {
                const unsigned char* rowp = rows_base + r * 128;
                x0 = *reinterpret_cast<const u32x4*>(rowp + ((g ^ sw) << 4));
                x1 = *reinterpret_cast<const u32x4*>(rowp + (((4 + g) ^ sw) << 4));
            }
#pragma unroll
            for (int a = 0; a < NA; ++a)
            {
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[0][a][0]), __builtin_bit_cast(bf16x8, x0), acc[a][m], 0, 0, 0);
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[0][a][1]), __builtin_bit_cast(bf16x8, x1), acc[a][m], 0, 0, 0);
            }
        }
        (void)pitch_rows;
    };
    auto shift_w = [&](int s) {
#pragma unroll
        for (int j = 0; j + 1 < PF; ++j)
#pragma unroll
            for (int a = 0; a < NA; ++a) { wq[j][a][0] = wq[j + 1][a][0]; wq[j][a][1] = wq[j + 1][a][1]; }
        load_w(wq[PF - 1], s + PF);
    };
    if constexpr (XALL)
    {
        // image [K-tile][row < M][128 B]
        const int M = p.M, nch = M * nk * 8;
        for (int c = tid; c < nch; c += 512)
        {
            const int ch = c & 7, kt = (c >> 3) % nk, row = (c >> 3) / nk;
            const int kb = kt * 128 + ch * 16;
            const u32x4 v = (kb < KB) ? ld16(Xb + (size_t)row * KB + kb) : u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(smem + (kt * M + row) * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v;
        }
        __syncthreads();
        for (int s = 0; s < steps; ++s)
        {
            const int kt = 8 * s + wave;
            products(smem + (size_t)kt * M * 128, M, kt < nk, M);
            shift_w(s);
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
            products(smem + (s & 1) * kStepBytes + wave * (kRows * 128), kRows, true, kRows);
            shift_w(s);
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
    for (int job = wave; job < MG * NR; job += 8)          // (row group of X, row group of W) pairs are dealt to the waves
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
        // D[W row 4 g + e][X row l15]
        const int row = m * 16 + l15, n = n0 + r * 16 + 4 * g;
        if (row >= p.M || n >= p.N) continue;
        float v[4];
        if constexpr (GEGLU)
        {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(round_bf16(sum[0][e])) * round_bf16(sum[1][e]);
        }
        else
        {
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                v[e] = sum[0][e];
                if (p.bias && n + e < p.N) v[e] = round_bf16(v[e]) + bf16_bits_to_f32(p.bias[n + e]);
                if (p.act) v[e] = gelu_tanh(round_bf16(v[e]));
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

template <int MG, bool GEGLU>
static void launch_bf16_skinny_mg(const Bf16SkinnyParams& p, hipStream_t s)
{
    constexpr bool kTwoFits = !(GEGLU && MG == 4);       // 4 A fragment pairs x 4 row groups of accumulators + the W pipeline do not fit
    const bool two = kTwoFits && (p.N + 31) / 32 >= 2 * kNumCU;
    if constexpr (MG == 1)
    {
        const int nk = (p.K * 2 + 127) / 128;
        if (p.M * nk * 128 <= 32768)      // the whole image fits the kernel's LDS: the barrier-free form
        {
            if (two) { if constexpr (kTwoFits) hipLaunchKernelGGL((gemm_bf16_skinny_kernel<1, GEGLU, 2, true>), dim3((p.N + 31) / 32), dim3(512), 0, s, p); }
            else hipLaunchKernelGGL((gemm_bf16_skinny_kernel<1, GEGLU, 1, true>), dim3((p.N + 15) / 16), dim3(512), 0, s, p);
            return;
        }
    }
    if (two) { if constexpr (kTwoFits) hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MG, GEGLU, 2, false>), dim3((p.N + 31) / 32), dim3(512), 0, s, p); }
    else hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MG, GEGLU, 1, false>), dim3((p.N + 15) / 16), dim3(512), 0, s, p);
}

template <bool GEGLU>
static int launch_bf16_skinny_rows(Bf16SkinnyParams p, int M, hipStream_t s)
{
    // 64 rows per launch (every launch streams W once)
    const uint16_t* X = p.X;
    uint16_t* Y = p.Y;
    for (int r0 = 0; r0 < M; r0 += 64)
    {
        p.X = X + (size_t)r0 * p.K;
        p.Y = Y + (size_t)r0 * p.N;
        p.M = std::min(64, M - r0);
        const int groups = (p.M + 15) / 16;
        if (groups <= 1) launch_bf16_skinny_mg<1, GEGLU>(p, s);
        else if (groups == 2) launch_bf16_skinny_mg<2, GEGLU>(p, s);
        else launch_bf16_skinny_mg<4, GEGLU>(p, s);
        const int rc = check_hip(hipGetLastError(), "gemm_bf16_skinny");
        if (rc) return rc;
    }
    return MILA_OK;
}

// Y[M, N] = (act ? gelu : id)(X W^T + bias), any M (64 rows per launch), K % 8 == 0
int launch_gemm_bf16_skinny(uint16_t* Y, const uint16_t* X, const uint16_t* W, const uint16_t* bias, int M, int K, int N, int act, hipStream_t s)
{
    note_form("skinny_bf16");
    return launch_bf16_skinny_rows<false>(Bf16SkinnyParams{Y, X, W, bias, 0, K, N, act}, M, s);
}
// Y[M, F] = GeGLU(X W^T), W = [gate rows | up rows]
int launch_gemm_bf16_skinny_geglu(uint16_t* Y, const uint16_t* X, const uint16_t* W, int M, int K, int F, hipStream_t s)
{
    note_form("skinny_bf16_geglu");
    return launch_bf16_skinny_rows<true>(Bf16SkinnyParams{Y, X, W, nullptr, 0, K, F, 0}, M, s);
}

}  // namespace mila
