// bf16 x bf16 GEMM for up to 32 rows with a caller workspace: the few-row form behind mila_cdna4_gemm_bf16_ws (gemm.hip: bf16_ws_plan).
// A 2 ... 32-token prompt (or chunk tail) is a weight stream: 448 MB per Gemma layer against a few MB of activations.  The forms that served it stream slower than the
// decode matvec does -- the 256 x 128 ring keeps 2 x 16 KB of weights in flight per CU and stages 240 padding rows per real one (fc_down at 64 rows: 3 TB/s), the skinny
// kernel (gemm_skinny_bf16.hip) re-reads its X image for every 16 W rows.  Here:
//   * a workgroup owns 128 W rows (8 waves x 16) and ONE slice of K; the slice's X image ([K-tile][row][128 B], <= 64 KB) goes to LDS once and serves all eight waves:
//     X traffic is 1/8 of the weights' at 16 rows;
//   * a wave owns its 16 W rows over the whole slice -- no cross-wave reduction: a lane's MFMA A operands ARE its two 16-byte loads of the K-tile
//     (row l15, bytes 16 g .. and 64 + 16 g ..), kept kPF K-tiles ahead in statically indexed registers: 8 x 2 KB per wave, 128 KB per workgroup in flight.
//     Plain (cached) loads: the two halves of a 128-byte line are fetched by two instructions, and with the non-temporal hint the second one went back to memory --
//     fc_gate_up at 16 rows 56 -> 47 us, a 16-token prefill 7.4 -> 6.7 ms (profiles/r03_splitk.txt);
//   * K is split over gridDim.y slices so that the launch has about two workgroups per CU; slice ks writes its raw fp32 accumulators to partials[ks][M][N] and
//     splitk_reduce_kernel (gemm256.hip) sums the slices in a fixed order and applies the epilogue (bias, GELU, bf16) -- the same second kernel as the tile form.
#include <algorithm>

#include "common.h"

namespace mila {

struct FewRowParams
{
    float* P;                 // [S][M][N] fp32
    const uint16_t* X;
    const uint16_t* W;
    int M, K, N, S;
};

constexpr int kFewRowPF = 8;              // K-tiles of W in flight per wave
constexpr int kFewRowMaxImage = 65536;    // bytes of LDS for the slice's X image

template <int MG>
__global__ __launch_bounds__(512) void gemm_bf16_fewrow_kernel(const FewRowParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PF = kFewRowPF;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int KB = p.K * 2, nk_all = KB / 128;                 // K % 64 == 0: whole 128-byte K-tiles
    const int ks = blockIdx.y;
    const int kt0 = ks * nk_all / p.S, nkl = (ks + 1) * nk_all / p.S - kt0;
    const int M = p.M;
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(p.X);
    const int n_row = blockIdx.x * 128 + wave * 16 + l15;      // this lane's W row
    const bool row_ok = n_row < p.N;
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(p.W) + (size_t)(row_ok ? n_row : 0) * KB + (size_t)kt0 * 128 + 16 * g;

    u32x4 wq[PF][2];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) wq[j][h] = (row_ok && j < nkl) ? ld16(wrow + j * 128 + 64 * h) : u32x4{0u, 0u, 0u, 0u};

    // the slice's X image: [K-tile][row < M][128 B], 16-byte chunks swizzled by row as in the skinny kernels
    {
        const int nch = M * nkl * 8;
        for (int c = tid; c < nch; c += 512)
        {
            const int ch = c & 7, kt = (c >> 3) % nkl, row = (c >> 3) / nkl;
            const u32x4 v = ld16(Xb + (size_t)row * KB + (size_t)(kt0 + kt) * 128 + ch * 16);
            *reinterpret_cast<u32x4*>(smem + (kt * M + row) * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v;
        }
    }
    __syncthreads();

    f32x4 acc[MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    for (int base = 0; base < nkl; base += PF)
    {
#pragma unroll
        for (int j = 0; j < PF; ++j)
        {
            const int kt = base + j;
            if (kt < nkl)
            {
                const unsigned char* img = smem + (size_t)kt * M * 128;
#pragma unroll
                for (int m = 0; m < MG; ++m)
                {
                    const int r = m * 16 + l15, sw = (r >> 1) & 7;
                    u32x4 x0{0u, 0u, 0u, 0u}, x1{0u, 0u, 0u, 0u};
                    if (r < M)
                    {
                        x0 = *reinterpret_cast<const u32x4*>(img + r * 128 + ((g ^ sw) << 4));
                        x1 = *reinterpret_cast<const u32x4*>(img + r * 128 + (((4 + g) ^ sw) << 4));
                    }
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[j][0]), __builtin_bit_cast(bf16x8, x0), acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[j][1]), __builtin_bit_cast(bf16x8, x1), acc[m], 0, 0, 0);
                }
                const int nxt = kt + PF;
                if (nxt < nkl)
                {
#pragma unroll
                    for (int h = 0; h < 2; ++h) wq[j][h] = row_ok ? ld16(wrow + (size_t)nxt * 128 + 64 * h) : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
    }

    // D[W row 4 g + e][X row l15]: 16 bytes of partials[ks][row][n .. n + 3]
    const int n = blockIdx.x * 128 + wave * 16 + 4 * g;
    if (n >= p.N) return;      // N % 16 == 0: a 4-column group is inside or outside
    float* P = p.P + (size_t)ks * M * p.N;
#pragma unroll
    for (int m = 0; m < MG; ++m)
    {
        const int row = m * 16 + l15;
        if (row < M) *reinterpret_cast<f32x4*>(P + (size_t)row * p.N + n) = acc[m];
    }
}

// S (>= 1) for the few-row form, or 0: about two workgroups per CU, a slice whose X image fits kFewRowMaxImage and has at least 4 K-tiles
int gemm_fewrow_splits(int M, int K, int N)
{
    if (M <= 0 || M > 32 || K % 64 != 0 || N % 16 != 0) return 0;
    const int nk = K / 64, tiles = (N + 127) / 128;
    const int nkl_max = kFewRowMaxImage / (M * 128);                      // K-tiles of the image
    const int s_min = (nk + nkl_max - 1) / nkl_max, s_max = std::max(1, nk / 4);
    // (at most 2 x 256 workgroups -- one round at two per CU: 540 of them ran as two rounds, fc_down 38 us)
    int S = std::max(s_min, std::max(1, 2 * kNumCU / tiles));
    S = std::min(S, s_max);
    if (S < s_min || S > 64) return 0;
    return S;
}

int launch_gemm_bf16_fewrow(float* partials, const uint16_t* X, const uint16_t* W, int M, int K, int N, int S, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set)
    {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_fewrow_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kFewRowMaxImage), "hipFuncSetAttribute(fewrow<1>)");
        if (rc) return rc;
        rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_fewrow_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kFewRowMaxImage), "hipFuncSetAttribute(fewrow<2>)");
        if (rc) return rc;
        attr_set = true;
    }
    const FewRowParams p{partials, X, W, M, K, N, S};
    const int nk = K / 64, nkl_max = (nk + S - 1) / S + 1;
    const size_t lds = std::min<size_t>(kFewRowMaxImage, (size_t)nkl_max * M * 128);
    const dim3 grid((N + 127) / 128, S);
    if (M <= 16) hipLaunchKernelGGL(gemm_bf16_fewrow_kernel<1>, grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL(gemm_bf16_fewrow_kernel<2>, grid, dim3(512), lds, s, p);
    return check_hip(hipGetLastError(), "gemm_bf16_fewrow");
}

}  // namespace mila
