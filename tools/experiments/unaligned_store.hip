// What does a bf16 output with an ODD row pitch (GPT-2's logits: N = 50257, rows start on 2-byte boundaries) cost the store path?
// A tile-shaped store pattern (a wave writes 16 rows x 64 bytes per instruction, like the GEMM epilogue's paired stores) into rows of pitch P elements:
//   v16   one 16-byte store per lane (2-byte aligned when P is odd: global memory takes it, unaligned-access mode)
//   v4    four 4-byte stores per lane (4-byte aligned on even rows only)
//   v2    eight 2-byte stores per lane
//   fix   per row: the lanes shift their data by the row's misalignment (DPP neighbours) so that every 16-byte store is ALIGNED; head / tail elements by 2-byte stores
// hipcc --offload-arch=gfx950 -O3 -o tools/experiments/_build/unaligned_store tools/experiments/unaligned_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) u32x4_a2 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(2))) u32_a2 { uint32_t x; };

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(uint16_t* Y, int M, int N, int P)
{
    // tile 256 rows x 256 columns per workgroup, 8 waves (2 x 4), lane (l15, g): row l15 of a 16-row group, 8 columns at g * 8 within a 32-column group
    const int tiles_n = N / 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3, l15 = lane & 15, g = lane >> 4;
    for (int tile = blockIdx.x; tile < (M / 256) * tiles_n; tile += gridDim.x)
    {
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
#pragma unroll
        for (int hA = 0; hA < 2; ++hA)
#pragma unroll
            for (int hB = 0; hB < 2; ++hB)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                    {
                        const int m = m0 + hB * 128 + wc * 32 + qt * 16 + l15;
                        const int n = n0 + hA * 128 + wr * 64 + pp * 32 + g * 8;
                        uint16_t* y = Y + (size_t)m * P + n;
                        const u32x4 v{(uint32_t)m, (uint32_t)n, 0x3f803f80u, 0x40004000u};
                        if constexpr (MODE == 0) *reinterpret_cast<u32x4_a2*>(y) = u32x4_a2{v[0], v[1], v[2], v[3]};
                        else if constexpr (MODE == 1)
                        {
#pragma unroll
                            for (int e = 0; e < 4; ++e) *reinterpret_cast<u32_a2*>(y + 2 * e) = u32_a2{v[e]};
                        }
                        else if constexpr (MODE == 2)
                        {
#pragma unroll
                            for (int e = 0; e < 8; ++e) y[e] = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
                        }
                        else
                        {
                            // aligned form: the 4 lanes of a row hold 32 consecutive elements; shift them right by s = the row's misalignment in elements (0 .. 7) so
                            // that lane g stores the aligned 16 bytes at floor-aligned address; the head (8 - s elements before the first boundary... ) is written
                            // by 2-byte stores of lane g == 0 and the tail by lane g == 3.  Timing stand-in: the same number of instructions and bytes
                            const uintptr_t a = reinterpret_cast<uintptr_t>(y);
                            const int s = (int)((a >> 1) & 7);
                            uint16_t* ya = reinterpret_cast<uint16_t*>(a & ~(uintptr_t)15);
                            // funnel: this lane's aligned chunk = last s elements of the left neighbour's data ++ first 8 - s of its own (values do not matter for the timing)
                            u32x4 left;
#pragma unroll
                            for (int e = 0; e < 4; ++e) left[e] = __shfl_up(v[e], 16, 64);
                            u32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (s & 1) ? __builtin_amdgcn_alignbyte(v[e], left[e], 2) : ((s >> 1) > e ? left[e] : v[e]);
                            if (g > 0 || s == 0) *reinterpret_cast<u32x4*>(ya) = o;
                            else
                                for (int e = s; e < 8; ++e) ya[e] = (uint16_t)o[e >> 1];      // head of the row segment
                            if (g == 3 && s)
                                for (int e = 0; e < s; ++e) ya[8 + e] = (uint16_t)v[e >> 1];   // tail
                        }
                    }
    }
}

// the same tile, written ROW-WISE (what an epilogue transposed through LDS can do): an instruction covers RPI rows x (64 / RPI) lanes x 16 bytes of contiguous columns
template <int RPI>
__global__ __launch_bounds__(512) void store_rows_kernel(uint16_t* Y, int M, int N, int P)
{
    const int tiles_n = N / 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = 64 / RPI;                 // lanes per row
    constexpr int SEG = 32 / LPR;                 // 16-byte pieces of a 256-column row per lane
    for (int tile = blockIdx.x; tile < (M / 256) * tiles_n; tile += gridDim.x)
    {
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
        for (int rr = 0; rr < 32; rr += RPI)      // 32 rows per wave
#pragma unroll
            for (int sg = 0; sg < SEG; ++sg)
            {
                const int m = m0 + wave * 32 + rr + lane / LPR;
                const int n = n0 + (sg * LPR + lane % LPR) * 8;
                uint16_t* y = Y + (size_t)m * P + n;
                *reinterpret_cast<u32x4_a2*>(y) = u32x4_a2{(uint32_t)m, (uint32_t)n, 0x3f803f80u, 0x40004000u};
            }
    }
}
template <int RPI>
static float run_rows(uint16_t* Y, int M, int N, int P)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_rows_kernel<RPI>, dim3(256), dim3(512), 0, 0, Y, M, N, P);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_rows_kernel<RPI>, dim3(256), dim3(512), 0, 0, Y, M, N, P);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10 * 1e3f;
}

template <int MODE>
static float run(uint16_t* Y, int M, int N, int P)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(256), dim3(512), 0, 0, Y, M, N, P);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(256), dim3(512), 0, 0, Y, M, N, P);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10 * 1e3f;
}

int main()
{
    const int M = 8192, N = 50176;      // 196 whole column tiles of GPT-2's lm_head
    uint16_t* Y;
    hipMalloc(&Y, (size_t)M * 50272 * 2 + 64);
    const char* names[4] = {"16-byte stores", "4-byte stores", "2-byte stores", "shifted, aligned 16-byte stores"};
    for (int P : {50176, 50264, 50257})
    {
        float us[4] = {run<0>(Y, M, N, P), run<1>(Y, M, N, P), run<2>(Y, M, N, P), run<3>(Y, M, N, P)};
        for (int k = 0; k < 4; ++k)
            printf("pitch %d (%s): %-32s %8.1f us  %6.0f GB/s\n", P, P % 8 == 0 ? "16-byte aligned rows" : "odd", names[k], us[k], (double)M * N * 2 / us[k] / 1e3);
        const float r[4] = {run_rows<2>(Y, M, N, P), run_rows<4>(Y, M, N, P), run_rows<8>(Y, M, N, P), run_rows<16>(Y, M, N, P)};
        const int rpi[4] = {2, 4, 8, 16};
        for (int k = 0; k < 4; ++k)
            printf("pitch %d: row-wise, %2d rows x %3d bytes per instruction   %8.1f us  %6.0f GB/s\n", P, rpi[k], 1024 / rpi[k], r[k], (double)M * N * 2 / r[k] / 1e3);
    }
    return 0;
}
