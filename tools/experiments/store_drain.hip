// How long does a workgroup wait for the stores of a finished 256 x 256 bf16 tile (128 KB, 16 16-byte stores per lane) to be acknowledged -- the s_waitcnt the
// persistent GEMM reaches one K-tile into its next tile (vmcnt counts loads and stores in one order)?  By the number of workgroups storing at once and by the pause
// between bursts (a stand-in for the K loop).  s_memtime ticks at 100 MHz.
// hipcc --offload-arch=gfx950 -O3 -o tools/experiments/_build/store_drain tools/experiments/store_drain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void drain_kernel(uint16_t* Y, int N, int tiles_n, int ntiles, int pause_sleeps, int nt, unsigned long long* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3, l15 = lane & 15, g = lane >> 4;
    unsigned long long issue = 0, drain = 0;
    int count = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    {
        for (int i = 0; i < pause_sleeps; ++i) __builtin_amdgcn_s_sleep(32);      // 32 x 64 cycles each
        __syncthreads();
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
        const unsigned long long t0 = __builtin_readcyclecounter();
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
                        u32x4* y = reinterpret_cast<u32x4*>(Y + (size_t)m * N + n);
                        const u32x4 v{(uint32_t)m, (uint32_t)n, 0x3f803f80u, 0x40004000u};
                        if (nt) __builtin_nontemporal_store(v, y);
                        else *y = v;
                    }
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        issue += t1 - t0;
        drain += t2 - t1;
        ++count;
    }
    if (threadIdx.x == 0 && count) { out[blockIdx.x * 2] = issue / count; out[blockIdx.x * 2 + 1] = drain / count; }
}

// the same 128 KB written in other shapes: ROWS rows x (64 / ROWS) lanes x BYTES per lane and instruction (row-wise, as an epilogue transposed through LDS would)
template <int ROWS, int BYTES>
__global__ __launch_bounds__(512) void shape_kernel(uint16_t* Y, int N, int tiles_n, int ntiles, unsigned long long* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = 64 / ROWS, EPL = BYTES / 2;            // lanes per row, elements per lane
    constexpr int SEG = 256 / (LPR * EPL);                     // instructions per group of ROWS rows
    unsigned long long total = 0;
    int count = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    {
        __syncthreads();
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int rr = 0; rr < 32; rr += ROWS)
#pragma unroll
            for (int sg = 0; sg < SEG; ++sg)
            {
                const int m = m0 + wave * 32 + rr + lane / LPR;
                const int n = n0 + (sg * LPR + lane % LPR) * EPL;
                uint16_t* y = Y + (size_t)m * N + n;
                if constexpr (BYTES == 16) *reinterpret_cast<u32x4*>(y) = u32x4{(uint32_t)m, (uint32_t)n, 1u, 2u};
                else if constexpr (BYTES == 8) *reinterpret_cast<uint2*>(y) = uint2{(uint32_t)m, (uint32_t)n};
                else *reinterpret_cast<uint32_t*>(y) = (uint32_t)m;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        total += __builtin_readcyclecounter() - t0;
        ++count;
    }
    if (threadIdx.x == 0 && count) out[blockIdx.x * 2] = total / count;
}
template <int ROWS, int BYTES>
static void run_shape(uint16_t* Y, int N, int tiles_n, unsigned long long* out, const char* what)
{
    for (int wgs : {32, 256})
    {
        std::vector<unsigned long long> h(512);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((shape_kernel<ROWS, BYTES>), dim3(wgs), dim3(512), 0, 0, Y, N, tiles_n, wgs * 24, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %3d workgroups: %6.2f us per 128 KB tile\n", what, wgs, ms * 1e3 / 24);
    }
}

int main()
{
    const int M = 8192, N = 50176, tiles_n = N / 256;
    uint16_t* Y;
    (void)hipMalloc(&Y, (size_t)M * N * 2);
    unsigned long long* out;
    (void)hipMalloc(&out, 256 * 2 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(512);
    printf("cycles are the shader clock's (__builtin_readcyclecounter); a 256 x 256 tile = 16 stores of 16 bytes per lane\n");
    for (int nt = 0; nt < 2; ++nt)
        for (int wgs : {8, 32, 128, 256})
            for (int pause : {0, 4, 16})
            {
                const int ntiles = wgs * 24;
                (void)hipMemset(out, 0, 512 * sizeof(unsigned long long));
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(drain_kernel, dim3(wgs), dim3(512), 0, 0, Y, N, tiles_n, ntiles, pause, nt, out);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                (void)hipMemcpy(h.data(), out, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                double is = 0, dr = 0;
                for (int i = 0; i < wgs; ++i) { is += h[2 * i]; dr += h[2 * i + 1]; }
                printf("%s stores, %3d workgroups, pause %5d cycles between bursts: issue %6.0f cycles, drain %6.0f cycles per tile; %7.1f us per tile round\n",
                       nt ? "non-temporal" : "plain       ", wgs, pause * 32 * 64, is / wgs, dr / wgs, ms * 1e3 / 24);
            }
    run_shape<16, 16>(Y, N, tiles_n, out, "row-wise, 16 rows x  64 B per instruction (16 B per lane)");
    run_shape<4, 16>(Y, N, tiles_n, out, "row-wise,  4 rows x 256 B per instruction (16 B per lane)");
    run_shape<2, 16>(Y, N, tiles_n, out, "row-wise,  2 rows x 512 B per instruction (16 B per lane)");
    run_shape<1, 8>(Y, N, tiles_n, out, "row-wise,  1 row  x 512 B per instruction ( 8 B per lane)");
    run_shape<2, 8>(Y, N, tiles_n, out, "row-wise,  2 rows x 256 B per instruction ( 8 B per lane)");
    run_shape<1, 4>(Y, N, tiles_n, out, "row-wise,  1 row  x 256 B per instruction ( 4 B per lane)");
    return 0;
}
