// Experiment (GPU box): how fast can a resident workgroup per CU pull a contiguous byte stream, by path?
//   dma : LDS-DMA (global_load_lds 16 B/lane) into a per-wave ring of DEPTH 1-KiB pieces, read back with ds_read_b128
//   reg : global_load_dwordx4 into a register ring of DEPTH pieces
// Every wave owns a contiguous slice of the buffer; the "use" of a piece is an xor into an accumulator.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_paths tools/experiments/stream_paths.hip ; run it with no arguments
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW) void dma_kernel(const uint8_t* buf, size_t per_wave, uint32_t* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t* src = buf + ((size_t)blockIdx.x * NW + wave) * per_wave + (size_t)lane * 16;
    unsigned char* ring = lds + (size_t)wave * DEPTH * 1024;
    const int pieces = (int)(per_wave / 1024);
    int issued = 0, islot = 0, cslot = 0;
    for (; issued < DEPTH && issued < pieces; ++issued)
    {
        __builtin_amdgcn_global_load_lds(src + (size_t)issued * 1024, (__attribute__((address_space(3))) void*)(ring + islot * 1024), 16, 0, 2);
        islot = islot + 1 == DEPTH ? 0 : islot + 1;
    }
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int p = 0; p < pieces; ++p)
    {
        if (issued - p == DEPTH) wait_vm<DEPTH - 1>();
        else wait_vm<0>();
        acc ^= *reinterpret_cast<const u32x4*>(ring + cslot * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cslot = cslot + 1 == DEPTH ? 0 : cslot + 1;
        if (issued < pieces)
        {
            __builtin_amdgcn_global_load_lds(src + (size_t)issued * 1024, (__attribute__((address_space(3))) void*)(ring + islot * 1024), 16, 0, 2);
            islot = islot + 1 == DEPTH ? 0 : islot + 1;
            ++issued;
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1u;
}


// the engine's mapping: wave (b, w) streams rows b + 256 (w + NW t), t = 0, 1, ... of `row_pieces` KiB each; XREAD: also read 16 B/lane of an
// LDS-resident vector per piece (the matvec's x)
template <int NW, int DEPTH, bool XREAD>
__global__ __launch_bounds__(64 * NW) void dma_rows_kernel(const uint8_t* buf, int row_pieces, int rows_per_wave, uint32_t* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* ring = lds + (size_t)wave * DEPTH * 1024;
    const u32x4* xs = reinterpret_cast<const u32x4*>(lds + (size_t)NW * DEPTH * 1024);
    const int pieces = row_pieces * rows_per_wave;
    const size_t row_bytes = (size_t)row_pieces * 1024;
    int issued = 0, islot = 0, cslot = 0, ip = 0, it = 0;
    auto issue = [&]() {
        const uint8_t* src = buf + (size_t)(blockIdx.x + 256 * (wave + NW * it)) * row_bytes + (size_t)ip * 1024 + (size_t)lane * 16;
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(ring + islot * 1024), 16, 0, 2);
        islot = islot + 1 == DEPTH ? 0 : islot + 1;
        ++issued;
        if (++ip == row_pieces) { ip = 0; ++it; }
    };
    while (issued < DEPTH && issued < pieces) issue();
    u32x4 acc = {0u, 0u, 0u, 0u};
    int xp = 0;
    for (int p = 0; p < pieces; ++p)
    {
        if (issued - p == DEPTH) wait_vm<DEPTH - 1>();
        else wait_vm<0>();
        acc ^= *reinterpret_cast<const u32x4*>(ring + cslot * 1024 + lane * 16);
        if constexpr (XREAD) { acc ^= xs[xp * 64 + lane]; xp = xp + 1 == row_pieces ? 0 : xp + 1; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cslot = cslot + 1 == DEPTH ? 0 : cslot + 1;
        if (issued < pieces) issue();
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1u;
}

template <int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW) void reg_kernel(const uint8_t* buf, size_t per_wave, uint32_t* out)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32x4* src = reinterpret_cast<const u32x4*>(buf + ((size_t)blockIdx.x * NW + wave) * per_wave) + lane;
    const int pieces = (int)(per_wave / 1024);       // a multiple of DEPTH
    u32x4 r[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) r[k] = __builtin_nontemporal_load(src + (size_t)k * 64);
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int p = 0; p < pieces; p += DEPTH)
    {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
        {
            wait_vm<DEPTH - 1>();
            acc ^= r[k];
            const int nx = p + DEPTH + k;
            r[k] = __builtin_nontemporal_load(src + (size_t)(nx < pieces ? nx : pieces - 1) * 64);
        }
    }
    wait_vm<0>();
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1u;
}

template <typename F>
static void timeit(const char* name, size_t bytes, F&& launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch(i);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.0f;
    const int reps = 10;
    for (int i = 0; i < reps; ++i)
    {
        hipEventRecord(a, 0);
        launch(i);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("%-28s best %7.1f us (%5.2f TB/s)   mean %7.1f us (%5.2f TB/s)   %s\n", name, best * 1e3, bytes / (best * 1e-3) * 1e-12, sum / reps * 1e3,
           bytes / (sum / reps * 1e-3) * 1e-12, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const size_t per_cu = 2u << 20;                  // 2 MiB per CU and launch: 512 MiB per launch
    const size_t total = per_cu * 256;
    const int NBUF = 4;                              // rotate buffers: 2 GiB > the 256 MiB MALL
    std::vector<uint8_t*> bufs(NBUF);
    for (auto& p : bufs) { hipMalloc(&p, total); hipMemset(p, 1, total); }
    uint32_t* out;
    hipMalloc(&out, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<8, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<16, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<4, 30>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<8, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    timeit("dma  8 waves x 15 KiB", total, [&](int i) { hipLaunchKernelGGL((dma_kernel<8, 15>), dim3(256), dim3(512), 8 * 15 * 1024, 0, bufs[i % NBUF], per_cu / 8, out); });
    timeit("dma  8 waves x  8 KiB", total, [&](int i) { hipLaunchKernelGGL((dma_kernel<8, 8>), dim3(256), dim3(512), 8 * 8 * 1024, 0, bufs[i % NBUF], per_cu / 8, out); });
    timeit("dma 16 waves x  8 KiB", total, [&](int i) { hipLaunchKernelGGL((dma_kernel<16, 8>), dim3(256), dim3(1024), 16 * 8 * 1024, 0, bufs[i % NBUF], per_cu / 16, out); });
    timeit("dma  4 waves x 30 KiB", total, [&](int i) { hipLaunchKernelGGL((dma_kernel<4, 30>), dim3(256), dim3(256), 4 * 30 * 1024, 0, bufs[i % NBUF], per_cu / 4, out); });
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_rows_kernel<8, 15, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_rows_kernel<8, 15, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // 2 MiB per CU = 2048 pieces = 256 per wave: rows of 8 / 30 KiB
    timeit("dma rows  8 KiB", total, [&](int i) { hipLaunchKernelGGL((dma_rows_kernel<8, 15, false>), dim3(256), dim3(512), 8 * 15 * 1024 + 32768, 0, bufs[i % NBUF], 8, 32, out); });
    timeit("dma rows  8 KiB + x", total, [&](int i) { hipLaunchKernelGGL((dma_rows_kernel<8, 15, true>), dim3(256), dim3(512), 8 * 15 * 1024 + 32768, 0, bufs[i % NBUF], 8, 32, out); });
    timeit("dma rows 32 KiB", total, [&](int i) { hipLaunchKernelGGL((dma_rows_kernel<8, 15, false>), dim3(256), dim3(512), 8 * 15 * 1024 + 32768, 0, bufs[i % NBUF], 32, 8, out); });
    timeit("dma rows 32 KiB + x", total, [&](int i) { hipLaunchKernelGGL((dma_rows_kernel<8, 15, true>), dim3(256), dim3(512), 8 * 15 * 1024 + 32768, 0, bufs[i % NBUF], 32, 8, out); });
    timeit("reg  8 waves x 16 KiB", total, [&](int i) { hipLaunchKernelGGL((reg_kernel<8, 16>), dim3(256), dim3(512), 0, 0, bufs[i % NBUF], per_cu / 8, out); });
    timeit("reg  8 waves x  8 KiB", total, [&](int i) { hipLaunchKernelGGL((reg_kernel<8, 8>), dim3(256), dim3(512), 0, 0, bufs[i % NBUF], per_cu / 8, out); });
    timeit("reg 16 waves x  8 KiB", total, [&](int i) { hipLaunchKernelGGL((reg_kernel<16, 8>), dim3(256), dim3(1024), 0, 0, bufs[i % NBUF], per_cu / 16, out); });
    timeit("reg  4 waves x 32 KiB", total, [&](int i) { hipLaunchKernelGGL((reg_kernel<4, 32>), dim3(256), dim3(256), 0, 0, bufs[i % NBUF], per_cu / 4, out); });
    timeit("reg 16 waves x 16 KiB", total, [&](int i) { hipLaunchKernelGGL((reg_kernel<16, 16>), dim3(256), dim3(1024), 0, 0, bufs[i % NBUF], per_cu / 16, out); });
    return 0;
}
