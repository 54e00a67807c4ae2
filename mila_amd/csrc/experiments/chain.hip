// Decode chain: the four Linears that sit between two attention calls of a Gemma decode step
// (Components/Transformers/Gemma/Gemma.Block.ixx:287-356: o_proj -> post-attention tail -> fc_gate_up + GeGLU ->
// fc_down -> post-FFN tail -> the next layer's input norm + qkv_proj, or the final norm + tied lm_head) in ONE
// launch of one 16-wave workgroup per CU.
//
// Why: at M == 1 every Linear is a 5-40 us HBM stream and a kernel boundary costs ~3.5-4 us on this part (drain,
// dispatch, first-byte latency with an empty memory pipeline).  Inside one launch the boundary becomes a grid-wide
// hand-off that the weight stream runs THROUGH: every wave requests its first two pipeline steps of the next
// phase's weights (they depend on nothing) before it waits for the previous phase's outputs.
//
// Hand-off between phases (MI355X_MICROARCH.md "visibility", valid-forms table row 1; each XCD has a private L2):
//   * a phase's output vector is written with 4-byte write-through (sc1) stores, one element per 32-bit word;
//   * every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane adds 1 to a
//     device-scope 64-bit arrival counter;
//   * the consumer's lane 0 polls that counter with sc1 loads until all workgroups of the phase have arrived, the
//     workgroup meets at a barrier, and EVERY load of the vector is an 8-byte sc1 load (bypasses the CU's L1; the
//     line was never in this XCD's L2 during the launch);
//   * everything else a phase reads (weights, scales, norm weights, the residual of the previous LAUNCH) is not
//     written inside the launch.  The residual stream between the two tails of a layer stays in registers.
// The arrival counter only ever grows (64 bits); the launch's base epoch is a word that workgroup 0 advances when
// it leaves, so replaying a captured graph needs no per-launch reset.  Spins are bounded by the wall clock: a
// launch whose workgroups are not all resident gives up, sets the error word (checked by the host) and still ends.
//
// The arithmetic of each phase is matvec_body (matvec_body.h), the same code the one-launch-per-Linear kernels
// run: results are bit-identical to the unfused sequence whatever (R, U) a phase uses.
#include <algorithm>

#include "../common.h"
#include "../internal.h"
#include "../rms_common.h"
#include "../matvec_body.h"

namespace mila {

struct ChainParams
{
    MatvecParams ph[4];                 // o_proj, fc_gate_up (+GeGLU), fc_down, next (qkv_proj or lm_head)
    unsigned long long* counter;        // arrivals, monotonically increasing
    unsigned long long* epoch;          // number of hand-offs completed by earlier launches
    uint32_t* error;                    // set to a phase code when a bounded spin gives up
    int nblocks;
};

constexpr long long kChainSpinTicks = 4000000;   // wall_clock64 runs at 100 MHz: 40 ms

__device__ __forceinline__ void chain_arrive(unsigned long long* counter)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(as_global(counter), 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // release: the arrival orders after this workgroup's stores
}

struct ChainWait
{
    unsigned long long* counter;
    unsigned long long target;
    uint32_t* error;
    uint32_t code;
    __device__ __forceinline__ void operator()() const
    {
        if (threadIdx.x == 0)
        {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(as_global(counter), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
            {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > kChainSpinTicks)
                {
                    __hip_atomic_store(as_global(error), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
    }
};

// per-format launch shapes of the four phases (rows per wave step R, chunk positions per step U)
template <int FMT> struct ChainShape;
template <> struct ChainShape<FMT_BF16> { static constexpr int Rs = 1, Us = 4, Rg = 2, Ug = 2; };
template <> struct ChainShape<FMT_FP8> { static constexpr int Rs = 1, Us = 2, Rg = 1, Ug = 2; };
template <> struct ChainShape<FMT_FP4> { static constexpr int Rs = 1, Us = 1, Rg = 1, Ug = 1; };

// FMT: format of the four layer Linears; HEAD: the last phase is the lm_head (format HFMT, fp32 logits)
template <int FMT, bool HEAD, int HFMT>
__global__ __launch_bounds__(1024) void decode_chain_kernel(const ChainParams c)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
    __shared__ float red_a[32], red_b[32];
    using Sh = ChainShape<FMT>;
    const int block = blockIdx.x, nb = c.nblocks;
    const unsigned long long base = *c.epoch * (unsigned long long)nb;

    u32x4 r1[1], rk1[1], rk2[2];
    // phase 0: a = o_proj(attn)
    matvec_body<FMT, Sh::Rs, Sh::Us, 0, false, Y_HANDOFF, 1, X_PLAIN, RES_MEM>(c.ph[0], xs, red_a, red_b, block, nb, rk1, NoWait{});
    chain_arrive(c.counter);
    // phase 1: r1 = res + rmsnorm(a); h = GeGLU(fc_gate_up(rmsnorm(r1)))
    matvec_body<FMT, Sh::Rg, Sh::Ug, 2, true, Y_HANDOFF, 1, X_HANDOFF, RES_MEM>(
        c.ph[1], xs, red_a, red_b, block, nb, r1, ChainWait{c.counter, base + nb, c.error, 1u});
    chain_arrive(c.counter);
    // phase 2: d = fc_down(h)
    matvec_body<FMT, Sh::Rs, Sh::Us, 0, false, Y_HANDOFF, 2, X_HANDOFF, RES_MEM>(
        c.ph[2], xs, red_a, red_b, block, nb, rk2, ChainWait{c.counter, base + 2ull * nb, c.error, 2u});
    chain_arrive(c.counter);
    // phase 3: r2 = (r1 + rmsnorm(d)) * layer_scalar; y = next(rmsnorm(r2))
    if constexpr (HEAD)
        matvec_body<HFMT, 4, 2, 2, false, Y_F32, 1, X_HANDOFF, RES_REG>(
            c.ph[3], xs, red_a, red_b, block, nb, r1, ChainWait{c.counter, base + 3ull * nb, c.error, 3u});
    else
        matvec_body<FMT, Sh::Rs, Sh::Us, 2, false, Y_BF16, 1, X_HANDOFF, RES_REG>(
            c.ph[3], xs, red_a, red_b, block, nb, r1, ChainWait{c.counter, base + 3ull * nb, c.error, 3u});
    // every workgroup has read `epoch` long before workgroup 0 gets here (it passed hand-off 3 after all arrived)
    if (block == 0 && threadIdx.x == 0) *c.epoch = *c.epoch + 3ull;
}

template <int FMT, int U>
static size_t x_lds_bytes(int K)
{
    constexpr int EPC = Fmt<FMT>::kElemsPerChunk;
    const int nchunks = K / EPC;
    const int S = (nchunks + 64 * U - 1) / (64 * U);
    return (size_t)S * 64 * U * EPC * 2;
}

template <int FMT, bool HEAD, int HFMT>
static int launch_chain(const ChainParams& c, hipStream_t s)
{
    using Sh = ChainShape<FMT>;
    size_t lds = x_lds_bytes<FMT, Sh::Us>(c.ph[0].K);
    lds = std::max(lds, x_lds_bytes<FMT, Sh::Ug>(c.ph[1].K));
    lds = std::max(lds, x_lds_bytes<FMT, Sh::Us>(c.ph[2].K));
    lds = std::max(lds, HEAD ? x_lds_bytes<HFMT, 2>(c.ph[3].K) : x_lds_bytes<FMT, Sh::Us>(c.ph[3].K));
    MILA_REQUIRE(lds <= 65536, "decode_chain: %zu bytes of LDS for x (limit 65536)", lds);
    hipLaunchKernelGGL((decode_chain_kernel<FMT, HEAD, HFMT>), dim3(c.nblocks), dim3(1024), lds, s, c);
    MILA_LAUNCH_CHECK("decode_chain");
}

static int chain_num_blocks()
{
    static int n = 0;
    if (n == 0)
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return kNumCU;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : kNumCU;
    }
    return n;
}

constexpr size_t kChainHeaderBytes = 64;   // counter, epoch, error

}  // namespace mila

using namespace mila;

extern "C" {

size_t mila_cdna4_decode_chain_scratch_bytes(int D, int F)
{
    if (D <= 0 || F <= 0) return 0;
    return kChainHeaderBytes + (size_t)(2 * D + F) * 4 + 64;
}

/* every instantiation must fit one 1024-thread workgroup per CU with the largest x buffer: the grid-wide hand-offs need all of them resident */
static int chain_check_residency()
{
    static int checked = 0;
    if (checked) return checked > 0 ? MILA_OK : set_error(MILA_E_UNSUPPORTED, "decode_chain: a workgroup of the chain kernel does not fit a compute unit on this device");
    auto fits = [](const void* fn) {
        int n = 0;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 1024, 65536) == hipSuccess && n >= 1;
    };
    const bool ok = fits(reinterpret_cast<const void*>(&decode_chain_kernel<FMT_BF16, false, FMT_BF16>)) && fits(reinterpret_cast<const void*>(&decode_chain_kernel<FMT_FP8, false, FMT_FP8>)) &&
                    fits(reinterpret_cast<const void*>(&decode_chain_kernel<FMT_FP4, false, FMT_FP4>)) && fits(reinterpret_cast<const void*>(&decode_chain_kernel<FMT_BF16, true, FMT_BF16>)) &&
                    fits(reinterpret_cast<const void*>(&decode_chain_kernel<FMT_FP8, true, FMT_FP8>)) && fits(reinterpret_cast<const void*>(&decode_chain_kernel<FMT_FP4, true, FMT_FP8>));
    checked = ok ? 1 : -1;
    return ok ? MILA_OK : set_error(MILA_E_UNSUPPORTED, "decode_chain: a workgroup of the chain kernel does not fit a compute unit on this device");
}

int mila_cdna4_decode_chain_init(void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    MILA_REQUIRE(scratch != nullptr && scratch_bytes >= kChainHeaderBytes, "decode_chain_init: scratch too small");
    (void)chain_num_blocks();   // device query outside any later stream capture
    const int rc = chain_check_residency();
    if (rc) return rc;
    return check_hip(hipMemsetAsync(scratch, 0, kChainHeaderBytes, as_stream(stream)), "decode_chain_init");
}

int mila_cdna4_decode_chain_status(const void* scratch, int32_t* error_out, mila_stream_t stream)
{
    MILA_REQUIRE(scratch != nullptr && error_out != nullptr, "decode_chain_status: null pointer");
    hipStream_t s = as_stream(stream);
    uint32_t e = 0;
    int rc = check_hip(hipMemcpyAsync(&e, static_cast<const unsigned char*>(scratch) + 16, 4, hipMemcpyDeviceToHost, s), "decode_chain_status");
    if (rc) return rc;
    rc = check_hip(hipStreamSynchronize(s), "decode_chain_status");
    if (rc) return rc;
    *error_out = (int32_t)e;
    return MILA_OK;
}

int mila_cdna4_decode_chain(const mila_decode_chain_args* a, mila_stream_t stream)
{
    MILA_REQUIRE(a != nullptr, "decode_chain: null args");
    const int D = a->D, F = a->F;
    MILA_REQUIRE(D > 0 && F > 0 && a->K_attn > 0 && a->N_next > 0, "decode_chain: dimensions must be positive");
    MILA_REQUIRE(D % 32 == 0 && F % 32 == 0 && a->K_attn % 32 == 0, "decode_chain: D, F, K_attn must be multiples of 32");
    MILA_REQUIRE(D <= 8192 && a->K_attn <= 8192 && F <= 16384, "decode_chain: D, K_attn <= 8192 and F <= 16384 (D=%d K_attn=%d F=%d)", D, a->K_attn, F);
    MILA_REQUIRE(a->fmt >= 0 && a->fmt <= 2 && a->next_fmt >= 0 && a->next_fmt <= 2, "decode_chain: unknown weight format");
    MILA_REQUIRE(a->attn && a->res && a->res_out && a->y, "decode_chain: null activation pointer");
    MILA_REQUIRE(a->res_out != a->res, "decode_chain: res_out must not alias res");
    MILA_REQUIRE(a->W_o && a->W_gate_up && a->W_down && a->W_next, "decode_chain: null weight pointer");
    MILA_REQUIRE(a->post_attn_w && a->pre_ffn_w && a->post_ffn_w && a->next_norm_w, "decode_chain: null norm weight");
    if (a->fmt != FMT_BF16) MILA_REQUIRE(a->s_o && a->s_gate_up && a->s_down, "decode_chain: quantized weights need scales");
    if (a->next_fmt != FMT_BF16) MILA_REQUIRE(a->s_next != nullptr, "decode_chain: quantized next weights need scales");
    if (a->fmt == FMT_FP4)
        MILA_REQUIRE((a->group == 64 || a->group == 128) && D % a->group == 0 && F % a->group == 0 && a->K_attn % a->group == 0,
                     "decode_chain: fp4 group %d must be 64 or 128 and divide D, F, K_attn", a->group);
    if (a->next_fmt == FMT_FP4)
        MILA_REQUIRE((a->next_group == 64 || a->next_group == 128) && D % a->next_group == 0, "decode_chain: bad next_group %d", a->next_group);
    MILA_REQUIRE(a->scratch != nullptr && a->scratch_bytes >= mila_cdna4_decode_chain_scratch_bytes(D, F),
                 "decode_chain: scratch too small (%zu bytes, need %zu)", a->scratch_bytes, mila_cdna4_decode_chain_scratch_bytes(D, F));
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(a->scratch) & 15) == 0, "decode_chain: scratch must be 16-byte aligned");
    if (a->f32_out) MILA_REQUIRE(a->next_fmt != FMT_FP4, "decode_chain: the lm_head phase takes a bf16 or fp8 table");
    else MILA_REQUIRE(a->next_fmt == a->fmt && (a->fmt != FMT_FP4 || a->next_group == a->group), "decode_chain: the next qkv_proj must use the layer's weight format");

    unsigned char* sc = static_cast<unsigned char*>(a->scratch);
    uint32_t* h0 = reinterpret_cast<uint32_t*>(sc + kChainHeaderBytes);
    uint32_t* h1 = h0 + D;
    uint32_t* h2 = h1 + F;
    ChainParams c;
    // {y, x, W, scales, bias, norm_w, post_w, res, res_out, post_scale, eps, K, N, group}
    c.ph[0] = MatvecParams{h0, a->attn, static_cast<const uint8_t*>(a->W_o), a->s_o, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, a->eps, a->K_attn, D, a->group, 0, 0};
    c.ph[1] = MatvecParams{h1, reinterpret_cast<const uint16_t*>(h0), static_cast<const uint8_t*>(a->W_gate_up), a->s_gate_up, nullptr, a->pre_ffn_w, a->post_attn_w, a->res, nullptr, 1.0f, a->eps, D, F, a->group, 0, 0};
    c.ph[2] = MatvecParams{h2, reinterpret_cast<const uint16_t*>(h1), static_cast<const uint8_t*>(a->W_down), a->s_down, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, a->eps, F, D, a->group, 0, 0};
    c.ph[3] = MatvecParams{a->y, reinterpret_cast<const uint16_t*>(h2), static_cast<const uint8_t*>(a->W_next), a->s_next, nullptr, a->next_norm_w, a->post_ffn_w, nullptr, a->res_out, a->layer_scalar, a->eps, D, a->N_next, a->next_group, 0, 0};
    c.counter = reinterpret_cast<unsigned long long*>(sc);
    c.epoch = reinterpret_cast<unsigned long long*>(sc + 8);
    c.error = reinterpret_cast<uint32_t*>(sc + 16);
    c.nblocks = chain_num_blocks();
    hipStream_t s = as_stream(stream);
    if (a->f32_out)
    {
        if (a->fmt == FMT_BF16 && a->next_fmt == FMT_BF16) return launch_chain<FMT_BF16, true, FMT_BF16>(c, s);
        if (a->fmt == FMT_FP8 && a->next_fmt == FMT_FP8) return launch_chain<FMT_FP8, true, FMT_FP8>(c, s);
        if (a->fmt == FMT_FP4 && a->next_fmt == FMT_FP8) return launch_chain<FMT_FP4, true, FMT_FP8>(c, s);
        if (a->fmt == FMT_BF16 && a->next_fmt == FMT_FP8) return launch_chain<FMT_BF16, true, FMT_FP8>(c, s);
        return set_error(MILA_E_INVALID_ARGUMENT, "decode_chain: unsupported (fmt=%d, lm_head fmt=%d) pair", a->fmt, a->next_fmt);
    }
    switch (a->fmt)
    {
        case FMT_BF16: return launch_chain<FMT_BF16, false, FMT_BF16>(c, s);
        case FMT_FP8: return launch_chain<FMT_FP8, false, FMT_FP8>(c, s);
        default: return launch_chain<FMT_FP4, false, FMT_FP4>(c, s);
    }
}

}  // extern "C"
