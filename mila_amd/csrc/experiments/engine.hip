// Decode ENGINE (opt-in, measured SLOWER than the launch sequence on MI355X -- see the table below): the four Linears that sit
// between two attention calls of a Gemma decode step -- o_proj -> post-attention tail -> fc_gate_up + GeGLU -> fc_down -> post-FFN tail
// -> the next layer's input norm + qkv_proj (or the final norm + tied lm_head) (Components/Transformers/Gemma/Gemma.Block.ixx:287-356;
// matvec kernels CudaMatVecBias.Bf16.cu:134-508) -- as ONE persistent launch whose weight stream runs THROUGH its dependencies.
//
// Structure (MI355X_MICROARCH.md price list: ldsdma-fill, prefetch-credit, engine-vs-launches):
//   * every wave owns a RING in LDS (8 waves x 12 KiB per CU = 24 MB chip-wide) that it fills itself with LDS-DMA
//     (global_load_lds_dwordx4, non-temporal) from its share of ALL FOUR weight matrices, in order, as far ahead as the ring allows:
//     weights are constants, so the requests of phase p + 1 are in flight while phase p's outputs are still being handed over.
//     Completion is the wave's own vmcnt; there is no loader role and no LDS protocol;
//   * requests, waits, LDS reads and releases work on GROUPS of B 1-KiB pieces (one per-lane address, one ring slot, B immediate
//     offsets), all bookkeeping wave-uniform;
//   * a wave consumes its ring exactly as matvec_body consumes registers (lane l owns the 16-byte chunks l, l + 64, ... of a row in
//     ascending order, then the wave butterfly): bit-identical to the unfused kernels, whatever the shape (tests/test_engine_gpu.py);
//   * a phase's outputs reach every CU as 4-byte DATA-TAGGED granules {tag16 | bf16}: one write-through (sc1) store per element, no
//     flag, no fence, no counter; consumers sweep the vector with sc1 loads until every tag is the current one (and keep topping up
//     their rings between sweeps).
// Stream geometry: CU b, wave cw owns the output columns col_t = b + 256 (cw + NW t); a column is one weight row (GeGLU: the gate
// row col, then the up row N + col); a row's RECORD is [one piece of group scales (fp4: aligned 16-byte windows, lanes < spr)] + its
// weight pieces (the final piece only on its first `tail` lanes: the others are neither fetched nor used).
//
// Measured (tools/bench_engine.py, Gemma-3 12B layer, 8 cold weight sets, us per layer; tools/engine_timeline.py for the in-kernel
// stamps; profiles/r02_engine_*.txt):
//     form                                                        bf16     fp8     fp4     (four graph-captured launches: 75.6 / 47.0 / 34.5)
//     one loader wave + 7 consumer rings per CU                   262      163     128     loader issue-bound (1 KiB per ~250 cycles)
//     the same, LDS control words through typed LDS pointers      202      132     116     (generic pointers made them flat ops + vmcnt(0))
//     every wave its own loader, 15-16 waves, 1-KiB bookkeeping   124       93      70     younger waves of a SIMD starve (issue arbitration by age)
//     8 waves, 15-KiB rings, scalar bookkeeping                   107       83      75
//     8 waves, groups of 4 (3) pieces  [this file]                107       82      69
// What the stamps say about the last two: while streaming, a wave waits for its pieces only ~15 % of the time and sits ~40 % in the
// LDS-DMA requests themselves (the CU's memory pipeline is full: the same back-pressure a register-streaming kernel sees at its
// waitcnt), so the phases run at the launches' rate (236 MB in 35-41 us) -- but a phase ends when its SLOWEST wave does (54 / 63 / 74 us
// min / median / max for equal shares; a launch re-balances through the dispatcher, a static share cannot), and each hand-off adds
// 3-5 us after that.  With perfect balance the launch would still be ~82 us against 76.  The launch sequence stays the default;
// this file stays as the measured counter-example and is not used by GemmaTransformer.
//
// Every spin is bounded (error word, checked by the host); all 256 workgroups must be resident (one per CU: > 80 KiB of LDS each).
// Nothing is retained between launches except the epoch word that makes tags unique.
#include <algorithm>

#include "../common.h"
#include "../internal.h"
#include "../rms_common.h"
#include "../matvec_body.h"

namespace mila {

constexpr int kMaxGroups = 3;                       // 64-chunk groups of x a wave owns in a prologue at NW = 8: K <= 8 * 3 * 512
constexpr int kEngMaxSpins = 1 << 18;               // polls of one wait before it gives up; no clock reads on a polling path

struct EngPhase
{
    const uint8_t* W;
    const uint8_t* S;             // fp4: group scales [rows, ngroups] fp32 (as bytes); else null
    const float* row_scales;      // fp8: one per weight row
    void* y;                      // granule vector (uint32), bf16 row or fp32 row
    const void* x;                // phase 0: bf16 vector in plain memory; later phases: granule vector of the previous phase
    const uint16_t* norm_w;
    const uint16_t* post_w;
    const uint16_t* res;
    uint16_t* res_out;
    float post_scale, eps;
    int K, N;
    int cpr, spr, ngroups, cpg_shift, rpc;      // weight chunks / scale units per record, fp4 groups per row, chunks-per-group shift, rows per column
};

struct EngParams
{
    EngPhase ph[4];
    unsigned long long* epoch;    // launches completed before this one
    uint32_t* error;
    int xa_off, xb_off, ctrl_off; // LDS byte offsets behind the rings
    int nblocks;
    unsigned long long* debug;    // diagnostic builds only (mila_cdna4_decode_engine_debug): wall-clock stamps [block][wave][16]; null otherwise
};

__device__ __forceinline__ void eng_stamp(const EngParams& c, int b, int wave, int lane, int idx)
{
    if (c.debug != nullptr && lane == 0 && wave < 16) c.debug[((size_t)b * 16 + wave) * 16 + idx] = (unsigned long long)wall_clock64();
}

// control block in LDS: the two reduction arrays of the sandwich prologue (floats) and the issue table
enum { C_RED_A = 0, C_RED_B = 32, C_ISSUE = 64, C_WORDS = 64 + 48 };      // C_ISSUE: 4 EngIssue records (40 bytes each, 48 words)
typedef __attribute__((address_space(3))) volatile float* ctrl_fptr;

template <int NW>
__device__ __forceinline__ int eng_ncols(int N, int b, int cw)
{
    const int first = b + 256 * cw;
    return first < N ? (N - 1 - first) / (256 * NW) + 1 : 0;
}

template <int N> __device__ __forceinline__ void eng_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
// wait until at most n (0 .. 15) of this wave's vector-memory requests are outstanding; requests retire in order
__device__ __forceinline__ void eng_wait_vmcnt_at_most(int n)
{
    switch (n)
    {
        case 0: eng_wait_vmcnt<0>(); break;
        case 1: eng_wait_vmcnt<1>(); break;
        case 2: eng_wait_vmcnt<2>(); break;
        case 3: eng_wait_vmcnt<3>(); break;
        case 4: eng_wait_vmcnt<4>(); break;
        case 5: eng_wait_vmcnt<5>(); break;
        case 6: eng_wait_vmcnt<6>(); break;
        case 7: eng_wait_vmcnt<7>(); break;
        case 8: eng_wait_vmcnt<8>(); break;
        case 9: eng_wait_vmcnt<9>(); break;
        case 10: eng_wait_vmcnt<10>(); break;
        case 11: eng_wait_vmcnt<11>(); break;
        case 12: eng_wait_vmcnt<12>(); break;
        case 13: eng_wait_vmcnt<13>(); break;
        case 14: eng_wait_vmcnt<14>(); break;
        default: eng_wait_vmcnt<15>(); break;
    }
}

// The fields of a phase the issue side needs.  The issue side runs ahead of the consume side, so it picks its phase at run time; a
// dynamic index into the kernel-argument struct would make the compiler copy the whole struct to scratch, so the four records are
// copied once (constant indices) into LDS and read from there.
struct EngIssue
{
    const uint8_t* W;
    const uint8_t* S;
    int N, cpr, spr, ngroups, rpc, pad;
};
typedef __attribute__((address_space(3))) EngIssue* issue_tab_ptr;
__device__ __forceinline__ void eng_issue_store(issue_tab_ptr t, const EngPhase& P)
{
    t->W = P.W; t->S = P.S; t->N = P.N; t->cpr = P.cpr; t->spr = P.spr; t->ngroups = P.ngroups; t->rpc = P.rpc; t->pad = 0;
}
__device__ __forceinline__ uint64_t eng_uniform64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ EngIssue eng_issue_load(issue_tab_ptr tab, int p)
{
    issue_tab_ptr t = tab + p;
    EngIssue q;
    q.W = reinterpret_cast<const uint8_t*>(eng_uniform64(reinterpret_cast<uint64_t>(t->W)));
    q.S = reinterpret_cast<const uint8_t*>(eng_uniform64(reinterpret_cast<uint64_t>(t->S)));
    q.N = __builtin_amdgcn_readfirstlane(t->N); q.cpr = __builtin_amdgcn_readfirstlane(t->cpr); q.spr = __builtin_amdgcn_readfirstlane(t->spr);
    q.ngroups = __builtin_amdgcn_readfirstlane(t->ngroups); q.rpc = __builtin_amdgcn_readfirstlane(t->rpc); q.pad = 0;
    return q;
}

// ------------------------------------------------------------------------------------------------------------------------------
// A worker wave.  Its stream over all four phases is a sequence of records (one weight row each): [a scale piece (fp4)] + the row's
// weight pieces, a piece = 64 x 16 bytes = one LDS-DMA instruction.  Records are cut into groups of B pieces (the last one short); a
// group is the unit of requesting, waiting, reading and releasing, so the bookkeeping -- all of it wave-uniform, one per-lane address
// per group -- is paid once per B KiB.  The ring holds RG groups.
// ------------------------------------------------------------------------------------------------------------------------------
template <int NW, int B, int RG>
struct EngW
{
    static_assert(RG >= 2 && RG <= 4 && (RG - 1) * B <= 15, "ring shape");
    unsigned char* ring;          // RG * B KiB
    uint32_t* error;
    issue_tab_ptr itab;
    int cw, lane, b;
    // issue side
    int iphase, icols_left, icol, iwhich;
    int ig, igpr, inp_last, itail, ipre;      // group within the record; groups per record; pieces of its last group; lanes of its final piece; scale pieces (0/1)
    EngIssue ip;
    const uint8_t* iwrow;         // weight row of the record being requested
    const uint8_t* isrow;         // its aligned scale windows (fp4)
    int islot;
    uint32_t issued;              // DMA instructions so far
    uint32_t gissued;             // groups requested so far
    uint32_t cum0, cum1, cum2, cum3;          // `issued` right after the group in ring slot 0..3 was requested
    // consume side
    uint32_t gconsumed;           // groups whose ring space may be overwritten
    int cslot;
    uint32_t tag;                 // granule tag of the phase being PRODUCED
    bool failed;
    bool dbg;                     // diagnostic launches: account the time spent waiting for pieces
    unsigned long long stall, t_lds, t_issue, t_fin;

    __device__ __forceinline__ void set_record_row()
    {
        const int row = iwhich ? ip.N + icol : icol;
        iwrow = ip.W + (size_t)row * (size_t)ip.cpr * 16;
        isrow = reinterpret_cast<const uint8_t*>((reinterpret_cast<uintptr_t>(ip.S) + (uintptr_t)row * (uintptr_t)(ip.ngroups * 4)) & ~(uintptr_t)15);
    }
    __device__ __forceinline__ void begin_phase()
    {
        while (iphase < 4)
        {
            ip = eng_issue_load(itab, iphase);
            const int ncols = eng_ncols<NW>(ip.N, b, cw);
            if (ncols > 0)
            {
                icols_left = ncols - 1;
                icol = b + 256 * cw;
                iwhich = 0;
                ig = 0;
                const int steps = (ip.cpr + 63) >> 6;
                ipre = ip.spr > 0 ? 1 : 0;
                const int ppr = ipre + steps;
                igpr = (ppr + B - 1) / B;
                inp_last = ppr - (igpr - 1) * B;
                itail = ip.cpr - 64 * (steps - 1);
                set_record_row();
                return;
            }
            ++iphase;
        }
    }
    __device__ __forceinline__ void next_record()
    {
        ig = 0;
        if (ip.rpc == 2 && iwhich == 0) iwhich = 1;
        else
        {
            iwhich = 0;
            if (icols_left == 0) { ++iphase; begin_phase(); return; }
            --icols_left;
            icol += 256 * NW;
        }
        set_record_row();
    }
    template <int OFF>
    __device__ __forceinline__ void dma(const uint8_t* src, unsigned char* dst)
    {
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, OFF, 2 /* nt: read once */);
    }
    __device__ __forceinline__ void issue_group()
    {
        unsigned char* dst = ring + (size_t)islot * (B * 1024);
        // piece k of this group is piece ig * B + k of the record; weight piece m sits at row + 1024 m
        const uint8_t* src = iwrow + ((ptrdiff_t)(ig * B - ipre) * 1024 + lane * 16);
        const bool last = ig + 1 == igpr;
        if (!last && !(ipre && ig == 0))
        {
            dma<0>(src, dst);
            if constexpr (B > 1) dma<1024>(src, dst);
            if constexpr (B > 2) dma<2048>(src, dst);
            if constexpr (B > 3) dma<3072>(src, dst);
            issued += B;
        }
        else
        {
            const int np = last ? inp_last : B;
            const int fin = last ? np - 1 : -1;          // the record's final piece: only its first `itail` lanes hold weights
            const bool sc0 = ipre && ig == 0;
            if (sc0) { if (lane < ip.spr) dma<0>(isrow + lane * 16, dst); }
            else if (fin == 0) { if (lane < itail) dma<0>(src, dst); }
            else dma<0>(src, dst);
            if constexpr (B > 1)
                if (np > 1) { if (fin == 1) { if (lane < itail) dma<1024>(src, dst); } else dma<1024>(src, dst); }
            if constexpr (B > 2)
                if (np > 2) { if (fin == 2) { if (lane < itail) dma<2048>(src, dst); } else dma<2048>(src, dst); }
            if constexpr (B > 3)
                if (np > 3) { if (fin == 3) { if (lane < itail) dma<3072>(src, dst); } else dma<3072>(src, dst); }
            issued += (uint32_t)np;
        }
        cum0 = islot == 0 ? issued : cum0;
        cum1 = islot == 1 ? issued : cum1;
        if constexpr (RG > 2) cum2 = islot == 2 ? issued : cum2;
        if constexpr (RG > 3) cum3 = islot == 3 ? issued : cum3;
        islot = (islot + 1 == RG) ? 0 : islot + 1;
        ++gissued;
        if (++ig == igpr) next_record();
    }
    // request groups while the ring has room; never blocks
    __device__ __forceinline__ void top_up()
    {
        while (iphase < 4 && (int)(gissued - gconsumed) < RG) issue_group();
    }
    // the oldest group is in LDS on return
    __device__ __forceinline__ void need()
    {
        // (each field through readfirstlane: a plain select between fields becomes a select between their addresses and pins the whole struct in scratch)
        const uint32_t c0 = __builtin_amdgcn_readfirstlane(cum0), c1 = __builtin_amdgcn_readfirstlane(cum1);
        uint32_t cm = cslot == 0 ? c0 : c1;
        if constexpr (RG > 2) { const uint32_t c2 = __builtin_amdgcn_readfirstlane(cum2); cm = cslot == 2 ? c2 : cm; }
        if constexpr (RG > 3) { const uint32_t c3 = __builtin_amdgcn_readfirstlane(cum3); cm = cslot == 3 ? c3 : cm; }
        const int allowed = (int)(issued - cm);           // requests younger than the group's last
        unsigned long long t0 = 0ull;
        if (dbg) t0 = __builtin_readcyclecounter();
        if (allowed == (RG - 1) * B) eng_wait_vmcnt<(RG - 1) * B>();          // steady state: whole groups behind it
        else eng_wait_vmcnt_at_most(allowed);
        if (dbg) stall += __builtin_readcyclecounter() - t0;
    }
    // the oldest group is in registers: reuse its space (and request what goes there)
    __device__ __forceinline__ void release()
    {
        ++gconsumed;
        cslot = (cslot + 1 == RG) ? 0 : cslot + 1;
        top_up();
    }
    // Only LDS data crosses these barriers, so only LDS operations are waited for: __syncthreads() would also wait for vmcnt(0),
    // i.e. for every weight request in flight.  The ring is topped up first (a wave parked at s_barrier cannot issue).
    __device__ __forceinline__ void barrier()
    {
        top_up();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __device__ __forceinline__ bool timed_out(int& spins, uint32_t code)
    {
        if (++spins <= kEngMaxSpins) return false;
        __hip_atomic_store(as_global(error), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        failed = true;
        return true;
    }
};

__device__ __forceinline__ uint32_t eng_granule(uint32_t tag, uint32_t bf16_bits) { return (tag << 16) | (bf16_bits & 0xffffu); }

// Sweep `n` granules (n % 4 == 0) of a vector every producer workgroup writes with 4-byte write-through stores; the NW waves split its
// 16-byte units, each lane takes up to UPL of them per pass with all their loads in flight; elements land in `dst` (LDS) as bf16.
template <int NW, int B, int RG>
__device__ __forceinline__ void eng_gather(EngW<NW, B, RG>& s, const uint32_t* gv, int n, uint32_t want_tag, uint16_t* dst)
{
    constexpr int UPL = 8;
    const int nun = n >> 2;
    const int per = (nun + NW - 1) / NW;
    const int u0 = s.cw * per, u1 = min(nun, u0 + per);
    const gu64* q = as_global(reinterpret_cast<const unsigned long long*>(gv));
    if (s.failed) return;
    int spins = 0;
    for (int base = u0 + s.lane; __any(base < u1); base += 64 * UPL)
    {
        uint32_t pending = 0u;
#pragma unroll
        for (int k = 0; k < UPL; ++k) pending |= (base + 64 * k < u1) ? (1u << k) : 0u;
        for (;;)
        {
            unsigned long long lo[UPL], hi[UPL];
#pragma unroll
            for (int k = 0; k < UPL; ++k)
            {
                const int u = min(base + 64 * k, nun - 1);
                lo[k] = __hip_atomic_load(q + (size_t)u * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi[k] = __hip_atomic_load(q + (size_t)u * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < UPL; ++k)
            {
                if (!(pending & (1u << k))) continue;
                const uint32_t g0 = (uint32_t)lo[k], g1 = (uint32_t)(lo[k] >> 32), g2 = (uint32_t)hi[k], g3 = (uint32_t)(hi[k] >> 32);
                if ((g0 >> 16) == want_tag && (g1 >> 16) == want_tag && (g2 >> 16) == want_tag && (g3 >> 16) == want_tag)
                {
                    u32x2 v;
                    v[0] = (g0 & 0xffffu) | (g1 << 16);
                    v[1] = (g2 & 0xffffu) | (g3 << 16);
                    *reinterpret_cast<u32x2*>(dst + (size_t)(base + 64 * k) * 4) = v;
                    pending &= ~(1u << k);
                }
            }
            if (__all(pending == 0u)) break;
            s.top_up();
            __builtin_amdgcn_s_sleep(2);
            if (s.timed_out(spins, 400u)) return;
        }
    }
}

// Stage x of a phase into LDS as matvec_body does (same canonical reductions, same roundings), zero-padded to the chunk positions the
// streaming loop touches.  PRO: 0 = x as is; 2 = sandwich tail (post-norm, residual, layer scalar, next norm).  XSRC: plain memory
// (phase 0) or the previous phase's granules.  rkeep: residual in (RES_REG) / the tail's result out; chunk 64 (cw + NW k) + lane.
template <int NW, int B, int RG, int FMT, int PRO, int XSRC, int RSRC>
__device__ __forceinline__ void eng_stage_x(const EngParams& c, EngW<NW, B, RG>& s, const EngPhase& P, u32x4* xs, ctrl_fptr ctrl, uint32_t want_tag,
                                            u32x4 (&rkeep)[kMaxGroups], bool write_res_out)
{
    constexpr int EPC = Fmt<FMT>::kElemsPerChunk;
    constexpr int XPC = EPC / 8;
    constexpr int MG = (NW >= 12) ? 2 : kMaxGroups;      // groups per wave: K <= NW * MG * 512
    const int K = P.K, nx16 = K / 8;
    const int steps = (P.cpr + 63) / 64;
    const int nx16_pad = steps * 64 * XPC;
    const int G = (nx16 + 63) / 64;
    const int lane = s.lane, cw = s.cw;
    // operands that wait for nobody first
    u32x4 pnw[MG], ppw[MG], pres[MG];
    if constexpr (PRO == 2)
    {
#pragma unroll
        for (int k = 0; k < MG; ++k)
        {
            const size_t e = (size_t)min(64 * (cw + NW * k) + lane, nx16 - 1) * 8;
            pnw[k] = ld16(P.norm_w + e);
            ppw[k] = ld16(P.post_w + e);
            if constexpr (RSRC == RES_MEM) pres[k] = ld16(P.res + e);
            else pres[k] = rkeep[k];
        }
    }
    // zero padding behind the vector
    for (int i = nx16 + (cw * 64 + lane); i < nx16_pad; i += 64 * NW) xs[i] = u32x4{0u, 0u, 0u, 0u};
    if constexpr (XSRC == X_PLAIN)
    {
        const uint16_t* xp = reinterpret_cast<const uint16_t*>(P.x);
        for (int i = cw * 64 + lane; i < nx16; i += 64 * NW) xs[i] = ld16(xp + (size_t)i * 8);
    }
    else
        eng_gather<NW, B, RG>(s, reinterpret_cast<const uint32_t*>(P.x), K, want_tag, reinterpret_cast<uint16_t*>(xs));
    s.barrier();
    if constexpr (PRO == 2)
    {
        ctrl_fptr red_a = ctrl + C_RED_A;
        ctrl_fptr red_b = ctrl + C_RED_B;
        u32x4 a[MG];
#pragma unroll
        for (int k = 0; k < MG; ++k)
        {
            const int g = cw + NW * k, ch = 64 * g + lane;
            if (g >= G) continue;
            a[k] = xs[min(ch, nx16 - 1)];
            float ss = ch < nx16 ? sumsq8(a[k], 0.0f) : 0.0f;
            ss = wave_sum(ss);
            if (lane == 0) red_a[g] = ss;
        }
        s.barrier();
        float t = 0.0f;
        for (int g = 0; g < G; ++g) t += red_a[g];
        const float rstd_a = rsqrtf(t / (float)K + P.eps);
#pragma unroll
        for (int k = 0; k < MG; ++k)
        {
            const int g = cw + NW * k, ch = 64 * g + lane;
            if (g >= G) continue;
            rkeep[k] = sandwich_tail8(rms_apply8(a[k], ppw[k], rstd_a, 0.0f), pres[k], P.post_scale);
            if (write_res_out && ch < nx16) st16(P.res_out + (size_t)ch * 8, rkeep[k]);
            float ss = ch < nx16 ? sumsq8(rkeep[k], 0.0f) : 0.0f;
            ss = wave_sum(ss);
            if (lane == 0) red_b[g] = ss;
        }
        s.barrier();
        float t2 = 0.0f;
        for (int g = 0; g < G; ++g) t2 += red_b[g];
        const float rstd_r = rsqrtf(t2 / (float)K + P.eps);
#pragma unroll
        for (int k = 0; k < MG; ++k)
        {
            const int g = cw + NW * k, ch = 64 * g + lane;
            if (g >= G || ch >= nx16) continue;
            xs[ch] = rms_apply8(rkeep[k], pnw[k], rstd_r, 0.0f);
        }
        s.barrier();
    }
}

// stream this wave's records of one phase out of its ring
template <int NW, int B, int RG, int FMT, bool GEGLU, int YDST>
__device__ __forceinline__ void eng_stream(EngW<NW, B, RG>& s, const EngPhase& P, const u32x4* xs)
{
    constexpr int NR = GEGLU ? 2 : 1;
    constexpr int PRE = FMT == FMT_FP4 ? 1 : 0;  // scale pieces ahead of the weights
    constexpr int MAXSTEPS_FP4 = 8;              // K <= 16384
    const int lane = s.lane, cw = s.cw, b = s.b;
    const int cpr = P.cpr;
    const int steps = (cpr + 63) >> 6;
    const int ppr = PRE + steps;
    const int gpr = (ppr + B - 1) / B;
    const int np_last = ppr - (gpr - 1) * B;
    const int tail = cpr - 64 * (steps - 1);
    const int ncols = eng_ncols<NW>(P.N, b, cw);
    for (int t = 0; t < ncols; ++t)
    {
        const int col = b + 256 * (cw + NW * t);
        float acc[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j)
        {
            float a = 0.0f;
            float sc[MAXSTEPS_FP4];
            for (int g = 0; g < gpr; ++g)
            {
                s.need();
                const unsigned char* gp = s.ring + (size_t)s.cslot * (B * 1024) + lane * 16;
                const int np = g + 1 == gpr ? np_last : B;
                const int fin = g + 1 == gpr ? np - 1 : -1;
                u32x4 w[B];
#pragma unroll
                for (int k = 0; k < B; ++k)
                    if (k < np) w[k] = *reinterpret_cast<const u32x4*>(gp + k * 1024);
                if constexpr (PRE)
                {
                    if (g == 0)
                    {
                        // the record's first piece holds the row's aligned scale windows: this lane's scale of every step
                        const int row = j ? P.N + col : col;
                        const uint32_t mis = (uint32_t)(((size_t)row * (size_t)(P.ngroups * 4)) & 15u);
                        const unsigned char* p0 = s.ring + (size_t)s.cslot * (B * 1024);
#pragma unroll
                        for (int m = 0; m < MAXSTEPS_FP4; ++m)
                        {
                            const int cc = min(64 * m + lane, cpr - 1);
                            sc[m] = (m < steps) ? *reinterpret_cast<const float*>(p0 + mis + 4u * (uint32_t)(cc >> P.cpg_shift)) : 0.0f;
                        }
                    }
                }
                unsigned long long t0 = 0ull, t1 = 0ull;
                if (s.dbg) t0 = __builtin_readcyclecounter();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (s.dbg) { t1 = __builtin_readcyclecounter(); s.t_lds += t1 - t0; }
                s.release();
                if (s.dbg) s.t_issue += __builtin_readcyclecounter() - t1;
#pragma unroll
                for (int k = 0; k < B; ++k)
                {
                    if (k >= np) continue;
                    if (PRE && g == 0 && k == 0) continue;
                    const int m = g * B + k - PRE;               // step of the row
                    const int ch = 64 * m + lane;
                    u32x4 wk = w[k];
                    if (k == fin && lane >= tail) wk = u32x4{0u, 0u, 0u, 0u};    // those lanes were not fetched
                    if constexpr (FMT == FMT_FP4)
                    {
                        float scm = sc[0];
#pragma unroll
                        for (int mm = 1; mm < MAXSTEPS_FP4; ++mm) scm = (m == mm) ? sc[mm] : scm;
                        a = fmaf(scm, chunk_dot<FMT>(wk, xs, ch, 0.0f), a);
                    }
                    else
                        a = chunk_dot<FMT>(wk, xs, ch, a);
                }
            }
            acc[j] = a;
        }
        // finish: the arithmetic of matvec_body's finish()
        unsigned long long tf = 0ull;
        if (s.dbg) tf = __builtin_readcyclecounter();
#pragma unroll
        for (int j = 0; j < NR; ++j)
        {
            float v = wave_sum(acc[j]);
            const int row = __builtin_amdgcn_readfirstlane(j ? P.N + col : col);
            if constexpr (FMT == FMT_FP8) v = __builtin_bit_cast(float, sload32(P.row_scales + row)) * v;
            acc[j] = v;
        }
        if (lane == 0)
        {
            float v = acc[0];
            if constexpr (GEGLU)
            {
                const float g = round_bf16(acc[0]), up = round_bf16(acc[1]);
                v = gelu_tanh(g) * up;
            }
            if constexpr (YDST == Y_F32) reinterpret_cast<float*>(P.y)[col] = v;
            else if constexpr (YDST == Y_HANDOFF)
                __hip_atomic_store(as_global(reinterpret_cast<uint32_t*>(P.y)) + col, eng_granule(s.tag, f32_to_bf16_bits(v)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else reinterpret_cast<uint16_t*>(P.y)[col] = f32_to_bf16_bits(v);
        }
        if (s.dbg) s.t_fin += __builtin_readcyclecounter() - tf;
    }
}

// FMT: format of the four layer Linears; HEAD: the last phase is the tied lm_head (format HFMT, fp32 logits); NW waves, rings of RG groups of B KiB
template <int FMT, bool HEAD, int HFMT, int NW, int B, int RG>
__global__ __launch_bounds__(64 * NW) void decode_engine_kernel(const EngParams c)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    ctrl_fptr ctrl = (ctrl_fptr)(lds + c.ctrl_off);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const unsigned long long epoch = *c.epoch;
    issue_tab_ptr itab = (issue_tab_ptr)(ctrl + C_ISSUE);
    if (tid == 0) eng_issue_store(itab + 0, c.ph[0]);
    if (tid == 64) eng_issue_store(itab + 1, c.ph[1]);
    if (tid == 128) eng_issue_store(itab + 2, c.ph[2]);
    if (tid == 192) eng_issue_store(itab + 3, c.ph[3]);
    __syncthreads();
    EngW<NW, B, RG> s;
    s.itab = itab;
    s.cw = wave; s.lane = lane; s.b = b;
    s.ring = lds + (size_t)wave * (RG * B * 1024);
    s.error = c.error;
    s.dbg = c.debug != nullptr; s.stall = s.t_lds = s.t_issue = s.t_fin = 0ull;
    s.iphase = 0; s.issued = 0u; s.gissued = 0u; s.islot = 0; s.gconsumed = 0u; s.cslot = 0; s.failed = false; s.tag = 0u;
    s.cum0 = s.cum1 = s.cum2 = s.cum3 = 0u;
    s.begin_phase();
    s.top_up();                                           // the weight stream starts before anything else
    eng_stamp(c, b, wave, lane, 0);
    if (s.dbg && lane == 0) c.debug[((size_t)b * 16 + wave) * 16 + 13] = __builtin_readcyclecounter();
    u32x4* xa = reinterpret_cast<u32x4*>(lds + c.xa_off);
    u32x4* xb = reinterpret_cast<u32x4*>(lds + c.xb_off);
    // tags: three hand-offs per launch, never 0, unique against what the same words held after the previous launch
    auto tag_of = [&](int k) { return (uint32_t)((epoch * 3ull + (unsigned long long)k) % 65535ull) + 1u; };
    u32x4 r1[kMaxGroups], rk[kMaxGroups];
    // phase 0: a = o_proj(attn)
    eng_stage_x<NW, B, RG, FMT, 0, X_PLAIN, RES_MEM>(c, s, c.ph[0], xa, ctrl, 0u, rk, false);
    eng_stamp(c, b, wave, lane, 1);
    s.tag = tag_of(0);
    eng_stream<NW, B, RG, FMT, false, Y_HANDOFF>(s, c.ph[0], xa);
    eng_stamp(c, b, wave, lane, 2);
    // phase 1: r1 = res + rmsnorm(a); h = GeGLU(fc_gate_up(rmsnorm(r1)))
    eng_stage_x<NW, B, RG, FMT, 2, X_HANDOFF, RES_MEM>(c, s, c.ph[1], xb, ctrl, tag_of(0), r1, false);
    eng_stamp(c, b, wave, lane, 3);
    s.tag = tag_of(1);
    eng_stream<NW, B, RG, FMT, true, Y_HANDOFF>(s, c.ph[1], xb);
    eng_stamp(c, b, wave, lane, 4);
    // phase 2: d = fc_down(h)
    eng_stage_x<NW, B, RG, FMT, 0, X_HANDOFF, RES_MEM>(c, s, c.ph[2], xa, ctrl, tag_of(1), rk, false);
    eng_stamp(c, b, wave, lane, 5);
    s.tag = tag_of(2);
    eng_stream<NW, B, RG, FMT, false, Y_HANDOFF>(s, c.ph[2], xa);
    eng_stamp(c, b, wave, lane, 6);
    // phase 3: r2 = (r1 + rmsnorm(d)) * layer_scalar; y = next(rmsnorm(r2))
    if constexpr (HEAD)
    {
        eng_stage_x<NW, B, RG, HFMT, 2, X_HANDOFF, RES_REG>(c, s, c.ph[3], xb, ctrl, tag_of(2), r1, b == 0 && c.ph[3].res_out != nullptr);
        eng_stamp(c, b, wave, lane, 7);
        eng_stream<NW, B, RG, HFMT, false, Y_F32>(s, c.ph[3], xb);
    }
    else
    {
        eng_stage_x<NW, B, RG, FMT, 2, X_HANDOFF, RES_REG>(c, s, c.ph[3], xb, ctrl, tag_of(2), r1, b == 0 && c.ph[3].res_out != nullptr);
        eng_stamp(c, b, wave, lane, 7);
        eng_stream<NW, B, RG, FMT, false, Y_BF16>(s, c.ph[3], xb);
    }
    eng_stamp(c, b, wave, lane, 8);
    if (s.dbg && lane == 0)
    {
        unsigned long long* d = c.debug + ((size_t)b * 16 + wave) * 16;
        d[9] = s.stall; d[10] = s.t_lds; d[11] = s.t_issue; d[12] = s.t_fin;
    }
    if (s.dbg && lane == 0) c.debug[((size_t)b * 16 + wave) * 16 + 14] = __builtin_readcyclecounter();
    // every workgroup read `epoch` before any could pass hand-off 2, which workgroup 0 has now behind it
    if (b == 0 && wave == 0 && lane == 0) *c.epoch = epoch + 1ull;
}

static int eng_num_blocks()
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

static unsigned long long* g_eng_debug = nullptr;   // diagnostic hook (mila_cdna4_decode_engine_debug), inert unless the tuning hooks are enabled

constexpr size_t kEngHeaderBytes = 64;      // [0] unused (chain counter), [8] epoch, [16] error -- the chain's header layout

// waves per workgroup and the ring: RG groups of B pieces (KiB) per wave.  8 waves x 12 KiB + 38 KiB of x (fp4: 40)
template <int FMT> struct EngShape { static constexpr int NW = 8, B = 4, RG = 3; };
template <> struct EngShape<FMT_FP4> { static constexpr int NW = 8, B = 3, RG = 4; };      // records of 3 / 9 pieces at Gemma's widths
static size_t eng_ring_bytes(int fmt)
{
    return fmt == FMT_FP4 ? (size_t)EngShape<FMT_FP4>::NW * EngShape<FMT_FP4>::B * EngShape<FMT_FP4>::RG * 1024
                          : (size_t)EngShape<FMT_BF16>::NW * EngShape<FMT_BF16>::B * EngShape<FMT_BF16>::RG * 1024;
}

static int eng_x_units(int fmt, int K)
{
    const int epc = fmt == FMT_BF16 ? 8 : fmt == FMT_FP8 ? 16 : 32;
    const int cpr = K / epc, steps = (cpr + 63) / 64;
    return steps * 64 * (epc / 8);
}

static void eng_fill_phase(EngPhase& P, int fmt, int group, bool geglu)
{
    const int epc = fmt == FMT_BF16 ? 8 : fmt == FMT_FP8 ? 16 : 32;
    P.cpr = P.K / epc;
    P.rpc = geglu ? 2 : 1;
    if (fmt == FMT_FP4)
    {
        P.ngroups = P.K / group;
        P.cpg_shift = group == 128 ? 2 : 1;
        P.spr = (P.ngroups * 4 + 12 + 15) / 16;      // aligned 16-byte windows over a row's scales, whatever its misalignment
    }
    else { P.ngroups = 0; P.cpg_shift = 0; P.spr = 0; }
}

template <int FMT, bool HEAD, int HFMT>
static hipError_t eng_allow_big_lds()
{
    const void* fn = reinterpret_cast<const void*>(&decode_engine_kernel<FMT, HEAD, HFMT, EngShape<FMT>::NW, EngShape<FMT>::B, EngShape<FMT>::RG>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    // the data-tagged hand-offs need every workgroup resident: one must fit a CU with the whole LDS it may ask for
    int n = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64 * EngShape<FMT>::NW, 160 * 1024);
    if (e != hipSuccess) return e;
    return n >= 1 ? hipSuccess : hipErrorLaunchOutOfResources;
}
// every instantiation may use the CU's whole LDS; called from decode_engine_init (never inside a stream capture)
static int eng_prepare_all()
{
    static bool done = false;
    if (done) return MILA_OK;
    hipError_t e = eng_allow_big_lds<FMT_BF16, false, FMT_BF16>();
    if (e == hipSuccess) e = eng_allow_big_lds<FMT_FP8, false, FMT_FP8>();
    if (e == hipSuccess) e = eng_allow_big_lds<FMT_FP4, false, FMT_FP4>();
    if (e == hipSuccess) e = eng_allow_big_lds<FMT_BF16, true, FMT_BF16>();
    if (e == hipSuccess) e = eng_allow_big_lds<FMT_FP8, true, FMT_FP8>();
    if (e == hipSuccess) e = eng_allow_big_lds<FMT_FP4, true, FMT_FP8>();
    if (e == hipSuccess) e = eng_allow_big_lds<FMT_BF16, true, FMT_FP8>();
    if (e != hipSuccess) return check_hip(e, "decode_engine: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    done = true;
    return MILA_OK;
}

template <int FMT, bool HEAD, int HFMT>
static int launch_engine(const EngParams& c, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL((decode_engine_kernel<FMT, HEAD, HFMT, EngShape<FMT>::NW, EngShape<FMT>::B, EngShape<FMT>::RG>), dim3(c.nblocks), dim3(64 * EngShape<FMT>::NW), lds, s, c);
    MILA_LAUNCH_CHECK("decode_engine");
}

}  // namespace mila

using namespace mila;

extern "C" {

size_t mila_cdna4_decode_engine_scratch_bytes(int D, int F)
{
    if (D <= 0 || F <= 0) return 0;
    return kEngHeaderBytes + (size_t)(2 * D + F) * 4 + 64;
}

int mila_cdna4_decode_engine_init(void* scratch, size_t scratch_bytes, mila_stream_t stream)
{
    MILA_REQUIRE(scratch != nullptr && scratch_bytes >= kEngHeaderBytes, "decode_engine_init: scratch too small");
    (void)eng_num_blocks();   // device query outside any later stream capture
    int rc = eng_prepare_all();
    if (rc) return rc;
    return check_hip(hipMemsetAsync(scratch, 0, scratch_bytes, as_stream(stream)), "decode_engine_init");
}

int mila_cdna4_decode_engine_status(const void* scratch, int32_t* error_out, mila_stream_t stream)
{
    MILA_REQUIRE(scratch != nullptr && error_out != nullptr, "decode_engine_status: null pointer");
    hipStream_t s = as_stream(stream);
    uint32_t e = 0;
    int rc = check_hip(hipMemcpyAsync(&e, static_cast<const unsigned char*>(scratch) + 16, 4, hipMemcpyDeviceToHost, s), "decode_engine_status");
    if (rc) return rc;
    rc = check_hip(hipStreamSynchronize(s), "decode_engine_status");
    if (rc) return rc;
    *error_out = (int32_t)e;
    return MILA_OK;
}

/* 1 when the engine serves this geometry: 256 compute units, every output width a multiple of 256 columns would be ideal but is not
 * required; K limits come from the x buffers in LDS and the prologue's per-wave share */
int mila_cdna4_decode_engine_applicable(int fmt, int group, int D, int F, int K_attn, int N_next, int next_fmt)
{
    if (fmt < 0 || fmt > 2 || next_fmt < 0 || next_fmt > 2) return 0;
    if (D <= 0 || F <= 0 || K_attn <= 0 || N_next <= 0) return 0;
    if (D % 32 || F % 32 || K_attn % 32) return 0;
    if (D > 8 * kMaxGroups * 512 || K_attn > 16384 || F > 16384) return 0;
    if (fmt == FMT_FP4)
    {
        if (!((group == 64 || group == 128) && D % group == 0 && F % group == 0 && K_attn % group == 0)) return 0;
        // every scale array ends on a 16-byte boundary (the loader fetches aligned 16-byte windows): rows x groups x 4 bytes
        if (((size_t)D * (K_attn / group) * 4) % 16 || ((size_t)2 * F * (D / group) * 4) % 16 || ((size_t)D * (F / group) * 4) % 16) return 0;
        if (next_fmt == FMT_FP4 && ((size_t)N_next * (D / group) * 4) % 16) return 0;
    }
    if (eng_num_blocks() != kNumCU) return 0;
    const size_t xa = (size_t)std::max(eng_x_units(fmt, K_attn), eng_x_units(fmt, F)) * 16;
    const size_t xb = (size_t)std::max(eng_x_units(fmt, D), eng_x_units(next_fmt, D)) * 16;
    return eng_ring_bytes(fmt) + xa + xb + C_WORDS * 4 <= 160 * 1024 ? 1 : 0;
}

/* diagnostic hook: wall-clock stamps of the first 8 workgroups' waves into `buf` (8 x 8 x 16 uint64) on the following launches */
int mila_cdna4_decode_engine_debug(unsigned long long* buf)
{
    if (!::mila::tuning_hooks_enabled()) return ::mila::set_error(MILA_E_UNSUPPORTED, "decode_engine_debug: inert unless MILA_CDNA4_TUNING=1 was set when the library was loaded");
    g_eng_debug = buf;
    return MILA_OK;
}

int mila_cdna4_decode_engine(const mila_decode_chain_args* a, mila_stream_t stream)
{
    MILA_REQUIRE(a != nullptr, "decode_engine: null args");
    const int D = a->D, F = a->F;
    MILA_REQUIRE(D > 0 && F > 0 && a->K_attn > 0 && a->N_next > 0, "decode_engine: dimensions must be positive");
    MILA_REQUIRE(a->fmt >= 0 && a->fmt <= 2 && a->next_fmt >= 0 && a->next_fmt <= 2, "decode_engine: unknown weight format");
    MILA_REQUIRE(mila_cdna4_decode_engine_applicable(a->fmt, a->group, D, F, a->K_attn, a->N_next, a->next_fmt),
                 "decode_engine: geometry outside the engine (fmt=%d group=%d D=%d F=%d K_attn=%d): ask decode_engine_applicable", a->fmt, a->group, D, F, a->K_attn);
    MILA_REQUIRE(a->attn && a->res && a->res_out && a->y, "decode_engine: null activation pointer");
    MILA_REQUIRE(a->res_out != a->res, "decode_engine: res_out must not alias res");
    MILA_REQUIRE(a->W_o && a->W_gate_up && a->W_down && a->W_next, "decode_engine: null weight pointer");
    MILA_REQUIRE(a->post_attn_w && a->pre_ffn_w && a->post_ffn_w && a->next_norm_w, "decode_engine: null norm weight");
    if (a->fmt != FMT_BF16) MILA_REQUIRE(a->s_o && a->s_gate_up && a->s_down, "decode_engine: quantized weights need scales");
    if (a->next_fmt != FMT_BF16) MILA_REQUIRE(a->s_next != nullptr, "decode_engine: quantized next weights need scales");
    if (a->f32_out) MILA_REQUIRE(a->next_fmt != FMT_FP4, "decode_engine: the lm_head phase takes a bf16 or fp8 table");
    else MILA_REQUIRE(a->next_fmt == a->fmt && (a->fmt != FMT_FP4 || a->next_group == a->group), "decode_engine: the next qkv_proj must use the layer's weight format");
    MILA_REQUIRE(a->scratch != nullptr && a->scratch_bytes >= mila_cdna4_decode_engine_scratch_bytes(D, F),
                 "decode_engine: scratch too small (%zu bytes, need %zu)", a->scratch_bytes, mila_cdna4_decode_engine_scratch_bytes(D, F));
    MILA_REQUIRE((reinterpret_cast<uintptr_t>(a->scratch) & 15) == 0, "decode_engine: scratch must be 16-byte aligned");
    for (const void* w : {a->W_o, a->W_gate_up, a->W_down, a->W_next})
        MILA_REQUIRE((reinterpret_cast<uintptr_t>(w) & 15) == 0, "decode_engine: weight pointers must be 16-byte aligned");

    unsigned char* sc = static_cast<unsigned char*>(a->scratch);
    uint32_t* h0 = reinterpret_cast<uint32_t*>(sc + kEngHeaderBytes);
    uint32_t* h1 = h0 + D;
    uint32_t* h2 = h1 + F;
    EngParams c{};
    auto phase = [](EngPhase& P, const void* W, const float* scales, int fmt, void* y, const void* x, const uint16_t* norm_w, const uint16_t* post_w,
                    const uint16_t* res, uint16_t* res_out, float post_scale, float eps, int K, int N) {
        P.W = static_cast<const uint8_t*>(W);
        P.S = fmt == FMT_FP4 ? reinterpret_cast<const uint8_t*>(scales) : nullptr;
        P.row_scales = fmt == FMT_FP8 ? scales : nullptr;
        P.y = y; P.x = x; P.norm_w = norm_w; P.post_w = post_w; P.res = res; P.res_out = res_out; P.post_scale = post_scale; P.eps = eps; P.K = K; P.N = N;
    };
    phase(c.ph[0], a->W_o, a->s_o, a->fmt, h0, a->attn, nullptr, nullptr, nullptr, nullptr, 1.0f, a->eps, a->K_attn, D);
    phase(c.ph[1], a->W_gate_up, a->s_gate_up, a->fmt, h1, h0, a->pre_ffn_w, a->post_attn_w, a->res, nullptr, 1.0f, a->eps, D, F);
    phase(c.ph[2], a->W_down, a->s_down, a->fmt, h2, h1, nullptr, nullptr, nullptr, nullptr, 1.0f, a->eps, F, D);
    phase(c.ph[3], a->W_next, a->s_next, a->next_fmt, a->y, h2, a->next_norm_w, a->post_ffn_w, nullptr, a->res_out, a->layer_scalar, a->eps, D, a->N_next);
    eng_fill_phase(c.ph[0], a->fmt, a->group, false);
    eng_fill_phase(c.ph[1], a->fmt, a->group, true);
    eng_fill_phase(c.ph[2], a->fmt, a->group, false);
    eng_fill_phase(c.ph[3], a->next_fmt, a->next_group, false);
    for (int p = 0; p < 4; ++p)
    {
        MILA_REQUIRE(c.ph[p].cpr >= 1, "decode_engine: phase %d has no whole chunk per row", p);
        if (c.ph[p].S) MILA_REQUIRE(((size_t)c.ph[p].N * c.ph[p].rpc * c.ph[p].ngroups * 4) % 16 == 0 && (reinterpret_cast<uintptr_t>(c.ph[p].S) & 15) == 0,
                                    "decode_engine: phase %d: the fp4 scale array must start and end on 16-byte boundaries", p);
    }
    c.epoch = reinterpret_cast<unsigned long long*>(sc + 8);
    c.error = reinterpret_cast<uint32_t*>(sc + 16);
    c.nblocks = eng_num_blocks();
    c.debug = g_eng_debug;
    const size_t xa = (size_t)std::max(eng_x_units(a->fmt, a->K_attn), eng_x_units(a->fmt, F)) * 16;
    const size_t xb = (size_t)std::max(eng_x_units(a->fmt, D), eng_x_units(a->next_fmt, D)) * 16;
    c.xa_off = (int)eng_ring_bytes(a->fmt);
    c.xb_off = c.xa_off + (int)xa;
    c.ctrl_off = c.xb_off + (int)xb;
    const size_t lds = (size_t)c.ctrl_off + C_WORDS * 4;
    int rc = eng_prepare_all();
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    if (a->f32_out)
    {
        if (a->fmt == FMT_BF16 && a->next_fmt == FMT_BF16) return launch_engine<FMT_BF16, true, FMT_BF16>(c, lds, s);
        if (a->fmt == FMT_FP8 && a->next_fmt == FMT_FP8) return launch_engine<FMT_FP8, true, FMT_FP8>(c, lds, s);
        if (a->fmt == FMT_FP4 && a->next_fmt == FMT_FP8) return launch_engine<FMT_FP4, true, FMT_FP8>(c, lds, s);
        if (a->fmt == FMT_BF16 && a->next_fmt == FMT_FP8) return launch_engine<FMT_BF16, true, FMT_FP8>(c, lds, s);
        return set_error(MILA_E_INVALID_ARGUMENT, "decode_engine: unsupported (fmt=%d, lm_head fmt=%d) pair", a->fmt, a->next_fmt);
    }
    switch (a->fmt)
    {
        case FMT_BF16: return launch_engine<FMT_BF16, false, FMT_BF16>(c, lds, s);
        case FMT_FP8: return launch_engine<FMT_FP8, false, FMT_FP8>(c, lds, s);
        default: return launch_engine<FMT_FP4, false, FMT_FP4>(c, lds, s);
    }
}

}  // extern "C"
