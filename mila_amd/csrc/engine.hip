// Decode ENGINE: the four Linears that sit between two attention calls of a Gemma decode step -- o_proj -> post-attention tail ->
// fc_gate_up + GeGLU -> fc_down -> post-FFN tail -> the next layer's input norm + qkv_proj (or the final norm + tied lm_head)
// (Components/Transformers/Gemma/Gemma.Block.ixx:287-356; matvec kernels CudaMatVecBias.Bf16.cu:134-508) -- as ONE persistent launch
// whose weight stream never stops at a dependency.
//
// Why another form after chain.hip (which was slower than the launches it replaced): there every wave streamed its own rows into
// registers, so at most two pipeline steps (32 KB per CU, ~1.3 us of HBM time) could run ahead of a grid-wide hand-off that took
// ~8 us.  Here (MI355X_MICROARCH.md price list: ldsdma-fill, prefetch-credit, allgather, engine-vs-launches):
//   * one LOADER wave per CU moves the CU's share of all four weight matrices, in order, HBM -> LDS with LDS-DMA
//     (global_load_lds_dwordx4, non-temporal), into seven 16-KiB rings -- one per CONSUMER wave -- and never waits for a dependency:
//     weights are constants.  112 KiB per CU = 28 MB chip-wide = ~4.5 us of HBM time run ahead of every hand-off;
//   * seven consumer waves read their ring (ds_read_b128), dequantize and accumulate exactly as matvec_body does (lane l owns the
//     16-byte chunks l, l + 64, ... of a row in ascending order, then the wave butterfly): bit-identical to the unfused kernels;
//   * a phase's outputs go to every CU as 4-byte DATA-TAGGED granules {tag16 | bf16}: one write-through (sc1) store per element, no
//     flag, no fence, no counter; a consumer sweeps the vector with sc1 loads until every tag is the current one.  One fabric round
//     trip after the last producer's store, under the loader's cover.
// Stream geometry (shared by loader and consumers): CU b, consumer wave cw owns the output columns col_t = b + 256 (cw + 7 t);
// a column is one weight row (GeGLU: the gate row col, then the up row N + col); a row's RECORD is [spr scale units | cpr weight
// units] in 16-byte units (fp4: the row's group scales, fetched as aligned 16-byte windows; else spr = 0); a wave's records follow
// each other with no padding, a phase ends on a 1-KiB piece boundary.  Any lane can fetch any unit (LDS-DMA takes per-lane addresses).
//
// Every spin is bounded by the wall clock (error word, checked by the host); all 256 workgroups must be resident (one per CU: the
// kernel asks for > 80 KiB of LDS).  Nothing here is retained between launches except the epoch word that makes tags unique.
#include <algorithm>

#include "common.h"
#include "internal.h"
#include "rms_common.h"
#include "matvec_body.h"

namespace mila {

constexpr int kEngCons = 7;                         // consumer waves (wave 0 of the workgroup is the loader)
constexpr int kEngThreads = 64 * (kEngCons + 1);
constexpr int kRingUnits = 1024;                    // 16-byte units per consumer ring (16 KiB)
constexpr int kRingPieces = kRingUnits / 64;
constexpr int kBurst = 4;                           // pieces the loader issues per ring and round
constexpr int kMaxGroups = 3;                       // 64-chunk groups of x a consumer wave owns in a prologue: K <= 7 * 3 * 512
constexpr long long kEngSpinTicks = 2000000;        // wall_clock64 at 100 MHz: 20 ms

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
};

// control block in LDS (uint32 words)
enum { C_FULL = 0, C_FREE = 8, C_BAR = 16, C_RED_A = 32, C_RED_B = 48, C_WORDS = 64 };

__device__ __forceinline__ uint32_t lds_ld(volatile uint32_t* p) { return *p; }
__device__ __forceinline__ void lds_st(volatile uint32_t* p, uint32_t v) { *p = v; }

__device__ __forceinline__ bool eng_timed_out(long long t0, uint32_t* error, uint32_t code)
{
    if (wall_clock64() - t0 <= kEngSpinTicks) return false;
    __hip_atomic_store(as_global(error), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// columns of consumer wave cw in a phase, its records and pieces
__device__ __forceinline__ int eng_ncols(int N, int b, int cw)
{
    const int first = b + 256 * cw;
    return first < N ? (N - 1 - first) / (256 * kEngCons) + 1 : 0;
}
__device__ __forceinline__ int eng_npieces(const EngPhase& P, int b, int cw)
{
    const long long units = (long long)eng_ncols(P.N, b, cw) * P.rpc * (P.cpr + P.spr);
    return (int)((units + 63) / 64);
}

// ------------------------------------------------------------------------------------------------------------------------------
// loader wave
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void eng_wait_vmcnt_at_most(int n)
{
    switch (n >> 2)
    {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    }
}

// One lane's cursor in one ring's stream: the record and the unit inside it that the lane fetches in the ring's NEXT piece, the
// address of that unit, and how many more pieces the lane can fetch by just adding 1 KiB (it stays inside one contiguous region: a
// row's weights, or its scale windows).  Most pieces take that fast path; a lane that leaves its region recomputes from (rec, pos).
struct EngCur
{
    const uint8_t* addr;
    int left;      // units from this lane's position to the end of its region
    int pos, rec;
};

__device__ __forceinline__ void eng_cur_locate(EngCur& cu, const EngPhase& P, int b, int w, int nrec, int upr, int rshift)
{
    while (cu.pos >= upr) { cu.pos -= upr; ++cu.rec; }
    if (cu.rec >= nrec)
    {
        // padding behind the stream's last record (the rest of the phase's last piece): re-read a valid unit; nobody consumes it
        cu.addr = P.W;
        cu.left = 1 << 30;
        return;
    }
    const int col = b + 256 * (w + kEngCons * (cu.rec >> rshift));
    const int row = (cu.rec & (P.rpc - 1)) ? P.N + col : col;
    if (cu.pos < P.spr)
    {
        const uintptr_t s0 = reinterpret_cast<uintptr_t>(P.S) + (uintptr_t)row * (uintptr_t)(P.ngroups * 4);
        cu.addr = reinterpret_cast<const uint8_t*>((s0 & ~(uintptr_t)15) + (uintptr_t)cu.pos * 16);
        cu.left = P.spr - cu.pos;
    }
    else
    {
        cu.addr = P.W + ((size_t)row * (size_t)P.cpr + (size_t)(cu.pos - P.spr)) * 16;
        cu.left = upr - cu.pos;
    }
}

__device__ void eng_loader(const EngParams& c, unsigned char* lds, volatile uint32_t* ctrl, int b, int lane)
{
    uint32_t issued[kEngCons], seen_free[kEngCons], published[kEngCons], prev_issued[kEngCons];
#pragma unroll
    for (int w = 0; w < kEngCons; ++w) issued[w] = seen_free[w] = published[w] = prev_issued[w] = 0u;

    for (int p = 0; p < 4; ++p)
    {
        const EngPhase& P = c.ph[p];
        const int upr = P.cpr + P.spr;
        const int rshift = P.rpc == 2 ? 1 : 0;
        int rem[kEngCons], nrec[kEngCons];
        EngCur cur[kEngCons];
#pragma unroll
        for (int w = 0; w < kEngCons; ++w)
        {
            rem[w] = eng_npieces(P, b, w);
            nrec[w] = eng_ncols(P.N, b, w) * P.rpc;
            cur[w].rec = 0;
            cur[w].pos = lane;
            eng_cur_locate(cur[w], P, b, w, nrec[w], upr, rshift);
        }
        bool any = true;
        long long t0 = wall_clock64();
        while (any)
        {
            any = false;
            int round_cnt = 0;
#pragma unroll
            for (int w = 0; w < kEngCons; ++w)
            {
                if (rem[w] <= 0) continue;
                any = true;
                int space = kRingPieces - (int)(issued[w] - seen_free[w]);
                if (space < kBurst)
                {
                    seen_free[w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld(ctrl + C_FREE + w));
                    space = kRingPieces - (int)(issued[w] - seen_free[w]);
                }
                const int n = min(min(kBurst, rem[w]), space);
                for (int k = 0; k < n; ++k)
                {
                    unsigned char* dst = lds + (size_t)w * (kRingUnits * 16) + (size_t)((issued[w] + (uint32_t)k) & (kRingPieces - 1)) * 1024;
                    __builtin_amdgcn_global_load_lds(cur[w].addr, (__attribute__((address_space(3))) void*)dst, 16, 0, 2 /* nt: read once */);
                    cur[w].pos += 64;
                    cur[w].left -= 64;
                    cur[w].addr += 1024;
                    if (__any(cur[w].left <= 0))
                    {
                        if (cur[w].left <= 0) eng_cur_locate(cur[w], P, b, w, nrec[w], upr, rshift);
                    }
                }
                issued[w] += (uint32_t)n;
                rem[w] -= n;
                round_cnt += n;
            }
            // everything issued before this round has landed once at most round_cnt requests are outstanding
            eng_wait_vmcnt_at_most(round_cnt);
#pragma unroll
            for (int w = 0; w < kEngCons; ++w)
                if (published[w] != prev_issued[w]) { published[w] = prev_issued[w]; lds_st(ctrl + C_FULL + w, published[w]); }
#pragma unroll
            for (int w = 0; w < kEngCons; ++w) prev_issued[w] = issued[w];
            if (round_cnt == 0 && any)
            {
                __builtin_amdgcn_s_sleep(4);
                if (eng_timed_out(t0, c.error, 100u + (uint32_t)p)) return;
            }
            else t0 = wall_clock64();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int w = 0; w < kEngCons; ++w) lds_st(ctrl + C_FULL + w, issued[w]);
}

// ------------------------------------------------------------------------------------------------------------------------------
// consumer waves
// ------------------------------------------------------------------------------------------------------------------------------
struct EngCons
{
    unsigned char* ring;          // this wave's ring
    volatile uint32_t* ctrl;
    uint32_t* error;
    int cw, lane, b;
    uint32_t base_piece;          // pieces of earlier phases
    uint32_t full_seen;
    uint32_t bar_gen;
    uint32_t tag;                 // granule tag of the phase being PRODUCED
    bool failed;
};

// barrier among the consumer waves only (the loader never stops): an LDS arrival counter
__device__ __forceinline__ void eng_cbar(EngCons& s)
{
    s.bar_gen += kEngCons;
    if (s.failed) return;             // one give-up is enough: the rest of the launch runs through without waiting
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (s.lane == 0) __hip_atomic_fetch_add(const_cast<uint32_t*>(s.ctrl + C_BAR), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const long long t0 = wall_clock64();
    while ((int)((uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld(s.ctrl + C_BAR)) - s.bar_gen) < 0)
    {
        __builtin_amdgcn_s_sleep(1);
        if (eng_timed_out(t0, s.error, 200u)) { s.failed = true; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void eng_wait_full(EngCons& s, uint32_t piece)
{
    if ((int)(s.full_seen - piece) > 0 || s.failed) return;
    const long long t0 = wall_clock64();
    for (;;)
    {
        s.full_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld(s.ctrl + C_FULL + s.cw));
        if ((int)(s.full_seen - piece) > 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (eng_timed_out(t0, s.error, 300u + (uint32_t)s.cw)) { s.failed = true; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ uint32_t eng_granule(uint32_t tag, uint32_t bf16_bits) { return (tag << 16) | (bf16_bits & 0xffffu); }

// Sweep `n` granules (n % 4 == 0) of a vector every producer workgroup writes with 4-byte write-through stores; the seven consumer
// waves split its 16-byte units; elements land in `dst` (LDS) as bf16.  Returns when every tag of this wave's share matched.
__device__ void eng_gather(EngCons& s, const uint32_t* gv, int n, uint32_t want_tag, uint16_t* dst)
{
    const int nun = n >> 2;
    const int per = (nun + kEngCons - 1) / kEngCons;
    const int u0 = s.cw * per, u1 = min(nun, u0 + per);
    const gu64* q = as_global(reinterpret_cast<const unsigned long long*>(gv));
    if (s.failed) return;
    const long long t0 = wall_clock64();
    for (int base = u0 + s.lane; __any(base < u1); base += 64 * 4)
    {
        // four units per lane and pass; a unit is kept once all four of its tags match
        bool done[4] = {false, false, false, false};
        for (;;)
        {
            bool all_ok = true;
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const int u = base + 64 * k;
                if (u >= u1 || done[k]) continue;
                const unsigned long long lo = __hip_atomic_load(q + (size_t)u * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(q + (size_t)u * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t g0 = (uint32_t)lo, g1 = (uint32_t)(lo >> 32), g2 = (uint32_t)hi, g3 = (uint32_t)(hi >> 32);
                const bool ok = (g0 >> 16) == want_tag && (g1 >> 16) == want_tag && (g2 >> 16) == want_tag && (g3 >> 16) == want_tag;
                if (ok)
                {
                    u32x2 v;
                    v[0] = (g0 & 0xffffu) | (g1 << 16);
                    v[1] = (g2 & 0xffffu) | (g3 << 16);
                    *reinterpret_cast<u32x2*>(dst + (size_t)u * 4) = v;
                    done[k] = true;
                }
                else all_ok = false;
            }
            if (__all(all_ok)) break;
            __builtin_amdgcn_s_sleep(8);
            if (eng_timed_out(t0, s.error, 400u)) { s.failed = true; return; }
        }
    }
}

// Stage x of a phase into LDS as matvec_body does (same canonical reductions, same roundings), zero-padded to the chunk positions the
// streaming loop touches.  PRO: 0 = x as is; 2 = sandwich tail (post-norm, residual, layer scalar, next norm).  XSRC: plain memory
// (phase 0) or the previous phase's granules.  rkeep: residual in (RES_REG) / the tail's result out; chunk 64 (cw + 7 k) + lane.
template <int FMT, int PRO, int XSRC, int RSRC>
__device__ void eng_stage_x(EngCons& s, const EngPhase& P, u32x4* xs, uint32_t want_tag, u32x4 (&rkeep)[kMaxGroups], bool write_res_out)
{
    constexpr int EPC = Fmt<FMT>::kElemsPerChunk;
    constexpr int XPC = EPC / 8;
    const int K = P.K, nx16 = K / 8;
    const int steps = (P.cpr + 63) / 64;
    const int nx16_pad = steps * 64 * XPC;
    const int G = (nx16 + 63) / 64;
    const int lane = s.lane, cw = s.cw;
    // operands that wait for nobody first
    u32x4 pnw[kMaxGroups], ppw[kMaxGroups], pres[kMaxGroups];
    if constexpr (PRO == 2)
    {
#pragma unroll
        for (int k = 0; k < kMaxGroups; ++k)
        {
            const size_t e = (size_t)min(64 * (cw + kEngCons * k) + lane, nx16 - 1) * 8;
            pnw[k] = ld16(P.norm_w + e);
            ppw[k] = ld16(P.post_w + e);
            if constexpr (RSRC == RES_MEM) pres[k] = ld16(P.res + e);
            else pres[k] = rkeep[k];
        }
    }
    // zero padding behind the vector
    for (int i = nx16 + (cw * 64 + lane); i < nx16_pad; i += 64 * kEngCons) xs[i] = u32x4{0u, 0u, 0u, 0u};
    if constexpr (XSRC == X_PLAIN)
    {
        const uint16_t* xp = reinterpret_cast<const uint16_t*>(P.x);
        for (int i = cw * 64 + lane; i < nx16; i += 64 * kEngCons) xs[i] = ld16(xp + (size_t)i * 8);
    }
    else
        eng_gather(s, reinterpret_cast<const uint32_t*>(P.x), K, want_tag, reinterpret_cast<uint16_t*>(xs));
    eng_cbar(s);
    if constexpr (PRO == 2)
    {
        volatile float* red_a = reinterpret_cast<volatile float*>(s.ctrl + C_RED_A);
        volatile float* red_b = reinterpret_cast<volatile float*>(s.ctrl + C_RED_B);
        u32x4 a[kMaxGroups];
#pragma unroll
        for (int k = 0; k < kMaxGroups; ++k)
        {
            const int g = cw + kEngCons * k, ch = 64 * g + lane;
            if (g >= G) continue;
            a[k] = xs[min(ch, nx16 - 1)];
            float ss = ch < nx16 ? sumsq8(a[k], 0.0f) : 0.0f;
            ss = wave_sum(ss);
            if (lane == 0) red_a[g] = ss;
        }
        eng_cbar(s);
        float t = 0.0f;
        for (int g = 0; g < G; ++g) t += red_a[g];
        const float rstd_a = rsqrtf(t / (float)K + P.eps);
#pragma unroll
        for (int k = 0; k < kMaxGroups; ++k)
        {
            const int g = cw + kEngCons * k, ch = 64 * g + lane;
            if (g >= G) continue;
            rkeep[k] = sandwich_tail8(rms_apply8(a[k], ppw[k], rstd_a, 0.0f), pres[k], P.post_scale);
            if (write_res_out && ch < nx16) st16(P.res_out + (size_t)ch * 8, rkeep[k]);
            float ss = ch < nx16 ? sumsq8(rkeep[k], 0.0f) : 0.0f;
            ss = wave_sum(ss);
            if (lane == 0) red_b[g] = ss;
        }
        eng_cbar(s);
        float t2 = 0.0f;
        for (int g = 0; g < G; ++g) t2 += red_b[g];
        const float rstd_r = rsqrtf(t2 / (float)K + P.eps);
#pragma unroll
        for (int k = 0; k < kMaxGroups; ++k)
        {
            const int g = cw + kEngCons * k, ch = 64 * g + lane;
            if (g >= G || ch >= nx16) continue;
            xs[ch] = rms_apply8(rkeep[k], pnw[k], rstd_r, 0.0f);
        }
        eng_cbar(s);
    }
}

// stream this wave's records of one phase out of its ring
template <int FMT, bool GEGLU, int YDST>
__device__ void eng_stream(EngCons& s, const EngPhase& P, const u32x4* xs)
{
    constexpr int NR = GEGLU ? 2 : 1;
    constexpr int MAXSTEPS_FP4 = 8;          // K <= 16384
    const int lane = s.lane, cw = s.cw, b = s.b;
    const int cpr = P.cpr, spr = P.spr, upr = cpr + spr;
    const int steps = (cpr + 63) / 64;
    const int ncols = eng_ncols(P.N, b, cw);
    const uint32_t base_units = s.base_piece * 64u;
    uint32_t u = 0;                          // units of this phase already behind us
    for (int t = 0; t < ncols; ++t)
    {
        const int col = b + 256 * (cw + kEngCons * t);
        float acc[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j)
        {
            const int row = j ? P.N + col : col;
            float sc[MAXSTEPS_FP4];
            if constexpr (FMT == FMT_FP4)
            {
                // the record's scale windows come first: copy this lane's scale of every step into registers
                eng_wait_full(s, s.base_piece + (u + (uint32_t)spr - 1u) / 64u);
                const uint32_t mis = (uint32_t)(((size_t)row * (size_t)(P.ngroups * 4)) & 15u);
#pragma unroll
                for (int m = 0; m < MAXSTEPS_FP4; ++m)
                {
                    const int cc = min(64 * m + lane, cpr - 1);
                    const uint32_t off = (((base_units + u) * 16u) + mis + 4u * (uint32_t)(cc >> P.cpg_shift)) & (kRingUnits * 16 - 1);
                    sc[m] = (m < steps) ? *reinterpret_cast<const float*>(s.ring + off) : 0.0f;
                }
            }
            float a = 0.0f;
            for (int m = 0; m < steps; ++m)
            {
                const int c = 64 * m + lane;
                const int cc = min(c, cpr - 1);
                const uint32_t last = u + (uint32_t)spr + (uint32_t)min(64 * m + 63, cpr - 1);
                eng_wait_full(s, s.base_piece + last / 64u);
                const u32x4 w = *reinterpret_cast<const u32x4*>(s.ring + (((base_units + u + (uint32_t)spr + (uint32_t)cc) & (kRingUnits - 1)) * 16u));
                if constexpr (FMT == FMT_FP4)
                {
                    float scm = sc[0];
#pragma unroll
                    for (int mm = 1; mm < MAXSTEPS_FP4; ++mm) scm = (m == mm) ? sc[mm] : scm;
                    a = fmaf(scm, chunk_dot<FMT>(w, xs, c, 0.0f), a);
                }
                else
                    a = chunk_dot<FMT>(w, xs, c, a);
                // the ring units below this point are in registers: hand the space back
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint32_t consumed = u + (uint32_t)spr + (uint32_t)min(64 * (m + 1), cpr);
                if (lane == 0) lds_st(s.ctrl + C_FREE + cw, s.base_piece + consumed / 64u);
            }
            acc[j] = a;
            u += (uint32_t)upr;
        }
        // finish: the arithmetic of matvec_body's finish()
#pragma unroll
        for (int j = 0; j < NR; ++j)
        {
            float v = wave_sum(acc[j]);
            const int row = j ? P.N + col : col;
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
    }
    const uint32_t np = (uint32_t)eng_npieces(P, b, cw);
    s.base_piece += np;
    if (lane == 0) lds_st(s.ctrl + C_FREE + cw, s.base_piece);      // the padding of the last piece counts as consumed
}

// FMT: format of the four layer Linears; HEAD: the last phase is the tied lm_head (format HFMT, fp32 logits)
template <int FMT, bool HEAD, int HFMT>
__global__ __launch_bounds__(kEngThreads) void decode_engine_kernel(const EngParams c)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    volatile uint32_t* ctrl = reinterpret_cast<volatile uint32_t*>(lds + c.ctrl_off);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    if (tid < C_WORDS) ctrl[tid] = 0u;
    __syncthreads();                                     // the only workgroup barrier: before the roles part
    const unsigned long long epoch = *c.epoch;
    if (wave == 0)
    {
        eng_loader(c, lds, ctrl, b, lane);
        return;
    }
    EngCons s;
    s.cw = wave - 1; s.lane = lane; s.b = b;
    s.ring = lds + (size_t)s.cw * (kRingUnits * 16);
    s.ctrl = ctrl; s.error = c.error;
    s.base_piece = 0u; s.full_seen = 0u; s.bar_gen = 0u; s.failed = false;
    u32x4* xa = reinterpret_cast<u32x4*>(lds + c.xa_off);
    u32x4* xb = reinterpret_cast<u32x4*>(lds + c.xb_off);
    // tags: three hand-offs per launch, never 0, unique against what the same words held after the previous launch
    auto tag_of = [&](int k) { return (uint32_t)((epoch * 3ull + (unsigned long long)k) % 65535ull) + 1u; };
    u32x4 r1[kMaxGroups], rk[kMaxGroups];
    // phase 0: a = o_proj(attn)
    eng_stage_x<FMT, 0, X_PLAIN, RES_MEM>(s, c.ph[0], xa, 0u, rk, false);
    s.tag = tag_of(0);
    eng_stream<FMT, false, Y_HANDOFF>(s, c.ph[0], xa);
    // phase 1: r1 = res + rmsnorm(a); h = GeGLU(fc_gate_up(rmsnorm(r1)))
    eng_stage_x<FMT, 2, X_HANDOFF, RES_MEM>(s, c.ph[1], xb, tag_of(0), r1, false);
    s.tag = tag_of(1);
    eng_stream<FMT, true, Y_HANDOFF>(s, c.ph[1], xb);
    // phase 2: d = fc_down(h)
    eng_stage_x<FMT, 0, X_HANDOFF, RES_MEM>(s, c.ph[2], xa, tag_of(1), rk, false);
    s.tag = tag_of(2);
    eng_stream<FMT, false, Y_HANDOFF>(s, c.ph[2], xa);
    // phase 3: r2 = (r1 + rmsnorm(d)) * layer_scalar; y = next(rmsnorm(r2))
    if constexpr (HEAD)
    {
        eng_stage_x<HFMT, 2, X_HANDOFF, RES_REG>(s, c.ph[3], xb, tag_of(2), r1, b == 0 && c.ph[3].res_out != nullptr);
        eng_stream<HFMT, false, Y_F32>(s, c.ph[3], xb);
    }
    else
    {
        eng_stage_x<FMT, 2, X_HANDOFF, RES_REG>(s, c.ph[3], xb, tag_of(2), r1, b == 0 && c.ph[3].res_out != nullptr);
        eng_stream<FMT, false, Y_BF16>(s, c.ph[3], xb);
    }
    // every workgroup read `epoch` before any could pass hand-off 2, which workgroup 0 has now behind it
    if (b == 0 && wave == 1 && lane == 0) *c.epoch = epoch + 1ull;
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

constexpr size_t kEngHeaderBytes = 64;      // [0] unused (chain counter), [8] epoch, [16] error -- the chain's header layout

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
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_engine_kernel<FMT, HEAD, HFMT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
    hipLaunchKernelGGL((decode_engine_kernel<FMT, HEAD, HFMT>), dim3(c.nblocks), dim3(kEngThreads), lds, s, c);
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
    if (D > 7 * kMaxGroups * 512 || K_attn > 16384 || F > 16384) return 0;
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
    return (size_t)kEngCons * kRingUnits * 16 + xa + xb + C_WORDS * 4 <= 160 * 1024 ? 1 : 0;
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
    const size_t xa = (size_t)std::max(eng_x_units(a->fmt, a->K_attn), eng_x_units(a->fmt, F)) * 16;
    const size_t xb = (size_t)std::max(eng_x_units(a->fmt, D), eng_x_units(a->next_fmt, D)) * 16;
    c.xa_off = kEngCons * kRingUnits * 16;
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
