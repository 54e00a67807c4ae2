// Test / tuning hooks exported by the shared object but NOT part of the drop-in ABI
// (include/mila_cdna4.h).  Used by tests/ and the micro-benchmarks only; nothing under mila_amd/host includes this file.
// The mila_cdna4_tune_* hooks change process-wide launch heuristics, so they are INERT (MILA_E_UNSUPPORTED, nothing stored)
// unless the process set MILA_CDNA4_TUNING=1 before the library was loaded: a product process cannot reach that state.
#pragma once
#include "../../include/mila_cdna4.h"

#ifdef __cplusplus
extern "C" {
#endif
/* override the matvec launch heuristics (0 = default): rows per wave, chunk positions in flight,
 * maximum workgroups */
MILA_API int mila_cdna4_tune_matvec(int R, int U, int max_blocks);
/* 1 = always use the 128 x 128 register-staged GEMM (A/B against the 256 x 256 direct-to-LDS kernel) */
MILA_API int mila_cdna4_tune_gemm(int force_128_tile);
/* schedule of the LDS-DMA GEMMs: 0 = all eight waves in lockstep; 1 = staggered (ping-pong), four phases per K-tile; 2 = 1, preferring the 256 x 128 ring;
 * 3 = staggered, two phases per K-tile; 4 = 3 + the fp8 shapes on the 256 x 256 kernel wherever it applies; 5 (default) = 4 with a static priority for waves 4-7
 * and persistent tiles; 6 = 5 with one workgroup per tile.  All give the same bits. */
MILA_API int mila_cdna4_tune_gemm_schedule(int pingpong);
/* 1 = the fp8 x fp8 GEMM entry points run EVERY row on the masked 128-row kernel (gemm_fp8_tail.hip), 0 (default) = LDS-DMA kernels on the leading
 * multiple of 256 rows.  Same bits either way: the test of that statement. */
MILA_API int mila_cdna4_tune_gemm_fp8_tail_only(int on);
/* positions of the live band one flash-decode split covers (default 64; 0 restores it): fewer, longer splits = smaller partial sets */
MILA_API int mila_cdna4_tune_attn_split(int positions_per_split);
/* flash-prefill form: 8 (default) = LDS-DMA kernels (HS 512: 8-wave workgroups, four heads x two d-halves; HS 256: double-buffered 4-wave workgroups);
 * 9 = 8 with 8-wave workgroups at HS 256 too; 2 = HS 512 as 4-wave d-split workgroups; 1 = the register-staged kernels.  All give the same bits. */
MILA_API int mila_cdna4_tune_flash_dsplit(int ds);
/* engine diagnostics: the next decode_engine launches write wall-clock stamps (100 MHz) of the first 8 workgroups' 8 waves, 16 slots each */
MILA_API int mila_cdna4_decode_engine_debug(unsigned long long* buf);
/* decode all 256 byte values with the hardware converts used by the kernels:
 * out_fp8[256] floats; out_fp4[512] floats (byte b -> [2b] low nibble, [2b+1] high nibble),
 * each for the 4 byte positions of a dword: out_fp4 has 4*512 floats, out_fp8 4*256. */
MILA_API int mila_cdna4_selftest_decode(float* out_fp8, float* out_fp4, mila_stream_t stream);
/* out[0..63] = wave_sum(in[lane]), out[64..127] = wave_max, out[128..191] = the ds_bpermute butterfly */
MILA_API int mila_cdna4_selftest_wave_reduce(float* out, const float* in, mila_stream_t stream);
MILA_API int mila_cdna4_selftest_mfma_fp8(float* C, const uint8_t* A, const uint8_t* B, int mode, mila_stream_t stream);
/* streaming-copy ceiling: dst <- src with 16-byte accesses; used to report a measured HBM roof */
MILA_API int mila_cdna4_stream_copy(void* dst, const void* src, size_t bytes, mila_stream_t stream);
MILA_API int mila_cdna4_stream_read(float* sink, const void* src, size_t bytes, mila_stream_t stream);
#ifdef __cplusplus
}
#endif
