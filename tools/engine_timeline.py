"""Diagnostic: one engine launch on the Gemma geometry with in-kernel wall-clock stamps (csrc/engine.hip eng_stamp), printed as a
per-wave timeline in microseconds.  usage: python tools/engine_timeline.py [bf16|fp8|fp4]"""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")
import ctypes as C
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mila_amd import capi  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_engine import D, F, KA, NN, FMT, u16, weights  # noqa: E402

lib = capi.load()
name = sys.argv[1] if len(sys.argv) > 1 else "bf16"
fmt = FMT[name]
sets = [[weights(D, KA, fmt, 10 * l + 1), weights(2 * F, D, fmt, 10 * l + 2), weights(D, F, fmt, 10 * l + 3), weights(NN, D, fmt, 10 * l + 4)] for l in range(3)]
nws = [u16(D) for _ in range(4)]
for t in nws:
    capi.call("fill_uniform_bf16", t, C.c_int64(D), C.c_uint64(99), 0.1, 1.0)
attn, res, y1, r2c = u16(KA), u16(D), u16(NN), u16(D)
capi.call("fill_uniform_bf16", attn, C.c_int64(KA), C.c_uint64(5), 1.0, 0.0)
capi.call("fill_uniform_bf16", res, C.c_int64(D), C.c_uint64(6), 1.0, 0.0)
nbytes = lib.mila_cdna4_decode_engine_scratch_bytes(D, F)
scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
capi.call("decode_engine_init", scratch, C.c_size_t(nbytes))
dbg = torch.zeros(256 * 16 * 16, dtype=torch.int64, device="cuda")
capi.check(lib.mila_cdna4_decode_engine_debug(C.c_void_p(dbg.data_ptr())))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return 0 if t is None else t.data_ptr()


for l in range(3):        # the last launch is the one reported (weights cold, code warm)
    (Wo, so), (Wg, sg), (Wd, sd), (Wn, sn) = sets[l]
    ca = capi.decode_chain_args(attn=attn.data_ptr(), res=res.data_ptr(), res_out=r2c.data_ptr(), y=y1.data_ptr(), W_o=Wo.data_ptr(), s_o=ptr(so),
                                W_gate_up=Wg.data_ptr(), s_gate_up=ptr(sg), W_down=Wd.data_ptr(), s_down=ptr(sd), W_next=Wn.data_ptr(), s_next=ptr(sn),
                                post_attn_w=nws[0].data_ptr(), pre_ffn_w=nws[1].data_ptr(), post_ffn_w=nws[2].data_ptr(), next_norm_w=nws[3].data_ptr(),
                                layer_scalar=0.75, eps=1e-6, fmt=fmt, group=128, next_fmt=fmt, next_group=128, f32_out=0, D=D, F=F, K_attn=KA, N_next=NN,
                                scratch=scratch.data_ptr(), scratch_bytes=nbytes)
    capi.check(lib.mila_cdna4_decode_engine(C.byref(ca), st))
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(256, 16, 16).astype(np.float64)
d[d == 0] = np.nan
t0 = np.nanmin(d[:, :, 0])
names_c = ["start", "staged0", "streamed0", "staged1", "streamed1", "staged2", "streamed2", "staged3", "streamed3"]
us = (d[:, :, :9] - t0) / 100.0
for b in (0, 1, 7, 100, 255):
    for w in (0, 7, 8, 14):
        print("block %3d wave %d : " % (b, w) + "  ".join("%s %.2f" % (n, us[b, w, i]) for i, n in enumerate(names_c)))
print("over all 256 workgroups x 8 waves (min / median / max us):")
for i, n in enumerate(names_c):
    v = us[:, :, i].reshape(-1)
    print("  %-10s %8.2f %8.2f %8.2f   by wave (max over workgroups): %s" % (n, np.nanmin(v), np.nanmedian(v), np.nanmax(v), " ".join("%.1f" % x for x in np.nanmax(us[:, :, i], axis=0))))
# slots 9..12: cycles spent waiting for pieces, cumulative after each phase; 13/14: the cycle counter at start / end (calibration)
raw = dbg.cpu().numpy().reshape(256, 16, 16).astype(np.float64)[:, :8, :]
cyc_per_us = np.nanmedian((raw[:, :, 14] - raw[:, :, 13]) / (us[:, :8, 8] - us[:, :8, 0]))
print("cycle counter: %.0f cycles per us" % cyc_per_us)
total = us[:, :8, 8] - us[:, :8, 0]
print("  per wave, median over workgroups: launch %.1f us; waiting for weight pieces %.1f, for LDS reads %.1f, requesting (release + top_up) %.1f, finishing rows %.1f" % (
    np.nanmedian(total), np.nanmedian(raw[:, :, 9]) / cyc_per_us, np.nanmedian(raw[:, :, 10]) / cyc_per_us, np.nanmedian(raw[:, :, 11]) / cyc_per_us, np.nanmedian(raw[:, :, 12]) / cyc_per_us))
