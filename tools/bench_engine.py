"""Engine vs the four launches it replaces, Gemma-4 12B geometry, cold weights (L distinct layers' matrices per replay, far beyond the
256 MiB Infinity Cache), each form captured in one hipGraph and replayed.  usage: python tools/bench_engine.py [bf16|fp8|fp4 ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from mila_amd import capi  # noqa: E402

FMT = {"bf16": 0, "fp8": 1, "fp4": 2}
D, F, KA, NN = 3840, 15360, 4096, 8192
LAYERS = int(os.environ.get("LAYERS", "8"))


def u16(*s):
    return torch.empty(s, dtype=torch.int16, device="cuda")


def weights(N, K, fmt, seed):
    W = u16(N, K)
    capi.call("fill_uniform_bf16", W, C.c_int64(N * K), C.c_uint64(seed), float(K) ** -0.5, 0.0)
    if fmt == 0:
        return W, None
    if fmt == 1:
        q, s = torch.empty((N, K), dtype=torch.uint8, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")
        capi.call("quantize_fp8_per_channel", q, s, W, N, K)
        return q, s
    q, s = torch.empty((N, K // 2), dtype=torch.uint8, device="cuda"), torch.empty((N, K // 128), dtype=torch.float32, device="cuda")
    capi.call("quantize_fp4_per_group", q, s, W, N, K, 128)
    return q, s


def main():
    lib = capi.load()
    for name in sys.argv[1:] or ["bf16", "fp8", "fp4"]:
        fmt = FMT[name]
        layers = []
        for l in range(LAYERS):
            layers.append([weights(D, KA, fmt, 10 * l + 1), weights(2 * F, D, fmt, 10 * l + 2), weights(D, F, fmt, 10 * l + 3), weights(NN, D, fmt, 10 * l + 4)])
        nws = [u16(D) for _ in range(4)]
        for t in nws:
            capi.call("fill_uniform_bf16", t, C.c_int64(D), C.c_uint64(99), 0.1, 1.0)
        attn, res = u16(KA), u16(D)
        capi.call("fill_uniform_bf16", attn, C.c_int64(KA), C.c_uint64(5), 1.0, 0.0)
        capi.call("fill_uniform_bf16", res, C.c_int64(D), C.c_uint64(6), 1.0, 0.0)
        a0, h0, d0, r1, r2, y0, y1, r2c = u16(D), u16(F), u16(D), u16(D), u16(D), u16(NN), u16(NN), u16(D)
        nbytes = lib.mila_cdna4_decode_engine_scratch_bytes(D, F)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        capi.call("decode_engine_init", scratch, C.c_size_t(nbytes))
        bytes_layer = sum(w.numel() * w.element_size() + (0 if s is None else s.numel() * 4) for w, s in layers[0])

        def ptr(t):
            return 0 if t is None else t.data_ptr()

        def launches(stream):
            for (Wo, so), (Wg, sg), (Wd, sd), (Wn, sn) in layers:
                if fmt == 0:
                    capi.call("matvec_bf16", a0, attn, Wo, None, KA, D)
                elif fmt == 1:
                    capi.call("matvec_bf16_qfp8", a0, attn, Wo, so, None, KA, D)
                else:
                    capi.call("matvec_bf16_qfp4", a0, attn, Wo, so, None, KA, D, 128)
                fa = capi.fused_matvec_args(y=h0.data_ptr(), x=a0.data_ptr(), W=Wg.data_ptr(), scales=ptr(sg), norm_w=nws[1].data_ptr(), post_w=nws[0].data_ptr(),
                                            res=res.data_ptr(), res_out=r1.data_ptr(), post_scale=1.0, eps=1e-6, fmt=fmt, K=D, N=F, group=128, geglu=1)
                capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(fa), stream))
                if fmt == 0:
                    capi.call("matvec_bf16", d0, h0, Wd, None, F, D)
                elif fmt == 1:
                    capi.call("matvec_bf16_qfp8", d0, h0, Wd, sd, None, F, D)
                else:
                    capi.call("matvec_bf16_qfp4", d0, h0, Wd, sd, None, F, D, 128)
                fb = capi.fused_matvec_args(y=y0.data_ptr(), x=d0.data_ptr(), W=Wn.data_ptr(), scales=ptr(sn), norm_w=nws[3].data_ptr(), post_w=nws[2].data_ptr(),
                                            res=r1.data_ptr(), res_out=r2.data_ptr(), post_scale=0.75, eps=1e-6, fmt=fmt, K=D, N=NN, group=128, geglu=0)
                capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(fb), stream))

        def engine(stream):
            for (Wo, so), (Wg, sg), (Wd, sd), (Wn, sn) in layers:
                ca = capi.decode_chain_args(attn=attn.data_ptr(), res=res.data_ptr(), res_out=r2c.data_ptr(), y=y1.data_ptr(), W_o=Wo.data_ptr(), s_o=ptr(so),
                                            W_gate_up=Wg.data_ptr(), s_gate_up=ptr(sg), W_down=Wd.data_ptr(), s_down=ptr(sd), W_next=Wn.data_ptr(), s_next=ptr(sn),
                                            post_attn_w=nws[0].data_ptr(), pre_ffn_w=nws[1].data_ptr(), post_ffn_w=nws[2].data_ptr(), next_norm_w=nws[3].data_ptr(),
                                            layer_scalar=0.75, eps=1e-6, fmt=fmt, group=128, next_fmt=fmt, next_group=128, f32_out=0, D=D, F=F, K_attn=KA, N_next=NN,
                                            scratch=scratch.data_ptr(), scratch_bytes=nbytes)
                capi.check(lib.mila_cdna4_decode_engine(C.byref(ca), stream))

        out = {}
        for label, fn in (("launches", launches), ("engine", engine)):
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                st = C.c_void_p(side.cuda_stream)
                fn(st)                                   # warm-up outside capture (function attributes, lazy module load)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    fn(C.c_void_p(torch.cuda.current_stream().cuda_stream))
                for _ in range(3):
                    g.replay()
                side.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 20
                e0.record(side)
                for _ in range(reps):
                    g.replay()
                e1.record(side)
                side.synchronize()
                out[label] = e0.elapsed_time(e1) * 1e3 / (reps * LAYERS)
        err = C.c_int32(-1)
        capi.check(lib.mila_cdna4_decode_engine_status(C.c_void_p(scratch.data_ptr()), C.byref(err), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        same = bool(torch.equal(y0, y1)) and bool(torch.equal(r2, r2c))
        print("%s: %.1f MB/layer  launches %.2f us/layer (%.2f TB/s)  engine %.2f us/layer (%.2f TB/s)  ratio %.3f  bit-identical %s  engine status %d"
              % (name, bytes_layer / 1e6, out["launches"], bytes_layer / out["launches"] / 1e6, out["engine"], bytes_layer / out["engine"] / 1e6,
                 out["engine"] / out["launches"], same, err.value), flush=True)
        del layers


if __name__ == "__main__":
    main()
