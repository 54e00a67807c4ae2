"""Per-layer decode table (us per kernel role, algorithmic bytes, TB/s, share of the 8 TB/s peak) from the committed round profiles.
usage: python tools/per_layer_table.py r02 > profiles/r02_decode_per_layer.md"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
D, F, V = 3840, 15360, 262144
QKV_L, QKV_G, KA_L, KA_G = (16 + 2 * 8) * 256, (16 + 1) * 512, 16 * 256, 16 * 512       # Gemma-4 12B: packed qkv widths, attention widths
BPW = {"bf16": 2.0, "fp8": 1.0, "fp4": 0.5}
FMT = {"bf16": 0, "fp8": 1, "fp4": 2}


def rows(policy):
    out = []
    for line in open(os.path.join(ROOT, "profiles", "%s_bench_%s_kernels.md" % (tag, policy))):
        c = [x.strip() for x in line.split("|")]
        if len(c) > 10 and c[1].startswith("`"):
            out.append(dict(name=c[1].strip("`"), grid=c[2], calls=int(c[5]), avg=float(c[6]), mn=float(c[7]), mx=float(c[8])))
    return out


def scale_bytes(policy, n, k):
    return {"bf16": 0, "fp8": 4 * n, "fp4": 4 * n * k // 128}[policy]


print("# round %s: decode, per layer and per kernel role (rocprofv3 kernel trace of the graph replay, profiles/%s_bench_<policy>_kernels.md)\n" % (tag[1:], tag))
print("48 layers = 40 sliding-window (HS 256, 8 KV heads, band 1024) + 8 global (HS 512, 1 KV head, band = context 2048+); the matvec rows average both kinds where the")
print("template instance is shared.  bytes = weights + scales (+ KV band for attention); TB/s = bytes / avg us; share = TB/s / 8.\n")
for policy in ("bf16", "fp8", "fp4"):
    r = rows(policy)
    f = FMT[policy]

    def pick(pattern, grid=None):
        for x in r:
            if re.search(pattern, x["name"]) and (grid is None or x["grid"].startswith(grid)):
                return x
        return None
    b = BPW[policy]
    qkv_bytes = (40 * (QKV_L * D * b + scale_bytes(policy, QKV_L, D)) + 8 * (QKV_G * D * b + scale_bytes(policy, QKV_G, D))) / 48
    o_bytes = (40 * (D * KA_L * b + scale_bytes(policy, D, KA_L)) + 8 * (D * KA_G * b + scale_bytes(policy, D, KA_G))) / 48
    roles = [
        ("qkv_proj (+ tail of the previous block + input_norm)", pick(r"matvec_kernel<%d, \d+, \d+, 2, false, false, 1, 0>" % f), qkv_bytes),
        ("attention, sliding-window layers (q/k/v norm + RoPE + KV append + flash-decode splits)", pick(r"attn_decode_kernel<256"), 2 * 1024 * 8 * 256 * 2),
        ("attention, global layers", pick(r"attn_decode_kernel<512"), 2 * 2112 * 1 * 512 * 2),
        ("combine of the splits", pick(r"attn_combine_kernel", "16,1,4"), None),
        ("o_proj", pick(r"matvec_kernel<%d, \d+, \d+, 0, false, false, 1, 0>" % f), o_bytes),
        ("post_attn_norm + residual + pre_ffn_norm + fc_gate_up + GeGLU", pick(r"matvec_kernel<%d, \d+, \d+, 2, true, false" % f), 2 * F * D * b + scale_bytes(policy, 2 * F, D)),
        ("fc_down", pick(r"matvec_kernel<%d, \d+, \d+, 0, false, false, 2, 0>" % f), D * F * b + scale_bytes(policy, D, F)),
    ]
    tf = 0 if policy == "bf16" else 1
    head = pick(r"matvec_kernel<%d, \d+, \d+, 2, false, true" % tf)
    print("## %s\n" % policy)
    print("| role | kernel instance | calls | avg us | min | max | MB | TB/s | share of 8 TB/s |\n|---|---|---|---|---|---|---|---|---|")
    total = 0.0
    for name, x, by in roles:
        if x is None:
            continue
        tbs = "%.2f" % (by / x["avg"] / 1e6) if by else "–"
        share = "%.0f %%" % (100 * by / x["avg"] / 1e6 / 8) if by else "–"
        print("| %s | `%s` %s | %d | %.2f | %.2f | %.2f | %s | %s | %s |" % (name, re.sub(r"\(.*", "", x["name"]).replace("void mila::", ""), x["grid"], x["calls"], x["avg"], x["mn"], x["mx"],
                                                                  "%.1f" % (by / 1e6) if by else "–", tbs, share))
    la = {n: x for n, x, _ in roles if x}
    loc = sum(x["avg"] for n, x in la.items() if "global" not in n)
    glo = sum(x["avg"] for n, x in la.items() if "sliding" not in n)
    print("\nsum of kernel durations: sliding-window layer %.1f us, global layer %.1f us; x 40 + x 8 = %.2f ms per token" % (loc, glo, (40 * loc + 8 * glo) / 1e3), end="")
    if head:
        hb = V * D * (2.0 if policy == "bf16" else 1.0) + (0 if policy == "bf16" else 4 * V)
        print("; final norm + lm_head %.1f us (%.2f TB/s)" % (head["avg"], hb / head["avg"] / 1e6), end="")
    print(".")
    pmc = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (tag, policy))
    if os.path.exists(pmc):
        for k in json.load(open(pmc))["kernels"]:
            if re.search(r"matvec_kernel<%d, \d+, \d+, 2, true, false" % f, k["kernel"]):
                alg = 2 * F * D * b + scale_bytes(policy, 2 * F, D)
                print("PMC traffic of the dominant kernel (2 x FETCH_SIZE + WRITE_SIZE, separate passes): %.1f MB per launch vs %.1f MB algorithmic = %.3fx." % (
                    k["hbm_bytes_per_launch"] / 1e6, alg / 1e6, k["hbm_bytes_per_launch"] / alg))
    under = os.path.join(ROOT, "profiles", "%s_bench_%s_under_rocprof.json" % (tag, policy))
    try:
        d = json.loads(open(under).read().strip().splitlines()[-1])
        p = d["policies"][policy]
        print("bench line of the same run (under the profiler): %.1f tok/s, %.3f ms per token, whole-token roofline %.3f.\n" % (p["tok_s"], p["ms_per_step"], p["token_roofline_frac"]))
    except Exception:
        print()
