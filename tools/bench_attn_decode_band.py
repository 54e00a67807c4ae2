"""the matrix-core decode of the global layers is chosen from the cache CAPACITY (the graph needs a static choice), not from the live band: a 32K-capacity
cache decoding at a short position runs attn_decode_mfma_kernel with 256 splits of a few keys each.  This tool times the global-layer attention entry (16 heads on one KV
head, HS 512, capacity 32768) at live positions from 512 to 32K with the matrix-core decode on (default) and off (attn.mfma_decode = 0), replayed from a graph.
    python tools/bench_attn_decode_band.py  ->  one JSON line per (position, form)"""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

NH, NKV, HS, CAP = 16, 1, 512, 32768
lib = capi.load()
K = (torch.randn((1, NKV, CAP, HS), device="cuda") * 0.3).to(torch.bfloat16).view(torch.int16)
V = torch.randn((1, NKV, CAP, HS), device="cuda").to(torch.bfloat16).view(torch.int16)
q = (torch.randn((1, NH * HS), device="cuda") * 0.3).to(torch.bfloat16).view(torch.int16)
Y = torch.empty((1, NH * HS), dtype=torch.int16, device="cuda")
nb = lib.mila_cdna4_attn_decode_scratch_bytes(1, NH, HS)
scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
filler = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")      # 512 MiB written between replays: K / V come from HBM, not from the Infinity Cache


def time_at(length, reps=30):
    call = lambda: capi.call("attn_decode_bf16", Y, q, K, V, scratch, C.c_size_t(nb), 1, NH, NKV, HS, CAP, length, 0, 1.0)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(reps):
        filler.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
    return total / reps * 1e3


for pos in (512, 1024, 2048, 4096, 8192, 16384, 32767):
    row = {"position": pos, "capacity": CAP}
    for name, on in (("mfma_us", 1), ("scalar_us", 0)):
        capi.tune("attn.mfma_decode", on)
        row[name] = round(time_at(pos + 1), 2)
    capi.tune_reset()
    print(json.dumps(row), flush=True)
