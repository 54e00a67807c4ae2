"""Where a tile of flash_prefill_kernel_s1 spends its cycles (diagnostic library built by flash_stamps.sh): per-segment shader cycles of wave 0 of workgroup 0 (the
heaviest query tile), Gemma local (HS 256, window 1024) and global (HS 512) at T = 2048."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mila_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "tools", "experiments", "_build", "libmila_cdna4_stamps.so")
lib = capi.load()
if os.environ.get("MILA_FLASH_DSPLIT"):
    capi.tune("flash.form", int(os.environ["MILA_FLASH_DSPLIT"]))
T = 2048
for name, NH, NKV, HS, window in (("local", 16, 8, 256, 1024), ("global", 16, 1, 512, 0)):
    q = (torch.randn((T, NH * HS), device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)
    K = (torch.randn((1, NKV, T, HS), device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)
    V = torch.randn((1, NKV, T, HS), device="cuda").to(torch.bfloat16).view(torch.int16)
    Y = torch.empty((T, NH * HS), dtype=torch.int16, device="cuda")
    for _ in range(5):
        capi.call("attn_prefill_bf16", Y, q, K, V, 1, T, NH, NKV, HS, T, 0, window, 1.0)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    assert lib.mila_dbg_flash_stamps(out) == 0
    seg, ntiles, total = list(out[:5]), out[5], out[6]
    names = ["wait + barrier", "staging issue", "QK^T", "softmax", "PV"]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        capi.call("attn_prefill_bf16", Y, q, K, V, 1, T, NH, NKV, HS, T, 0, window, 1.0)
    e1.record()
    torch.cuda.synchronize()
    if os.environ.get("MILA_FLASH_DSPLIT") == "10":
        pp = ["A body", "A slot end", "B body", "B slot end"]
        print(json.dumps({"shape": name, "form": 10, "tiles": ntiles, "group0_per_tile": {n: round(out[8 + i] / max(ntiles, 1)) for i, n in enumerate(pp)},
                          "group1_per_tile": {n: round(out[12 + i] / max(ntiles, 1)) for i, n in enumerate(pp)}}), flush=True)
        continue
    print(json.dumps({"shape": name, "tiles": ntiles, "kernel_us": round(e0.elapsed_time(e1) * 100, 1), "stamped_wave_cycles": out[7], "tile_loop_cycles": total,
                      "cycles_per_tile": round(total / max(ntiles, 1)),
                      "segments_per_tile": {n: round(c / max(ntiles, 1)) for n, c in zip(names, seg)}}), flush=True)
