"""Time the flash prefill kernel on the Gemma-4 12B attention shapes (T = 2048).  MILA_FLASH_DSPLIT selects a kernel form (tuning variable flash.form);
where the cycles of a tile go: tools/experiments/flash_stamps.sh + flash_stamps.py (a diagnostic build with in-kernel stamps)."""
import json, os, sys
os.environ.setdefault("MILA_CDNA4_TUNING", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi
T = 2048
if os.environ.get("MILA_FLASH_LIB"):          # an experiment build of the library (tools/experiments/flash_stamps.sh with FLASH_DEFS=...)
    capi.LIB_PATH = os.environ["MILA_FLASH_LIB"]
if os.environ.get("MILA_FLASH_DSPLIT"):
    capi.tune("flash.form", int(os.environ["MILA_FLASH_DSPLIT"]))      # csrc/internal.h: 8 default; 9 lockstep 8-wave workgroups at HS 256; 10 ping-pong; 2 / 1 the older forms
for name, NH, NKV, HS, window in (("local", 16, 8, 256, 1024), ("global", 16, 1, 512, 0)):
    q = (torch.randn((T, NH * HS), device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)
    K = (torch.randn((1, NKV, T, HS), device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)
    V = torch.randn((1, NKV, T, HS), device="cuda").to(torch.bfloat16).view(torch.int16)
    Y = torch.empty((T, NH * HS), dtype=torch.int16, device="cuda")
    call = lambda: capi.call("attn_prefill_bf16", Y, q, K, V, 1, T, NH, NKV, HS, T, 0, window, 1.0)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    keys = sum(min(t + 1, window) if window else t + 1 for t in range(T))
    print(json.dumps({"shape": name, "form": os.environ.get("MILA_FLASH_DSPLIT", "default"), "us": round(us, 1), "TFLOPs": round(4.0 * NH * HS * keys / us / 1e6, 1)}), flush=True)
