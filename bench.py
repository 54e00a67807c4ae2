#!/usr/bin/env python3
"""Headline benchmark: Gemma-4 12B decode tok/s (+ prefill TFLOP/s) on MI355X for the three weight
policies of BASELINE.json (bf16 / PerChannelFp8 / PerGroupFp4), with the roofline fraction of the
dominant kernel and the restated reference CPU backend timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--policies bf16,fp8,fp4] [--no-prefill]

A "step" is one decode token of the whole model (embedding -> 48 blocks -> final norm -> lm_head, fp32
logits) at context 2048 with every weight resident in HBM, replayed from one captured hipGraph.  The
path does not shard: with --gpus N every rank runs an independent replica on its
own GPU; there is no data-path collective.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0       # dense e4m3 (block-scaled MFMA): the fp4 policy's W4A8 prefill runs its Linears there
CONTEXT = 2048                # BASELINE.json configs[2..4]: B=1, T=2048


def _oracle(omp_threads=None):
    """the CPU oracle as a ctypes library: the serial build, or the OpenMP build (the reference's own `#pragma omp` placements,
    oracle/mila_oracle.c) pinned to `omp_threads` threads.  bench.py's cpu_baseline leg is one of the three places allowed to."""
    import ctypes as C
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    name = "libmila_oracle_omp.so" if omp_threads else "libmila_oracle.so"
    path = os.path.join(odir, "_build", name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", odir, "-s", "_build/" + name])
    if omp_threads:
        os.environ["OMP_NUM_THREADS"] = str(omp_threads)       # read by libgomp when it is first loaded
    return C.CDLL(path)


def usable_cores():
    """host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box exposes all of the
    host's cores to os.cpu_count() but grants a share of them; oversubscribing OpenMP there is slower than one thread)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return min(n, int(os.environ.get("MILA_BENCH_CPU_CORES", "16")))    # one GPU's CPU share on the pool's boxes is 16 cores


def cpu_baseline(cfg):
    """Mila's CPU backend, restated (oracle/mila_oracle.c, kind "port"), timed on this box's host cores -- SURVEY.md section 8d:
      (i)  single thread, the reference default (MILA_ENABLE_OPENMP is OFF, CMakeLists.txt:78);
      (ii) the reference's own `#pragma omp` placements on the host cores this process may use (usable_cores(); `host_cores_visible`
          is os.cpu_count()).
    Two workloads, each a BOUNDED sample:
      * one Gemma-4 12B decode token's Linear work (CpuLinearOp::forwardNaive, the batch-1 path; the reference has no CPU
        RMSNorm / RoPE / GQA / GeGLU ops -- < 1 % of the work): a few passes over a local and a global layer's four Linear shapes +
        1/8 of the lm_head rows, extrapolated by weight count.  The reference parallelises forwardNaive over BATCH ROWS
        (CpuLinearOp.ixx:389), so at batch 1 leg (ii) runs on one core as well: measured, not assumed;
      * BASELINE config 1 end to end: GPT-2 124M FP32 forward, B = 1, T = 64 (the one configuration the CPU backend covers),
        15.9 GFLOP, on random parameters."""
    import ctypes as C
    import numpy as np
    rng = np.random.default_rng(0)
    D, H, V = cfg["embedding_dim"], cfg["hidden_dim"], cfg["vocab_size"]
    f32p = C.POINTER(C.c_float)
    ncores = usable_cores()

    def shapes(g):
        hd = cfg["global_head_dim"] if g else cfg["head_dim"]
        nkv = cfg["num_global_kv_heads"] if g else cfg["num_kv_heads"]
        qw = cfg["num_heads"] * hd
        return [(D, qw + (1 if g else 2) * nkv * hd), (qw, D), (D, 2 * H), (H, D)]

    REPS_LOC, REPS_GLB, HEAD_FRAC = 8, 2, 1.0 / 8
    n_glb = sum(1 for i in range(cfg["num_layers"]) if (i + 1) % cfg["sliding_window_pattern"] == 0)
    n_loc = cfg["num_layers"] - n_glb
    # the sample's operands, shared by both legs: 30-470 MB matrices, far beyond the CPU caches -- every pass streams from DRAM
    mats = {}
    for g in (False, True):
        for K, N in shapes(g):
            mats.setdefault((K, N), (rng.standard_normal((N, K), dtype=np.float32), rng.standard_normal((1, K), dtype=np.float32)))
    nh = max(1, int(V * HEAD_FRAC))
    head = (rng.standard_normal((nh, D), dtype=np.float32), rng.standard_normal((1, D), dtype=np.float32))

    def linear(lib, W, x):
        N, K = W.shape
        y = np.empty((1, N), dtype=np.float32)
        t0 = time.perf_counter()
        lib.orc_cpu_linear(y.ctypes.data_as(f32p), x.ctypes.data_as(f32p), W.ctypes.data_as(f32p), None, C.c_int64(1), C.c_int64(K), C.c_int64(N))
        return time.perf_counter() - t0

    def token_leg(lib):
        t_loc = sum(linear(lib, *mats[s]) for _ in range(REPS_LOC) for s in shapes(False))
        t_glb = sum(linear(lib, *mats[s]) for _ in range(REPS_GLB) for s in shapes(True))
        t_head = linear(lib, *head)
        macs = REPS_LOC * sum(k * n for k, n in shapes(False)) + REPS_GLB * sum(k * n for k, n in shapes(True)) + nh * D
        token_s = t_loc / REPS_LOC * n_loc + t_glb / REPS_GLB * n_glb + t_head / HEAD_FRAC
        return {"tok_s": round(1.0 / token_s, 5), "sample_s": round(t_loc + t_glb + t_head, 2), "GFLOPs": round(2 * macs / (t_loc + t_glb + t_head) / 1e9, 3)}

    # BASELINE config 1: GPT-2 124M, B = 1, T = 64 (oracle orc_cpu_gpt2_forward = GptTransformer::forward on the CPU backend)
    Cg, Lg, NHg, Vg, Tg, maxT = 768, 12, 12, 50257, 64, 1024
    params = [rng.standard_normal((Vg, Cg), dtype=np.float32) * np.float32(0.02), rng.standard_normal((maxT, Cg), dtype=np.float32) * np.float32(0.01)]
    for _ in range(Lg):
        params += [np.ones(Cg, np.float32), np.zeros(Cg, np.float32), rng.standard_normal((3 * Cg, Cg), dtype=np.float32) * np.float32(0.02), np.zeros(3 * Cg, np.float32),
                   rng.standard_normal((Cg, Cg), dtype=np.float32) * np.float32(0.02), np.zeros(Cg, np.float32), np.ones(Cg, np.float32), np.zeros(Cg, np.float32),
                   rng.standard_normal((4 * Cg, Cg), dtype=np.float32) * np.float32(0.02), np.zeros(4 * Cg, np.float32),
                   rng.standard_normal((Cg, 4 * Cg), dtype=np.float32) * np.float32(0.02), np.zeros(Cg, np.float32)]
    params += [np.ones(Cg, np.float32), np.zeros(Cg, np.float32), params[0]]
    tokens = (np.arange(Tg, dtype=np.int32) * 7919 + 13) % Vg
    arr = (f32p * len(params))(*[p.ctypes.data_as(f32p) for p in params])
    logits = np.empty((1, Tg, Vg), dtype=np.float32)
    gpt_flop = 2.0 * Tg * (Lg * 12 * Cg * Cg + Vg * Cg) + Lg * 4.0 * NHg * (Cg // NHg) * Tg * Tg

    def gpt_leg(lib):
        t0 = time.perf_counter()
        lib.orc_cpu_gpt2_forward(logits.ctypes.data_as(f32p), tokens.ctypes.data_as(C.POINTER(C.c_int32)), arr, 1, Tg, Cg, Lg, NHg, Vg, maxT)
        dt = time.perf_counter() - t0
        return {"forward_ms": round(dt * 1e3, 1), "tok_s": round(Tg / dt, 2), "GFLOPs": round(gpt_flop / dt / 1e9, 3)}

    serial = _oracle()
    one_tok, one_gpt = token_leg(serial), gpt_leg(serial)
    omp = _oracle(omp_threads=ncores)
    all_tok, all_gpt = token_leg(omp), gpt_leg(omp)
    return {"value": one_tok["tok_s"], "unit": "tok/s", "cores": 1, "kind": "port",
            "sample": "restated CpuLinearOp::forwardNaive (FP32, long double accumulation: the reference's batch-1 path) on %d passes over a local + %d over a "
                      "global layer's four Linear shapes + 1/%d of the lm_head rows (%.1f s on one core), extrapolated by weight count to %d layers + head; "
                      "`value` is the single-thread leg = the reference's default build (MILA_ENABLE_OPENMP OFF)" % (REPS_LOC, REPS_GLB, round(1 / HEAD_FRAC), one_tok["sample_s"], cfg["num_layers"]),
            "host_cores": ncores, "host_cores_visible": os.cpu_count(),
            "legs": {"single_thread": dict(one_tok, cores=1),
                     "all_cores_reference_omp_placements": dict(all_tok, cores=ncores, note="forwardNaive parallelises over batch rows (CpuLinearOp.ixx:389): one row at decode, so one busy core")},
            "config1_gpt2_124M_fp32_B1_T64": {"GFLOP": round(gpt_flop / 1e9, 2), "single_thread": dict(one_gpt, cores=1), "all_cores_reference_omp_placements": dict(all_gpt, cores=ncores)}}


def measured_traffic(policy):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/, produced by
    tools/summarize_pmc.py with the gfx950 FETCH_SIZE correction); None when no pass exists for this policy."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic_%s.json" % policy)))      # the latest round's pass
    if not found:
        return None, None
    path = found[-1]
    fmt = {"bf16": 0, "fp8": 1, "fp4": 2}[policy]
    import re
    want = re.compile(r"matvec_kernel<%d, \d+, \d+, 2, true, false" % fmt)      # <FMT, R, U, PRO=2, GEGLU, !F32OUT, ...>: fc_gate_up
    for k in json.load(open(path))["kernels"]:
        if want.search(k["kernel"]):
            return k["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


PREFILL_UNTIMED, PREFILL_TIMED = 2, 7


def timed_prefill(m):
    """SURVEY.md section 8(d) protocol: >= 2 untimed passes, then the median of >= 7 timed ones (HIP events on the model's stream around one T = 2048 prefill each;
    host.Gemma.time_prefill runs one more untimed pass in front of every timed one).  Returns (median ms, min ms, spread = max - min)."""
    for _ in range((PREFILL_UNTIMED + 1) // 2):
        m.time_prefill(CONTEXT, 1)
    xs = [m.time_prefill(CONTEXT, 1) for _ in range(PREFILL_TIMED)]
    return _median(xs), min(xs), max(xs) - min(xs)


def launch_boundary_us(capi):
    """what one more graph node costs when it does no work: a chain of 64 one-thread launches (advance_position) replayed from a hipGraph, microseconds per launch
    (tools/bench_fixed_cost.py's first case, measured in this process) -- the `fixed_us_per_launch` of the decode design floor"""
    import torch
    pos = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(3):
        capi.call("advance_position", pos)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(64):
            capi.call("advance_position", pos)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (20 * 64)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_replicas(n, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N child processes, one per GPU, BEFORE anything in this process
    touches a GPU, with the env torch.distributed.run would give them; rank 0's JSON line is this command's output.  The path does
    not shard: the children are independent replicas that only share the timing barrier (gloo)."""
    import subprocess
    port = str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # poll every child: as soon as one exits non-zero the rest are ended (fresh processes of ours, so killing them is safe) instead of
    # rank 0 sitting in the gloo barrier until the process-group timeout; rank 0's stdout is drained by a thread so its pipe never fills
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().splitlines()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0):
                failed = p.returncode
                break
        else:
            time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    reader.join(timeout=10)
    rcs = [p.returncode for p in procs]
    lines = [ln for ln in out0 if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    if failed is not None or any(rcs):
        return max(1, max(abs(rc) for rc in rcs))
    if not lines:
        print("bench.py: every replica exited 0 but rank 0 printed no JSON line", file=sys.stderr)
        return 1
    return 0


def stub_result(steps, warmup, rank):
    """--stub: no GPU work at all -- a sleep stands where the timed decode loop is, so that the launcher / barrier / aggregation
    plumbing can be exercised on a CPU box (tests/test_replicas_cpu.py).  The line it prints says so in `data`."""
    for _ in range(warmup):
        time.sleep(0.001)
    t0 = time.perf_counter()
    for _ in range(steps):
        time.sleep(0.002 + 0.001 * rank)        # rank 1 is the slow one: the aggregate must use the MAX over ranks
    return (time.perf_counter() - t0) * 1e3 / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--policies", default="bf16,fp8,fp4")
    ap.add_argument("--mode", default="graph", choices=["graph", "fused", "reference"])
    ap.add_argument("--no-prefill", action="store_true", help="skip the timed T=2048 prefill (KV cache left zero-filled)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--context", type=int, default=None, help="decode at this context instead of 2048 (e.g. 32768, the reference's published regime): the prompt is prefilled in chunks of 2048 through the KV caches")
    ap.add_argument("--no-gpt2", action="store_true", help="skip the BASELINE config-2 entry (GPT-2 124M bf16, B=8, T=1024 forward)")
    ap.add_argument("--stub", action="store_true", help="plumbing test on a CPU box: no GPU work, the line says data = stub")
    ap.add_argument("--attn-split", type=int, default=None, help="tuning hook: positions per flash-decode split (default 64)")
    ap.add_argument("--resident", type=int, default=None, help="1/0: resident prefill staging for the quantized policies (default: the model's = 1)")
    ap.add_argument("--gemm-schedule", type=int, default=None, help="tuning hook: 0 lockstep, 1 ping-pong (4 phases), 3 ping-pong (2 phases, default)")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=VALUE", help="named tuning variable (csrc/internal.h: mila_cdna4_tune), repeatable; experiments only")
    ap.add_argument("--bounded-local-kv", type=int, default=0, help="1: SlidingWindowKvCache on the sliding-window layers (bounded ring of window + chunk - 1 rows)")
    a = ap.parse_args()

    # N > 1 and no launcher env: this process only starts one child per GPU (it never initialises a GPU itself)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_replicas(a.gpus, sys.argv[1:]))
    if a.attn_split is not None or a.gemm_schedule is not None or a.tune:
        os.environ["MILA_CDNA4_TUNING"] = "1"      # the tuning hooks are inert unless asked for before the library loads (csrc/internal.h)

    from mila_amd import host
    from mila_amd.replicas import Ranks
    ranks = Ranks(backend="gloo")                  # timing barrier + MAX-over-ranks only; no RCCL on this path (north star)
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
    if a.gpus != world:
        print("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d" % (a.gpus, world, world), file=sys.stderr)
    cfg = dict(host.GEMMA4_12B)
    if a.bounded_local_kv:
        cfg["bounded_local_kv"] = 1
    policies = [p for p in a.policies.split(",") if p]

    # the CPU baseline FIRST (rank 0 of the single-GPU line only): the GPU phases then run back to back to the end of the process
    cpu = cpu_baseline(cfg) if (rank == 0 and world == 1 and not a.no_cpu and not a.stub) else None

    if a.stub:
        if os.environ.get("MILA_BENCH_STUB_FAIL_RANK") == str(rank):      # tests: a replica that dies before the barrier must end the whole command, promptly
            os._exit(3)
        ranks.barrier()
        ms = stub_result(a.steps, a.warmup, rank)
        ranks.barrier()
        ms = ranks.max_over_ranks(ms)
        if rank == 0:
            print(json.dumps({"metric": "Gemma-4 12B decode tok/s (B=1, context 2048), 1xMI355X", "value": round(1e3 / ms * world, 2), "unit": "tok/s", "n_gpus": world,
                              "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "none", "data": "stub (no GPU work: launcher / barrier / aggregation plumbing only)",
                              "config": {"workload": "stub", "replicas": world, "parallelism": "replicas only (no collective)"}}), flush=True)
        ranks.close()
        return

    import torch
    torch.cuda.set_device(local_rank)
    from mila_amd import capi
    capi.load()
    capi.check(capi.load().mila_cdna4_set_device(local_rank))
    if a.attn_split is not None:
        capi.tune("attn.positions_per_split", a.attn_split)
    if a.gemm_schedule is not None:
        capi.tune("gemm.schedule", a.gemm_schedule)
    for kv in a.tune:
        name, value = kv.split("=", 1)
        capi.tune(name, int(value))
    results = {}
    CHUNK = CONTEXT                      # the prefill chunk (BASELINE config 3-5: T = 2048)
    ctx = a.context if a.context else CONTEXT
    if ctx < CHUNK:
        raise SystemExit("bench.py: --context must be at least %d" % CHUNK)
    for pol in policies:
        m = host.Gemma(pol, cfg, max_seq=ctx + a.steps + a.warmup + 8, max_prefill=1 if a.no_prefill else CHUNK, seed=1234)
        if a.resident is not None:
            m.set_resident_prefill_weights(a.resident)
        info = m.info(ctx)
        r = {"weight_GB": round(info["weight_bytes"] / 1e9, 3), "bytes_per_token_GB": round(info["decode_bytes_per_token"] / 1e9, 3)}
        # the quantized policies keep their prefill staging resident: HBM it costs beside the quantized weights, as the ops report it
        resident = a.resident is None or bool(a.resident)
        r["resident_prefill_weights_GB"] = round(m.resident_staging_bytes() / 1e9, 3)
        r["total_resident_GB"] = round((info["weight_bytes"] + m.resident_staging_bytes()) / 1e9, 3)      # weights + scales + tied table + the staging above (KV caches / activations aside)
        if not a.no_prefill:
            ms, ms_min, spread = timed_prefill(m)
            r["prefill_timing"] = "median of %d timed passes (HIP events, one T=2048 prefill each, each behind an untimed pass) after %d untimed; min and max - min beside it" % (PREFILL_TIMED, PREFILL_UNTIMED)
            r["prefill_ms_min"] = round(ms_min, 3)
            r["prefill_ms_spread"] = round(spread, 3)
            # algorithmic FLOPs: Linear 2*params*T + head (last position) + attention 4*NH*HS*sum(keys)
            lin = 2.0 * info["linear_params"] * CONTEXT + 2.0 * info["table_params"]
            att = 0.0
            for i in range(cfg["num_layers"]):
                g = (i + 1) % cfg["sliding_window_pattern"] == 0
                hd = cfg["global_head_dim"] if g else cfg["head_dim"]
                w = 0 if g else cfg["window"]
                keys = sum(min(t + 1, w) if w else t + 1 for t in range(CONTEXT))
                att += 4.0 * cfg["num_heads"] * hd * keys
            r["prefill_ms"] = round(ms, 3)
            r["prefill_TFLOPs"] = round((lin + att) / ms / 1e9, 2)
            r["prefill_tok_s"] = round(CONTEXT / ms * 1e3, 1)
            # the fp4 policy's Linears (97 % of the FLOPs) run on the fp8 matrix cores, attention on the bf16 ones: price each part at its own peak
            ideal_ms = (lin / (MFMA_FP8_PEAK_TFLOPS if pol == "fp4" else MFMA_BF16_PEAK_TFLOPS) + att / MFMA_BF16_PEAK_TFLOPS) / 1e9
            r["prefill_mfma_frac"] = round(ideal_ms / ms, 4)
            r["prefill_mfma_peak_TFLOPs"] = {"linear": MFMA_FP8_PEAK_TFLOPS if pol == "fp4" else MFMA_BF16_PEAK_TFLOPS, "attention": MFMA_BF16_PEAK_TFLOPS}
            if pol in ("fp8", "fp4") and resident:
                # the reference's semantics (CudaLinearOp.ixx:597-644, :648-715: the staging pass runs inside EVERY forward, on a 12 GB card): the same prefill with the
                # resident staging off -- same bits, tests/test_gemma_host_gpu.py
                m.set_resident_prefill_weights(0)
                r["prefill_ms_per_forward_staging"] = round(timed_prefill(m)[0], 3)
                m.set_resident_prefill_weights(1)
            if pol == "fp8":
                # BASELINE config 4 ("Linear<PerChannelFp8<>> weights, CDNA4 fp8_e4m3 MFMA"): the OPT-IN W8A8 prefill -- the policy's own e4m3 weights + per-channel scales
                # on the fp8 matrix cores, per-token e4m3 activations, no staging and no bf16 copy (Policies.ixx:39-40).  The default above stays the reference's W8A16.
                m.set_fp8_activation_prefill(1)
                ms8, ms8_min, _ = timed_prefill(m)
                r["prefill_ms_w8a8"] = round(ms8, 3)
                r["prefill_ms_w8a8_min"] = round(ms8_min, 3)
                r["prefill_TFLOPs_w8a8"] = round((lin + att) / ms8 / 1e9, 2)
                r["prefill_mfma_frac_w8a8"] = round((lin / MFMA_FP8_PEAK_TFLOPS + att / MFMA_BF16_PEAK_TFLOPS) / 1e9 / ms8, 4)      # Linears priced at the fp8 peak
                r["resident_prefill_weights_GB_w8a8"] = round(m.resident_staging_bytes() / 1e9, 3)
                r["total_resident_GB_w8a8"] = round((info["weight_bytes"] + m.resident_staging_bytes()) / 1e9, 3)
                m.set_fp8_activation_prefill(0)
            if ctx > CHUNK:
                # the long prompt as the L6 caller feeds it: chunks of 2048 through the KV caches (Gemma.ixx:234-267); fills the caches to the context
                cms = m.time_prefill_chunked(ctx)
                r["chunked_prefill_tokens"] = ctx
                r["chunked_prefill_ms"] = round(cms, 2)
                r["chunked_prefill_tok_s"] = round(ctx / cms * 1e3, 1)
        ranks.barrier()
        torch.cuda.synchronize()
        t = m.time_decode(ctx, a.steps, a.warmup, a.mode)
        torch.cuda.synchronize()
        ranks.barrier()
        t["wall_ms_per_step"] = ranks.max_over_ranks(t["wall_ms_per_step"])      # timing only: max over ranks
        k = m.time_dominant_kernel(3)
        r["launches_per_token"] = m.graph_node_count() if a.mode == "graph" else None
        r.update({"ms_per_step": round(t["wall_ms_per_step"], 4), "device_ms_per_step": round(t["device_ms_per_step"], 4),
                  "tok_s": round(1e3 / t["wall_ms_per_step"], 2),
                  "token_roofline_frac": round(info["decode_bytes_per_token"] / (t["wall_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                  "dominant_kernel": {"name": "matvec_kernel<fmt,R,U,PRO=2,GEGLU> (fc_gate_up)", "avg_us": round(k["avg_us"], 3),
                                      "bytes": k["bytes"], "GBps": round(k["bytes"] / k["avg_us"] / 1e3, 1)}})
        results[pol] = r
        m.close()
        del m

    # BASELINE config 2 beside the headline: GPT-2 124M bf16 forward, B = 8, T = 1024 (tools/bench_gpt2.py; rank 0 of the single-GPU line only)
    gpt2 = None
    if rank == 0 and world == 1 and not a.no_gpt2:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_gpt2
        gpt2 = bench_gpt2.run(5)

    # measured streaming ceiling beside the 8 TB/s spec: a read-only pass over 2 GiB with 16-byte loads
    import ctypes as C
    n = 2 << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    sink = torch.zeros(4096, dtype=torch.float32, device="cuda")
    lib = capi.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        capi.check(lib.mila_cdna4_stream_read(C.c_void_p(sink.data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(n), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        capi.check(lib.mila_cdna4_stream_read(C.c_void_p(sink.data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(n), st))
    e1.record()
    torch.cuda.synchronize()
    stream_read_gbps = round(5 * n / e0.elapsed_time(e1) / 1e6, 1)
    del src

    # the decode design floor: one launch per fused Linear / attention step means every token pays launches_per_token node boundaries on top of its
    # bytes at the rate a pure stream reaches on this box; what fraction of the 8 TB/s roofline that design can reach at best, per policy
    fixed_us = round(launch_boundary_us(capi), 3)
    for pol, r in results.items():
        if r.get("launches_per_token"):
            floor_ms = r["bytes_per_token_GB"] / stream_read_gbps * 1e3 + r["launches_per_token"] * fixed_us * 1e-3
            r["design_floor_ms"] = round(floor_ms, 4)
            r["design_floor_frac"] = round(r["bytes_per_token_GB"] / HBM_PEAK_GBPS * 1e3 / floor_ms, 4)

    head = results[policies[0]]
    out = {
        "metric": "Gemma-4 12B decode tok/s (B=1, context %d), 1xMI355X" % ctx,
        "value": round(head["tok_s"] * world, 2), "unit": "tok/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp8": "bf16 activations x fp8_e4m3 weights", "fp4": "bf16 activations x fp4_e2m1 weights"}[policies[0]],
        "data": "synthetic (counter-based uniform weights, random-init architecture; KV cache filled by %s)" % ("a T=2048 prefill" if ctx == CONTEXT else "a chunked prefill of %d tokens (chunks of 2048)" % ctx),
        "config": {"workload": "Gemma-4 12B, weight policy %s, B=1, prefill T=2048%s then decode at positions %d.." % (policies[0], "" if ctx == CONTEXT else " (chunked to %d)" % ctx, ctx),
                   "decode_mode": a.mode, "replicas": world, "parallelism": "replicas only (no collective)", "bounded_local_kv": int(bool(a.bounded_local_kv)),
                   **({"tune": a.tune} if a.tune else {})},
        "roofline": {"bound": "hbm", "achieved": head["dominant_kernel"]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(head["dominant_kernel"]["GBps"] / HBM_PEAK_GBPS, 4), "traffic": measured_traffic(policies[0])[0],
                     "traffic_source": measured_traffic(policies[0])[1],
                     "kernel": head["dominant_kernel"]["name"], "avg_us": head["dominant_kernel"]["avg_us"],
                     "algorithmic_bytes_per_launch": head["dominant_kernel"]["bytes"],
                     "whole_token_frac": head["token_roofline_frac"],
                     # the floor of THIS design (one graph node per fused Linear / attention step): bytes at the measured stream rate + launches x the cost of an empty node
                     "launches_per_token": head.get("launches_per_token"), "fixed_us_per_launch": fixed_us,
                     "design_floor_frac": {pol: r.get("design_floor_frac") for pol, r in results.items()},
                     "design_floor": "bytes_per_token / measured_stream_read + launches_per_token x fixed_us_per_launch, as a fraction of bytes_per_token / 8 TB/s",
                     # beside the 8 TB/s datasheet figure: what a read-only pass reaches on this box (the matvec's launch shape, non-temporal 16-byte loads,
                     # 2 GiB: a yardstick measured in the same process) and the guide's float4-copy figure; the whole token against each
                     "measured_stream_read_GBps": stream_read_gbps,
                     "frac_of_measured_stream_read": round(head["dominant_kernel"]["GBps"] / stream_read_gbps, 4),
                     "whole_token_frac_of_measured_stream_read": round(head["token_roofline_frac"] * HBM_PEAK_GBPS / stream_read_gbps, 4),
                     "whole_token_frac_of_guide_achievable_6300GBps": round(head["token_roofline_frac"] * HBM_PEAK_GBPS / 6300.0, 4)},
        "policies": results,
    }
    if gpt2 is not None:
        out["gpt2_124M_bf16_B8_T1024"] = gpt2
    if rank == 0:
        if cpu is not None:                    # the CPU baseline belongs to the single-GPU line only
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
