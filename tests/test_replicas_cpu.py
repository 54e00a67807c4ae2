"""world_size-2 gloo rehearsal of the replicas-only multi-GPU path used by bench.py --gpus N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_replicas_aggregate_with_max_time_and_no_data_collective():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "tools", "replica_selftest.py")],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world"] == 2
    assert abs(d["ms"] - 15.0) < 1e-9                    # MAX over ranks
    assert abs(d["value"] - 2 * 64 / 15e-3) < 1e-6       # units of all ranks / slowest rank's time


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from mila_amd.replicas import Ranks
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r = Ranks()
    assert r.world == 1 and r.dist is None
    v, ms = r.aggregate_throughput(10, 5.0)
    assert v == 10 / 5e-3 and ms == 5.0


def test_each_replica_builds_its_model_on_its_own_gpu(monkeypatch):
    """bench.py --gpus N is one process per GPU: the model runner must be created on the launcher's LOCAL_RANK device, not on
    device 0 (host.Gemma's default), or all replicas would share one card"""
    import inspect
    from mila_amd import host, replicas
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert replicas.local_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert replicas.local_device() == 5
    src = inspect.getsource(host.Gemma.__init__)
    assert "local_device()" in src and "mila_gemma_create(POLICIES[policy], C.byref(c), max_seq, max_prefill, seed, device)" in src


def _bench(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "6", "--warmup", "1", *args],
                         capture_output=True, text=True, env=env or os.environ, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_gpus_flag_starts_one_replica_per_gpu_and_aggregates():
    """`python bench.py --gpus 2` outside a launcher: the parent starts two children (it never touches a GPU itself), the line
    reports n_gpus 2, the time is the MAX over ranks (rank 1 is the slow one in the stub) and value = all ranks' steps / that time"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one, two = _bench(env=env), _bench("--gpus", "2", env=env)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["replicas"] == 2
    assert two["ms_per_step"] >= 3.0 > one["ms_per_step"] >= 2.0            # stub: 2 ms + 1 ms per rank
    assert abs(two["value"] - 2 * 1e3 / two["ms_per_step"]) < 0.02 * two["value"]
    assert two["data"].startswith("stub")


def test_bench_under_the_drivers_launcher():
    """the driver's own form: torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 (ranks from the env, no second spawn)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "2", "--steps", "6", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                   # rank 0 prints ONE line
    assert json.loads(lines[0])["n_gpus"] == 2


def test_a_replica_that_dies_ends_the_command_non_zero_and_promptly():
    """a rank >= 1 that exits before the barrier: the parent ends the others instead of leaving rank 0 in the gloo barrier until the
    process-group timeout, and the command's exit code is non-zero"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MILA_BENCH_STUB_FAIL_RANK"] = "1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "2", "--steps", "6", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0
    assert time.time() - t0 < 120
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
