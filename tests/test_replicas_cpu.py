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
