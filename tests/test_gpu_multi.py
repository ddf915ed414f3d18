"""The N > 1 path on real devices.  Ranks are fresh child processes (the children never inherit an initialised GPU: they are new
programs started by torch.distributed.run).
  * two ranks over RCCL, one GPU each (needs >= 2 GPUs, skipped on the 1-GPU test boxes): ShardedNoiseSampler + normalise_global_ +
    gather_batch (ring all-gather and direct peer copies) against the single-process result, single generators and cfg5's chain;
  * the same worker with two ranks SHARING cuda:0 over gloo (collectives staged through the host): every kernel and every line of
    host code of the sharded path runs on a one-GPU box;
  * `bench.py --gpus 2` as the driver calls it: starts its own ranks, reports n_gpus from the process group, a `ranks` list and the
    gather timings under `extra` (gloo / shared GPU here; over RCCL when two GPUs exist)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_worker(backend):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SONAR_TEST_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "tests", "multi_gpu_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and f"multi-gpu ok 2 {backend}" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least 2 GPUs")
def test_two_ranks_over_rccl_match_one_process():
    _run_worker("nccl")


def test_two_ranks_sharing_one_gpu_match_one_process():
    _run_worker("gloo")


def _bench(n, backend):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SONAR_BENCH_BACKEND=backend)
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--prewarm", "0"],
                         env=env, capture_output=True, text=True, timeout=900)
    return res


def test_bench_starts_its_own_ranks():
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    res = _bench(2, backend)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 1024 and out["scaling"] == "weak"
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and [r["shard_start"] for r in out["ranks"]] == [0, 512]
    assert out["roofline"]["peak"] == 16000.0 and 0 < out["roofline"]["frac"] < 1
    g = out["extra"]["gather"]
    assert g["backend"] == backend and g["rccl_all_gather_ms"] > 0 and g["direct_peer_copies_ms"] > 0, g
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    # the line says which devices ran (two ranks sharing cuda:0 show the same address), what carried the collectives, and times the
    # 8-GPU configurations on every rank's own shard
    for r in out["ranks"]:
        assert r["gcn_arch"].startswith("gfx950") and r["device_name"]
        assert "pci_bus_id" in r or "uuid" in r
    assert out["collectives"]["backend"] == backend and out["collectives"]["world_size"] == 2 and "rccl_version" in out["collectives"]
    ex = out["extra"]
    assert [r["rank"] for r in ex["per_rank"]] == [0, 1]
    assert ex["cfg5_shard_step_ms"]["ranks"] == 2 and 0 < ex["cfg5_shard_step_ms"]["min"] <= ex["cfg5_shard_step_ms"]["max"]
    assert ex["cfg3_chain_b64_us"]["ranks"] == 2 and ex["cfg3_chain_b64_latents_per_s"] > 0 and ex["cfg5_latent_steps_per_s"] > 0


@pytest.mark.skipif(torch.cuda.device_count() >= 3, reason="the box really has 3 GPUs")
def test_bench_refuses_more_gpus_than_the_node_has():
    res = _bench(3, "nccl")
    assert res.returncode != 0 and '"metric"' not in res.stdout
    assert "needs 3 GPUs" in res.stderr
