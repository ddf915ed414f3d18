"""CPU, world_size = 2 over gloo: the N > 1 path of the batch-sharded noise generator.
Covers the partition function, the 3-double statistics all-reduce that gives exact whole-batch normalisation,
and the optional final gather (incl. uneven shards).  The device kernels themselves are covered by the -m gpu
shard-invariance tests (one process, virtual ranks)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sonar_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, global_batch, out_dir):
    import sonar_pkg

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sonar_pkg.load()
    par = importlib.import_module("comfyui_sonar_amd.parallel")
    try:
        torch.manual_seed(0)  # every rank can rebuild the logical batch; it only works on its shard
        full = torch.randn(global_batch, 4, 8, 8) * 1.3 + 0.2
        start, count = par.shard_range(global_batch, rank, world)
        local = full[start:start + count].clone()
        d = local.double()
        stats = par.allreduce_stats(torch.tensor([d.sum(), (d * d).sum(), float(d.numel())], dtype=torch.float64))
        gd = full.double()
        assert torch.allclose(stats, torch.tensor([gd.sum(), (gd * gd).sum(), float(gd.numel())], dtype=torch.float64), rtol=1e-12)
        # exact whole-batch normalisation from the reduced statistics == the oracle on the full batch
        n = stats[2].item()
        mean = stats[0].item() / n
        std = ((stats[1].item() - stats[0].item() * mean) / (n - 1)) ** 0.5
        want = orc.scale_noise(full.clone(), 0.9, normalized=True)[start:start + count]
        thr = 2.5 / n**0.5
        got = local.clone()
        if abs(mean) > thr:
            got -= torch.tensor(mean, dtype=torch.float32)
        if abs(1.0 - torch.tensor(std, dtype=torch.float32).item()) > thr:
            got /= torch.tensor(std, dtype=torch.float32)
        got *= 0.9
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
        gathered = par.gather_batch(local, global_batch)
        assert torch.equal(gathered, full)
        assert torch.equal(par.gather_batch(local, global_batch, direct=True), full)  # world - 1 concurrent point-to-point copies
        torch.save(torch.tensor([start, count]), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 7])
def test_two_rank_shard_stats_and_gather(tmp_path, global_batch):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), global_batch, str(tmp_path)), nprocs=world, join=True)
    spans = [torch.load(tmp_path / f"rank{r}.pt").tolist() for r in range(world)]
    assert spans[0][0] == 0 and spans[0][0] + spans[0][1] == spans[1][0] and spans[1][0] + spans[1][1] == global_batch


def test_shard_range_partitions_every_batch(pkg):
    par = importlib.import_module("comfyui_sonar_amd.parallel")
    for world in (1, 2, 4, 8):
        for batch in (0, 1, 7, 8, 512, 1024, 1027):
            spans = [par.shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == batch
            for (s0, c0), (s1, _c1) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert par.shard_range(1024, 3, 8) == (384, 128)  # cfg5: 128 Flux latents per GPU
    with pytest.raises(ValueError):
        par.shard_range(8, 2, 2)


def test_bench_refuses_to_mislabel_the_gpu_count():
    """`python bench.py --gpus 2` on a node with fewer GPUs ends with a non-zero code and no result line (it used to measure one GPU
    and print n_gpus: 1); the launcher role touches no GPU, so this runs anywhere."""
    import subprocess
    import sys

    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SONAR_BENCH_BACKEND")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and '"metric"' not in res.stdout and "needs 2 GPUs" in res.stderr


def test_bench_annotations_survive_a_damaged_profile_file(tmp_path, monkeypatch):
    """The bench line's HBM-traffic and vector-ALU annotations come from tracked profile files; a missing, empty or damaged one must not
    cost the run its line (round 5: an empty profiles/r05_traffic.json made every rank of `bench.py --gpus 2` exit with a JSON error)."""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sonar_bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert isinstance(bench.traffic_table(), dict) and bench.traffic_table()  # the tracked table of the round parses
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.traffic_table() == {} and bench.valu_active_per_launch() is None           # nothing there
    (tmp_path / "profiles" / "r05_traffic.json").write_text("")
    (tmp_path / "profiles" / "r04_traffic.json").write_text(json.dumps({"power_noise_b512": {"hbm_bytes_per_launch": 1}}))
    assert bench.traffic_table() == {"power_noise_b512": {"hbm_bytes_per_launch": 1}}        # the newest one that parses
    (tmp_path / "profiles" / "r04_traffic.json").write_text("{not json")
    assert bench.traffic_table() == {}
    (tmp_path / "profiles" / "r05_pmc_issue_a.txt").write_text("void sonar::power_pipe_kernel<128, 128, false, true>\n  SQ_ACTIVE_INST_VALU\n")
    assert bench.valu_active_per_launch() is None
