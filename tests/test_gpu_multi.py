"""-m gpu, needs >= 2 GPUs (skipped on the 1-GPU test boxes): the N > 1 path on real devices over RCCL -- ShardedNoiseSampler +
normalise_global_ + gather_batch (ring all-gather and direct peer copies) against the single-process result.  The ranks are
fresh child processes started before anything here touches the GPU."""
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


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least 2 GPUs")
def test_two_ranks_over_rccl_match_one_process():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "tests", "multi_gpu_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "multi-gpu ok 2" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
