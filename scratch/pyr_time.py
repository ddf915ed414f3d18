"""Pyramid generator (raw and normalised, generate mode), batch 64 and 512: event-timed per call."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for b in (64, 512):
    x = torch.zeros(b, 4, 128, 128, device="cuda")
    for normalized in (False, True):
        torch.manual_seed(5)
        ns = nz.get_noise_sampler("pyramid", x, 0.03, 14.6, seed=None, cpu=False, normalized=normalized)
        for _ in range(300 if b == 64 else 60): ns(*sig)
        n = 100
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): ns(*sig)
        e1.record(); torch.cuda.synchronize()
        print(f"B={b:4d} normalized={normalized!s:5s}: {e0.elapsed_time(e1) / n * 1e3:7.1f} us per call", flush=True)
