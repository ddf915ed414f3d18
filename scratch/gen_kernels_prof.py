"""Gaussian / Perlin / pyramid normalised generate at batch 512 and 64 for a kernel trace (per-kernel durations of the generator passes)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for b in (512, 64):
    x = torch.zeros(b, 4, 128, 128, device="cuda")
    for name in ("gaussian", "perlin", "pyramid", "uniform"):
        ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        for _ in range(300 if b == 512 else 100): ns(*sig)
        torch.cuda.synchronize()
