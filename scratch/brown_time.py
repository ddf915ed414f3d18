import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
for B in (1, 64, 512):
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    ns = nz.get_noise_sampler("brownian", x, 0.03, 14.6, seed=7, cpu=False, normalized=False)
    f = lambda: ns(torch.tensor(9.0), torch.tensor(7.5))
    for _ in range(3): f()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    print(f"brownian B={B}: {best:.1f} us/call -> {B/best:.3f} M latents/s")
