import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = torch.linspace(14.6, 0.03, 41).tolist()  # a 40-step run: every call ends where the next begins
for B in (1, 64, 512):
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    best = 1e9
    for rep in range(3):
        ns = nz.get_noise_sampler("brownian", x, 0.03, 14.6, seed=7, cpu=False, normalized=False)
        ns(torch.tensor(sig[0]), torch.tensor(sig[1]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for a, b in zip(sig[1:-1], sig[2:]): ns(torch.tensor(a), torch.tensor(b))
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 39)
    print(f"brownian B={B}: {best:.1f} us/call -> {B/best:.3f} M latents/s")
