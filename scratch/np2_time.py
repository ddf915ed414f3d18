import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for shape in ((512, 4, 128, 128), (512, 4, 104, 152), (512, 4, 96, 96)):
    x = torch.zeros(shape, device="cuda")
    for name in ("pyramid", "perlin", "gaussian", "brownian"):
        ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=3, cpu=False, normalized=True)
        for _ in range(3): ns(*sig)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): out = ns(*sig)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"{name} {shape}: {us:.0f} us/call -> {shape[0]/us:.2f} M latents/s  std {out.std().item():.4f}")
