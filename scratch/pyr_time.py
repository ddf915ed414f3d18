import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for B in (64, 512):
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    for name in ("pyramid", "perlin"):
        ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        for _ in range(5): ns(*sig)
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30): ns(*sig)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
        print(f"{name} B={B}: {best:.1f} us/call -> {B/best:.3f} M latents/s, {12*65536*B/best/1e6:.2f} TB/s at 12N")
