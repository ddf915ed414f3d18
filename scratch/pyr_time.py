"""Pyramid noise through the sampler API at batch 512 / 64: us per call (normalised and raw), and the cfg3 chain."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
def ev(fn, n=100, w=100):
    for _ in range(w): fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
for B in (512, 64):
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    for normalized in (True, False):
        ns = nz.get_noise_sampler("pyramid", x, 0.03, 14.6, seed=None, cpu=False, normalized=normalized)
        print(f"pyramid B={B} normalized={normalized}: {ev(lambda: ns(*sig)):.1f} us per call", flush=True)
