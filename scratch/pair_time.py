"""Normalised power-law noise, 512 SDXL latents, through the sampler: us per call with the look-ahead statistics (product default) and
without (SONAR_NO_LOOKAHEAD=1 in the environment of this script: the sampler is handed no PowerLookahead)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for B in (512, 256, 128, 96):
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    for label, off in (("look-ahead", False), ("plain", True)):
        real = hl.power_noise
        if off:
            hl.power_noise = lambda *a, lookahead=None, **k: real(*a, **k)
        try:
            ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
            for _ in range(600): ns(*sig)
            best = 1e9
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(100): ns(*sig)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 10)
        finally:
            hl.power_noise = real
        print(f"B={B} {label:10s}: {best:.1f} us per call -> {B / best:.3f} M latents/s", flush=True)
