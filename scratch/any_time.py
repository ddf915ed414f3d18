"""Power-law noise (normalised, generate mode) at latent sizes off the fast path: event-timed per call, batch 512 (and the 128 x 128 fast path beside it)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for (h, w) in ((128, 128), (104, 152), (112, 144), (96, 96), (96, 168), (160, 96), (64, 64), (136, 136)):
    x = torch.zeros(512, 4, h, w, device="cuda")
    ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
    for _ in range(5): ns(*sig)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ns(*sig)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{h:4d} x {w:4d}: {us:8.1f} us per 512 latents   {512 * 4 * h * w * 4 / us / 1e3:7.1f} GB/s written   {us / (h * w) * 16384 / 1:8.1f} us per 128x128-equivalent", flush=True)
