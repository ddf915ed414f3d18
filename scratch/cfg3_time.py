"""cfg3: Perlin + pyramid chain (normalised, generate mode), batch 64 and 512 SDXL latents: event-timed per call, and the kernels of one call."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for order in (("perlin", "pyramid"), ("pyramid", "perlin"), ("gaussian", "pyramid"), ("gaussian", "perlin")):
    for b in (64, 512):
        chain = nz.CustomNoiseChain()
        for name in order:
            chain.add(nz.CustomNoiseItem(0.5, noise_type=name))
        x = torch.zeros(b, 4, 128, 128, device="cuda")
        ns = chain.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
        for _ in range(300 if b == 64 else 60): ns(*sig)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): ns(*sig)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"{'+'.join(order):18s} B={b:4d}: {us:8.1f} us per call   {b / us:6.2f} M latents/s", flush=True)
