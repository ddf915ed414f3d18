"""Host enqueue cost vs GPU time per call for the generate-mode samplers (power-law, Perlin, pyramid; batch 512 and 64)."""
import cProfile, pstats, importlib, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for b in (512, 64):
    x = torch.zeros(b, 4, 128, 128, device="cuda")
    for name in ("power", "perlin", "pyramid", "gaussian"):
        ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True) if name == "power" else nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        for _ in range(20): ns(*sig)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): ns(*sig)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"B={b:4d} {name:9s} host {(t1 - t0) / 200 * 1e6:6.1f} us/call   total {(t2 - t0) / 200 * 1e6:6.1f} us/call", flush=True)
x = torch.zeros(64, 4, 128, 128, device="cuda")
for name in sys.argv[1:] or ["perlin"]:
    ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True) if name == "power" else nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    for _ in range(20): ns(*sig)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): ns(*sig)
    pr.disable(); torch.cuda.synchronize()
    print("=====", name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)
