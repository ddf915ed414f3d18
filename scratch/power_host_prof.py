"""Host time per call of the normalised power-noise sampler (B SDXL latents, default 512; B=1: the launch-bound floor): issue time against GPU time, and the host profile."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
x = torch.zeros((int(os.environ.get("B", "512")), 4, 128, 128), device="cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
for _ in range(500): ns(*sig)
torch.cuda.synchronize()
n = 400
t0 = time.perf_counter()
for _ in range(n): ns(*sig)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {(t1 - t0) / n * 1e6:.1f} us per call; with the final sync {(t2 - t0) / n * 1e6:.1f} us per call")
for K in (20, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): ns(*sig)
    torch.cuda.synchronize()
    print(f"K = {K}: {(time.perf_counter() - t0) / K * 1e6:.1f} us per step (sync on both sides)")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): ns(*sig)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
