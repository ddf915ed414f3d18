"""Timed regions of K headline steps bracketed by synchronize (bench.py's contract): per-step time against K -> the fixed cost of a region."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1")
x = torch.zeros(512, 4, 128, 128, device="cuda")
ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
sig = (torch.tensor(14.6), torch.tensor(10.0))
step = lambda: ns(*sig)
for _ in range(1500): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize()
for K in (5, 10, 20, 50, 100, 200, 20, 20):
    walls, spans = [], []
    for rep in range(12):
        for _ in range(5): step()
        torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K): step()
        e1.record()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / K * 1e6)
        spans.append(e0.elapsed_time(e1) / K * 1e3)
    walls.sort(); spans.sort()
    print(f"K={K:4d}: wall us/step median {walls[len(walls)//2]:6.1f} min {walls[0]:6.1f} max {walls[-1]:6.1f} | event span median {spans[len(spans)//2]:6.1f} min {spans[0]:6.1f}", flush=True)
