"""Brownian noise call on cfg5's shard (128 x 16 x 128 x 128) and on 64 SDXL latents: a DPM++ SDE run's two queries per step, event-timed,
kept-tensor bridge route vs whole-expansion route (CACHE_POINTS = 0)."""
import importlib, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
NG = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
for shape in ((128, 16, 128, 128), (64, 4, 128, 128)):
    x = torch.zeros(shape, device="cuda")
    for keep in (4, 0):
        ns = NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=7)
        ns.CACHE_POINTS = keep
        sig = torch.linspace(14.6, 0.05, 31).tolist()
        calls = []
        for k in range(30):
            calls += [(sig[k], math.sqrt(sig[k] * sig[k + 1])), (sig[k], sig[k + 1])]
        for a, b in calls[:6]:
            ns(torch.tensor(a), torch.tensor(b))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for a, b in calls[6:]:
            ns(torch.tensor(a), torch.tensor(b))
        e1.record(); torch.cuda.synchronize()
        n = len(calls) - 6
        print(f"{shape} keep={keep}: {e0.elapsed_time(e1) / n * 1e3:8.1f} us per call (wall {(time.perf_counter() - t0) / n * 1e6:8.1f})", flush=True)
