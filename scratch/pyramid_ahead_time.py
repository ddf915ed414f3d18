"""Normalised pyramid call with and without the round-6 look-ahead (one launch per call inside a plan), batch 64 / 512 / 4."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for b in (4, 64, 256, 512):
    x = torch.zeros((b, 4, 128, 128), device="cuda")
    for ahead in (True, False):
        hl.PYRAMID_AHEAD = ahead
        ns = nz.get_noise_sampler("pyramid", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        for _ in range(30):
            ns(*sig)
        host, gpu = bench.host_and_event_us(lambda: ns(*sig), steps=200, warmup=50)
        planned = ns if isinstance(ns, hl.Planned) else getattr(ns, "_planned", None)
        hooks = [f"hits={h.hits} misses={h.misses}" for h in (planned.plan.hooks if planned and planned.plan else [])]
        print(f"b={b:4d} ahead={ahead!s:5}: gpu {gpu:7.1f} us  host {host:6.1f} us per call  {hooks}", flush=True)
