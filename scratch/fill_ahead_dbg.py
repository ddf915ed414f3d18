"""Which plan a normalised Gaussian / uniform sampler gets, and what a call costs with and without the fill look-ahead."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for b in (512, 64):
    x = torch.zeros((b, 4, 128, 128), device="cuda")
    for name, factor in (("gaussian", 1.0), ("gaussian", 0.8), ("uniform", 1.0)):
        for ahead in (True, False):
            hl.FILL_AHEAD = ahead
            ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True, factor=factor)
            us = bench.event_us(lambda: ns(*sig), 100, 20)
            planned = ns if isinstance(ns, hl.Planned) else getattr(ns, "_planned", None)
            plan = planned.plan if planned is not None else None
            hooks = [type(h).__name__ + f"(hits={h.hits}, misses={h.misses})" for h in (plan.hooks if plan else [])]
            print(f"b={b:4d} {name:9s} factor={factor} ahead={ahead!s:5} {us:7.1f} us  plan={'yes' if plan else getattr(planned, 'reason', None)} len={hl.load().sonar_plan_length(plan.handle) if plan else 0} hooks={hooks}")
