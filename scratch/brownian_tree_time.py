"""Brownian-interval noise on cfg5's shard (128 x 16 x 128 x 128), a DPM++ SDE run's pattern ((t, s) and (t, t') per step): us per call with
the default path of bridges and with the opt-in virtual Brownian tree (depth 24)."""
import importlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
NG = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
x = torch.zeros(128, 16, 128, 128, device="cuda")
sig = [14.6 * 0.93**k for k in range(41)]
for depth in (0, 24, 16):
    ns = NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=5, tree_depth=depth)
    def run(k0, k1):
        for k in range(k0, k1):
            ns(torch.tensor(sig[k]), torch.tensor(math.sqrt(sig[k] * sig[k + 1])))
            ns(torch.tensor(sig[k]), torch.tensor(sig[k + 1]))
    run(0, 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    run(8, 40)
    e1.record(); torch.cuda.synchronize()
    print(f"tree depth {depth:2d}: {e0.elapsed_time(e1) / 64 * 1e3:8.1f} us per call", flush=True)
