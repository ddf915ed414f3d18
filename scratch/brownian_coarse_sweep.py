"""Tree call on cfg5's shard against the coarse level of the two-stage evaluation, for three schedules (DPM++ SDE's (t, s), (t, t') per step)."""
import importlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
NG = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
x = torch.zeros(128, 16, 128, 128, device="cuda")
def schedule(kind, n):
    if kind == "linear":
        return torch.linspace(14.6, 0.5, n + 1).tolist()
    if kind == "geometric":
        return [14.6 * (0.03 / 14.6) ** (k / n) for k in range(n + 1)]
    rho = 7.0  # karras
    lo, hi = 0.03 ** (1 / rho), 14.6 ** (1 / rho)
    return [(hi + k / n * (lo - hi)) ** rho for k in range(n + 1)]
for kind, n in (("linear", 10), ("linear", 20), ("karras", 20), ("karras", 40), ("geometric", 30)):
    sig = schedule(kind, n)
    row = []
    for level in (2, 3, 4, 5, 6):
        NG.BrownianTreeNoiseSampler.TREE_COARSE_LEVEL = level
        ns = NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=5, tree_depth=24)
        def run(k0, k1):
            for k in range(k0, k1):
                ns(torch.tensor(sig[k]), torch.tensor(math.sqrt(sig[k] * sig[k + 1])))
                ns(torch.tensor(sig[k]), torch.tensor(sig[k + 1]))
        run(0, 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        run(2, n)
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / (2 * (n - 2)) * 1e3)
    print(f"{kind:10s} {n:3d} steps: " + "  ".join(f"L{l}: {v:6.1f}" for l, v in zip((2, 3, 4, 5, 6), row)) + "  us per call", flush=True)
