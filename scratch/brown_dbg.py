import importlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
NG = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
x = torch.zeros(3, 4, 64, 64, device="cuda")
kept = NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=77)
bare = NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=77); bare.CACHE_POINTS = 0
sig = [14.6 * 0.9**k for k in range(61)]
for k in range(6):
    mid = math.sqrt(sig[k] * sig[k + 1])
    for pair in ((sig[k], mid), (sig[k], sig[k + 1])):
        a = kept(torch.tensor(pair[0]), torch.tensor(pair[1])); b = bare(torch.tensor(pair[0]), torch.tensor(pair[1]))
        print(k, pair, "kept var %.4f bare var %.4f maxdiff %.3e" % (a.var().item(), b.var().item(), (a - b).abs().max().item()), "cache", [round(t, 3) for t in kept._points])
