import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for B in (1, 8, 64, 512):
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
    ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
    for _ in range(10): ns(*sig)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): ns(*sig)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 20)
    print(f"group={os.environ.get('SONAR_RNG_GROUP','auto')} B={B}: {best:.1f} us/call")
