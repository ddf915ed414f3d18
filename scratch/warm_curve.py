"""Launch-pair time of the headline step as the process warms up: 40 blocks of 50 steps, event-timed per block."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
x = torch.zeros(512, 4, 128, 128, device="cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
t00 = time.perf_counter()
row = []
for blk in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ns(*sig)
    e1.record(); torch.cuda.synchronize()
    row.append(e0.elapsed_time(e1) / 50 * 1e3)
print("us per step per block of 50:", " ".join(f"{v:.1f}" for v in row))
print("total s", time.perf_counter() - t00)
time.sleep(2.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): ns(*sig)
torch.cuda.synchronize(); e0.record()
for _ in range(20): ns(*sig)
e1.record(); torch.cuda.synchronize()
print("after 2 s idle, 5 warm-up + 20 timed:", e0.elapsed_time(e1) / 20 * 1e3)
