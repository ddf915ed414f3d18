"""Which CPU-side torch ops run when a sampler is created and called (and how much CPU time all threads burn): torch profiler table of
20 creations + 20 calls per sampler kind, plus process CPU time against wall time."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
from torch.profiler import profile, ProfilerActivity
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
dev = torch.device("cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
x4 = torch.zeros((4, 4, 128, 128), device=dev); x64 = torch.zeros((64, 4, 128, 128), device=dev)
def make(kind):
    if kind == "power": return bench.power_item(pn).make_noise_sampler(x4, None, None, seed=None, cpu=False, normalized=True)
    return nz.get_noise_sampler(kind, x64, 0.03, 14.6, seed=None, cpu=False, normalized=True)
for kind in ("power", "perlin", "pyramid"):
    ns = make(kind); ns(*sig); torch.cuda.synchronize()
    t0 = time.perf_counter(); c0 = os.times()
    for _ in range(20):
        ns = make(kind)
        for _ in range(20): ns(*sig)
    torch.cuda.synchronize()
    c1 = os.times(); t1 = time.perf_counter()
    print(f"{kind}: wall {t1 - t0:.3f} s, process CPU {c1.user - c0.user + c1.system - c0.system:.3f} s (threads: {torch.get_num_threads()})", flush=True)
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        for _ in range(5):
            ns = make(kind)
            for _ in range(5): ns(*sig)
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=40, max_shapes_column_width=60), flush=True)
