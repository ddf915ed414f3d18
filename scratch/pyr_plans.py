import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
x = torch.zeros((512, 4, 128, 128), device="cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
orig = ng.PyramidNoiseGenerator._plan
plans = []
def spy(self, h, w, draw):
    out = list(orig(self, h, w, draw)); plans.append([(a[1], a[2]) for a in out]); return iter(out)
ng.PyramidNoiseGenerator._plan = spy
ns = nz.get_noise_sampler("pyramid", x, 0.03, 14.6, seed=(int(sys.argv[1]) if len(sys.argv) > 1 else None), cpu=False, normalized=True)
for s in range(12):
    torch.manual_seed(s)
    ns(*sig); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.manual_seed(s); e0.record(); ns(*sig); e1.record(); torch.cuda.synchronize()
    print(s, f"{e0.elapsed_time(e1)*1e3:.0f} us", plans[-1])
