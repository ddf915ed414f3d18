import sys, importlib
sys.path.insert(0, '.')
import torch, sonar_pkg
hl = sonar_pkg.load().hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
x = torch.zeros(512, 4, 128, 128, device='cuda'); n = x.numel()
sig = (torch.tensor(14.6), torch.tensor(10.0))
for name in ('perlin', 'pyramid', 'gaussian', 'uniform'):
    for normalized in (True, False):
        ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=normalized)
        t = timeit(lambda: ns(*sig))
        print(f"{name:10s} normalized={normalized!s:5s} {t:8.1f} us  {512/t:6.2f} M latents/s  12N-equiv {n*12/t/1e6:6.2f} TB/s")
