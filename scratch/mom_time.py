import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
sonar = importlib.import_module("comfyui_sonar_amd.py.sonar")
B=512; N=4*128*128
x = torch.randn((B, 4, 128, 128), device="cuda"); den = torch.randn_like(x)
sb = sonar.SonarBase(sonar.SonarBase.get_config(None, {}))
sb.momentum_step(0, x, den, torch.tensor(10.0), torch.tensor(8.0))
def t(fn, n=50):
    for _ in range(5): fn()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
us = t(lambda: sb.momentum_step(1, x, den, torch.tensor(8.0), torch.tensor(6.0)))
print(f"euler momentum step B=512: {us:.1f} us, {20*N*B/us/1e6:.2f} TB/s at 20N")
a = torch.randn_like(x)
us = t(lambda: hl.blend("lerp", x, a, 0.3))
print(f"blend lerp: {us:.1f} us, {12*N*B/us/1e6:.2f} TB/s at 12N")
us = t(lambda: hl.axpby_(a, 1.0, x, 0.5))
print(f"axpby: {us:.1f} us, {12*N*B/us/1e6:.2f} TB/s at 12N")
p = hl.stats(x)
us = t(lambda: hl.scale_noise_(a, 1.0, True, p))
print(f"scale_noise apply: {us:.1f} us, {8*N*B/us/1e6:.2f} TB/s at 8N")
us = t(lambda: hl.stats(x, p))
print(f"stats: {us:.1f} us, {4*N*B/us/1e6:.2f} TB/s at 4N")
