import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
for (h, w) in ((104, 152), (96, 96), (128, 160), (168, 96), (120, 120)):
    B = 512
    filt = torch.rand(h, w // 2 + 1, device="cuda") + 0.5
    shape = (B, 4, h, w)
    f = lambda: hl.power_noise(filt, shape, seed=1, stream_id=0, plane_offset=0, factor=1.0)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    x = torch.randn(shape, device="cuda")
    g = lambda: hl.spectral_filter(x, filt)
    for _ in range(2): g()
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): g()
    e1.record(); torch.cuda.synchronize()
    print(f"{h}x{w}: power_noise {us:.0f} us / 512 latents ({B/us:.2f} M latents/s, {12*4*h*w*4*B/us/1e6:.2f} TB/s at 12N); spectral filter {e0.elapsed_time(e1)*200:.0f} us")
