import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
for (h, w, B) in ((128, 128, 512), (64, 64, 2048), (32, 32, 8192), (16, 16, 32768), (256, 128, 256), (128, 256, 256), (128, 64, 1024), (64, 128, 1024), (256, 64, 512), (64, 256, 512)):
    filt = torch.rand(h, w // 2 + 1, device="cuda") + 0.5
    shape = (B, 4, h, w)
    f = lambda: hl.power_noise(filt, shape, seed=1, stream_id=0, plane_offset=0, factor=1.0)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"{h}x{w} B={B}: {us:.0f} us ({4*h*w*4*B/us/1e6:.2f} TB/s written, {3*4*h*w*4*B/us/1e6:.2f} TB/s at 12N)")
