import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for (h, w) in ((104, 152), (96, 96)):
    B = 512
    filt = torch.rand(h, w // 2 + 1, device="cuda") + 0.5
    shape = (B, 4, h, w)
    z = hl.power_spectrum(shape, "cuda", seed=1, stream_id=0)
    print(h, w, "replay", round(t(lambda: hl.power_irfft2(z, filt, shape))), "gen", round(t(lambda: hl.power_irfft2(None, filt, shape, seed=1, stream_id=0))),
          "gen+norm", round(t(lambda: hl.power_noise(filt, shape, seed=1, stream_id=0, plane_offset=0, factor=1.0))), "dump", round(t(lambda: hl.power_spectrum(shape, "cuda", seed=1, stream_id=0))))
