import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
def t(f, n=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
shape = (512, 4, 128, 128)
filt = torch.rand(128, 65, device="cuda") + 0.5
x = torch.randn(shape, device="cuda")
z = hl.power_spectrum(shape, "cuda", seed=1, stream_id=0)
print("spectral filter", round(t(lambda: hl.spectral_filter(x, filt)), 1), "replay irfft2", round(t(lambda: hl.power_irfft2(z, filt, shape)), 1),
      "generate irfft2", round(t(lambda: hl.power_irfft2(None, filt, shape, seed=1, stream_id=0)), 1))
