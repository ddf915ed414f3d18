"""Spectral filter (irfft2(rfft2(x) * filter), 512 SDXL latents) and the replay-mode power noise: us per launch."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
def ev(fn, n=100, w=200):
    for _ in range(w): fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
x = torch.randn(512, 4, 128, 128, device="cuda")
filt = torch.rand(128, 65, device="cuda") + 0.5
print(f"spectral filter b512: {ev(lambda: hl.spectral_filter(x, filt)):.1f} us")
z = torch.randn(512 * 4, 128, 65, 2, device="cuda")
zc = torch.view_as_complex(z)
print(f"power_irfft2 replay b512: {ev(lambda: hl.power_irfft2(zc, filt, (512, 4, 128, 128))):.1f} us")
x64 = torch.randn(64, 4, 128, 128, device="cuda")
print(f"spectral filter b64: {ev(lambda: hl.spectral_filter(x64, filt)):.1f} us")
print(f"power generate b64 (phase-serial kernel): {ev(lambda: hl.power_irfft2(None, filt, (64, 4, 128, 128), seed=1, stream_id=2)):.1f} us")
