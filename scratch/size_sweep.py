"""the normalised power-law call (draw + filter + irfft2 + normalise, hip_lib.power_noise) over plane sizes at the same number of
elements (33.5 M = 512 SDXL latents): us per call and output bytes per second, so every size reads against the 128 x 128 path; the
spectral filter (rfft2 x gain, irfft2: the OneF family) on the same planes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
TOTAL = 512 * 4 * 128 * 128
sizes = [(128, 128), (64, 64), (32, 32), (16, 16), (256, 128), (128, 256), (128, 64), (64, 128), (96, 96), (160, 160), (192, 192),
         (120, 120), (144, 112), (104, 152), (168, 96), (192, 80), (136, 104), (80, 80), (48, 48), (100, 100), (90, 160), (240, 136), (256, 256)]
if len(sys.argv) > 1: sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for H, W in sizes:
    planes = max(4, TOTAL // (H * W) // 4 * 4)
    shape = (planes // 4, 4, H, W)
    filt = (torch.rand(H, W // 2 + 1, device="cuda") + 0.5).contiguous()
    kind = hl.power_plane_kind(H, W)
    def call(i): return hl.power_noise(filt, shape, seed=7, stream_id=100 + i, plane_offset=0, factor=1.0)
    for i in range(20): call(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 100
    for i in range(n): call(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    x = torch.randn(shape, device="cuda")
    for i in range(10): hl.spectral_filter(x, filt)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): hl.spectral_filter(x, filt)
    e1.record(); torch.cuda.synchronize()
    fus = e0.elapsed_time(e1) / n * 1e3
    print(f"{H:4d} x {W:4d} kind {kind} planes {planes:6d}: power noise {us:8.1f} us per call, {planes * H * W * 4 / us / 1e6:5.2f} TB/s of output;"
          f" spectral filter {fus:8.1f} us, {2 * planes * H * W * 4 / fus / 1e6:5.2f} TB/s read + written")
