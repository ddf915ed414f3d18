"""Planes beyond LDS (kind 4), generated: the column-block route against round 3's white-noise route, per 33.5 M values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()


def timed(fn, n=60, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for shape in ((128, 4, 256, 256), (32, 4, 256, 256), (32, 4, 512, 512), (16, 4, 384, 512), (1, 4, 256, 256)):
    H, W = shape[-2:]
    filt = (torch.rand(H, W // 2 + 1, device="cuda") + 0.5)
    vals = shape[0] * shape[1] * H * W
    new_norm = timed(lambda: hl.power_noise(filt, shape, seed=1, stream_id=2, plane_offset=0, factor=1.0))
    new_raw = timed(lambda: hl.power_irfft2(None, filt, shape, seed=1, stream_id=2, plane_offset=0))

    def old():
        ws = hl.new_partials("cuda")
        white = hl.philox_normal(tuple(shape), "cuda", 1, 2, 0)
        return hl.scale_noise_(hl._direct_spectral_filter(white, filt, ws), 1.0, True, ws)
    old_norm = timed(old)
    k = 33554432 / vals
    print(f"{shape}: normalised {new_norm:8.1f} us ({new_norm * k:7.1f} per 33.5 M values) | raw {new_raw:8.1f} | white-noise route {old_norm:8.1f} ({old_norm * k:7.1f})", flush=True)
