"""Off-fast-path planes, 33.5 M values: the normalised call (statistics launch + final pass) beside the un-normalised generating launch alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
for tag, (hh, ww, nb) in {"128x128": (128, 128, 512), "64x64": (64, 64, 2048), "32x32": (32, 32, 8192), "128x64": (128, 64, 1024), "104x152": (104, 152, 530), "112x144": (112, 144, 520), "256x256": (256, 256, 128)}.items():
    fz = torch.rand(hh, ww // 2 + 1, device="cuda") + 0.5
    shp, ctr = (nb, 4, hh, ww), [0]
    def both():
        ctr[0] += 1
        return hl.power_noise(fz, shp, seed=11, stream_id=ctr[0], plane_offset=0, factor=1.0)
    def gen():
        ctr[0] += 1
        return hl.power_irfft2(None, fz, shp, seed=11, stream_id=ctr[0], plane_offset=0)
    a = sorted(bench.event_us(both, 20, 5) for _ in range(3))[1]
    b = sorted(bench.event_us(gen, 20, 5) for _ in range(3))[1]
    print(f"{tag:8s}: normalised call {a:7.1f} us   generating launch alone {b:7.1f} us   (statistics launch + boundary ~{a - b:5.1f})", flush=True)
