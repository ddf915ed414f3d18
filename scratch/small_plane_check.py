"""irfft2 of a supplied spectrum and the spectral filter on the smallest fixed-size planes, several plane counts, against torch.fft."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
def poison():
    """leave large values in every CU's LDS (a kernel that reads LDS it did not write shows up as a mismatch after this)"""
    zz = torch.full((512, 128, 65), 1e30, dtype=torch.complex64, device="cuda")
    hl.power_irfft2(zz, torch.ones(128, 65, device="cuda"), (512, 1, 128, 128))
    zz = torch.full((64, 200, 101), 1e30, dtype=torch.complex64, device="cuda")
    hl.power_irfft2(zz, torch.ones(200, 101, device="cuda"), (64, 1, 200, 200))
    hl.power_irfft2(None, torch.full((256, 129), 1e30, device="cuda"), (8, 4, 256, 256), seed=1, stream_id=1)


bad = 0
for (H, W) in ((16, 16), (32, 32), (64, 32), (32, 64), (64, 64)):
    for planes in (1, 2, 3, 5, 8, 13, 64, 300):
        for rep in range(3):
            g = torch.Generator(device="cuda").manual_seed(1000 * rep + planes)
            K = W // 2 + 1
            z = torch.randn(planes, H, K, dtype=torch.complex64, device="cuda", generator=g)
            filt = torch.rand(H, K, device="cuda", generator=g) + 0.5
            if os.environ.get("POISON"): poison()
            got = hl.power_irfft2(z, filt, (planes, 1, H, W)).reshape(planes, H, W)
            want = torch.fft.irfft2(z * filt, s=(H, W), norm="ortho")
            err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
            x = torch.randn(planes, H, W, device="cuda", generator=g)
            if os.environ.get("POISON"): poison()
            f2 = hl.spectral_filter(x, filt)
            w2 = torch.fft.irfft2(torch.fft.rfft2(x, norm="ortho") * filt, s=(H, W), norm="ortho")
            err2 = (f2 - w2).abs().max().item() / max(1.0, w2.abs().max().item())
            if err > 3e-5 or err2 > 3e-5:
                bad += 1
                per = ((got - want).abs().amax((1, 2)) > 1e-4).nonzero().flatten().tolist()
                print(f"MISMATCH {H}x{W} planes {planes} rep {rep}: irfft2 {err:.2e} filter {err2:.2e}; bad planes {per[:12]}", flush=True)
print("bad", bad)
