"""Spectral filter at 128 x 128 against torch.fft (the checker), plain and with the statistics / normalised variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
torch.manual_seed(3)
worst = 0.0
for b, filt in ((3, torch.rand(128, 65) + 0.5), (17, torch.randn(128, 65)), (64, torch.ones(128, 65)), (5, torch.rand(128, 65) * torch.linspace(0, 3, 65))):
    x = torch.randn(b, 4, 128, 128, device="cuda")
    f = filt.cuda()
    want = torch.fft.irfft2(torch.fft.rfft2(x.double()) * f.double(), s=(128, 128))
    got = hl.spectral_filter(x, f)
    err = (got.double() - want).abs().max().item() / want.abs().max().item()
    worst = max(worst, err)
    print(f"batch {b}: max rel err {err:.2e}")
    assert err < 2e-6, err
print("ok", worst)
# the statistics variant: same values, partials = (sum, sum of squares) of the output
x = torch.randn(9, 4, 128, 128, device="cuda"); f = (torch.rand(128, 65) + 0.5).cuda()
part = torch.zeros(2 * 2048, dtype=torch.float64, device="cuda")
a = hl.spectral_filter(x, f); b = hl.spectral_filter(x, f, partials=part)
assert torch.equal(a, b)
p = part.view(-1, 2).sum(0)
assert abs(p[0].item() - a.double().sum().item()) < 1e-6 * a.numel() and abs(p[1].item() / (a.double() ** 2).sum().item() - 1) < 1e-9, p
print("statistics ok")
# buffers aligned to 8 bytes only (the ABI's contract)
flat = torch.randn(2 * 4 * 128 * 128 + 2, device="cuda")
xo = flat[2:].view(2, 4, 128, 128)
assert xo.data_ptr() % 16 == 8
want = torch.fft.irfft2(torch.fft.rfft2(xo.double()) * f.double(), s=(128, 128))
got = hl.spectral_filter(xo, f)
assert (got.double() - want).abs().max().item() < 2e-6 * want.abs().max().item()
print("8-byte aligned input ok")
# ... and an OUTPUT aligned to 8 bytes only, through the C ABI itself (the store pass writes 16 bytes at a time)
import ctypes as C
lib = hl.load()
oflat = torch.zeros(2 * 4 * 128 * 128 + 2, device="cuda")
o = oflat[2:]
assert o.data_ptr() % 16 == 8
fn = lib.sonar_spectral_filter_f32
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
assert fn(xo.data_ptr(), f.data_ptr(), o.data_ptr(), 8, 128, 128, None, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
assert torch.equal(o.view(2, 4, 128, 128), got) and float(oflat[:2].abs().sum()) == 0.0
print("8-byte aligned output ok")
