"""Randomised differential runs of the spectral kernels over plane sizes (every route: fixed-size, general-size with codelets, line
transforms pass by pass, direct sums for odd widths): irfft2 of a supplied spectrum and the spectral filter against torch.fft on the
device, generate mode against the replay of its own dumped spectrum (LDS-resident routes), statistics partials against the tensor.
python scratch/fuzz_spectral.py [iterations] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
kinds = {}
for it in range(iters):
    style = rnd.random()
    if style < 0.45:   # latent-like: multiples of 8
        H, W = 8 * rnd.randint(1, 40), 8 * rnd.randint(1, 40)
    elif style < 0.8:  # any even
        H, W = 2 * rnd.randint(1, 150), 2 * rnd.randint(1, 150)
    elif style < 0.9:  # odd somewhere
        H, W = rnd.randint(1, 200), rnd.randint(1, 200)
    else:              # long lines
        H, W = rnd.choice([2, 4, 6, 10, 34]) * rnd.randint(1, 4), 2 * rnd.randint(300, 1024)
        if rnd.random() < 0.5: H, W = W, H
    kind = hl.power_plane_kind(H, W)
    if kind == 0: continue
    kinds[kind] = kinds.get(kind, 0) + 1
    planes = rnd.choice([1, 2, 3, 5, 8])
    if it < int(os.environ.get("FUZZ_FROM", "0")): continue  # replay a tail of the sequence (the shapes still follow the seed)
    K = W // 2 + 1
    g = torch.Generator(device="cuda").manual_seed(it)
    z = torch.randn(planes, H, K, dtype=torch.complex64, device="cuda", generator=g)
    filt = torch.rand(H, K, device="cuda", generator=g) + 0.5
    shape = (planes, 1, H, W)
    tag = f"it {it}: {planes} x {H} x {W} kind {kind}"
    try:
        part = hl.new_partials("cuda")
        got = hl.power_irfft2(z, filt, shape, partials=part).reshape(planes, H, W)
        want = torch.fft.irfft2(z * filt, s=(H, W), norm="ortho")
        err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
        sums = part.view(-1, 2).sum(0)
        serr = abs(sums[1].item() - (got.double() ** 2).sum().item()) / max(1e-30, (got.double() ** 2).sum().item())
        x = torch.randn(planes, H, W, device="cuda", generator=g)
        f2 = hl.spectral_filter(x, filt)
        w2 = torch.fft.irfft2(torch.fft.rfft2(x, norm="ortho") * filt, s=(H, W), norm="ortho")
        err2 = (f2 - w2).abs().max().item() / max(1.0, w2.abs().max().item())
        err3 = 0.0
        if kind in (1, 2):
            sh4 = (planes, 4, H, W)
            zz = hl.power_spectrum(sh4, "cuda", seed=it, stream_id=3, plane_offset=4)
            a = hl.power_irfft2(None, filt, sh4, seed=it, stream_id=3, plane_offset=4)
            b = hl.power_irfft2(zz, filt, sh4)
            # round 5: the generate path takes the filter under the radius' square root, the replay multiplies the dumped value by it: the
            # two agree to the last bits of every spectrum value (tests/test_gpu_kernels.py GEN_VS_REPLAY_ATOL), not bit for bit
            d3 = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
            err3 = 0.0 if d3 < 2e-6 else d3 + 1.0
        if kind == 4:  # generated in column blocks: against the direct passes over its own dumped spectrum, and the normalised call against scale_noise
            sh4 = (planes, 4, H, W)
            zz = hl.power_spectrum(sh4, "cuda", seed=it, stream_id=3, plane_offset=4)
            p4 = hl.new_partials("cuda")
            a = hl.power_irfft2(None, filt, sh4, seed=it, stream_id=3, plane_offset=4, partials=p4)
            b = hl.power_irfft2(zz, filt, sh4)
            e_ab = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
            nrm = hl.power_noise(filt, sh4, seed=it, stream_id=3, plane_offset=4, factor=0.8)
            ref = hl.scale_noise_(a.clone(), 0.8, True, p4)
            e_n = (nrm - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            err3 = 0.0 if (e_ab < 3e-5 and e_n < 3e-5) else 1.0 + e_ab + e_n
        if not (err < 3e-5 and err2 < 3e-5 and serr < 1e-5 and err3 == 0.0) and H * W <= 65536:
            # who is right?  irfft2 by explicit DFT sums in fp64 (Hermitian extension of the half-spectrum, imaginary parts of the DC / Nyquist
            # columns ignored as irfft does), against both
            zf = (z * filt).to(torch.complex128)
            ky = torch.arange(H, device="cuda", dtype=torch.float64)
            cols = torch.fft.ifft(zf, dim=1, norm="ortho") if False else torch.einsum("yk,pkx->pyx", torch.exp(2j * torch.pi * ky[:, None] * ky[None, :] / H), zf) / H ** 0.5
            kx = torch.arange(K, device="cuda", dtype=torch.float64)
            xs = torch.arange(W, device="cuda", dtype=torch.float64)
            wgt = torch.full((K,), 2.0, device="cuda", dtype=torch.float64); wgt[0] = 1.0
            if W % 2 == 0: wgt[-1] = 1.0
            ph = torch.exp(2j * torch.pi * kx[:, None] * xs[None, :] / W)
            c = cols.clone(); c[:, :, 0] = c[:, :, 0].real + 0j
            if W % 2 == 0: c[:, :, -1] = c[:, :, -1].real + 0j
            exact = (torch.einsum("pyk,kx->pyx", c * wgt, ph)).real / W ** 0.5
            e_mine = (got.double() - exact).abs().max().item() / max(1.0, exact.abs().max().item())
            e_torch = (want.double() - exact).abs().max().item() / max(1.0, exact.abs().max().item())
            print(f"   {tag}: against explicit fp64 DFT sums: this library {e_mine:.2e}, torch.fft.irfft2 {e_torch:.2e}")
            if e_mine < 3e-5 and e_torch > 3e-5 and serr < 1e-5 and err3 == 0.0:
                # seen once in 1500 iterations (seed 3, iteration 298, 8 x 16 x 16, only after the ~290 sizes before it in the same process):
                # torch.fft.irfft2 itself returned a wrong transform (3.6e-1), reproducibly; the checker, not the library
                torch_off = globals().get("torch_off", 0) + 1
                globals()["torch_off"] = torch_off
                continue
        if not (err < 3e-5 and err2 < 3e-5 and serr < 1e-5 and err3 == 0.0):
            bad += 1
            print(f"MISMATCH {tag}: irfft2 {err:.2e} filter {err2:.2e} sumsq {serr:.2e} generate-vs-replay {err3:.2e}")
    except Exception as exc:
        bad += 1
        print(f"ERROR {tag}: {type(exc).__name__}: {str(exc)[:160]}")
print(f"{iters} iterations, routes {kinds}, {bad} bad" + (f", {globals()['torch_off']} where torch.fft (the checker) was the one off" if globals().get("torch_off") else ""))
sys.exit(1 if bad else 0)
