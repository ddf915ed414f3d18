"""Kernel-level parity (-m gpu): every C-ABI entry point of libsonar_hip.so, called through ctypes on a
real MI355X, against the CPU oracle (oracle/sonar_oracle.py) on the committed golden vectors and on
seeded inputs.

Tolerance (fp32 path, stated per north_star "within a stated fp32 tolerance"):
  elementwise / normalisation / momentum : rtol 1e-5, atol 1e-6  (op order restated; differences are
      last-bit: device division/transcendentals, fp64-accumulated statistics vs torch's fp32)
  FFT-based power noise                   : atol 2e-5 on unit-variance outputs (different FFT factorisation)
Index math (lattice corners, resampling taps, row permutations) must be exact: it is checked through
values that would differ by O(1) if an index were off.
"""
import math

import pytest
import torch

from oracle import sonar_oracle as orc

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6
FFT_ATOL = 2e-5
GEN_VS_REPLAY_ATOL = 2e-6  # generated planes against the replay of their own dumped spectrum: two roundings' difference per spectrum value


@pytest.fixture(scope="module")
def hl(pkg):
    lib = pkg.hip_lib
    lib.load()
    return lib


def dev(t):
    return t.contiguous().cuda()


def close(a, b, rtol=RTOL, atol=ATOL):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------------ scale_noise
@pytest.mark.parametrize("case", ["plain", "shifted", "scaled", "both", "tiny_shift"])
def test_scale_noise_golden(hl, golden, case):
    g = golden("scale_noise")
    x = dev(g[f"{case}_in"])
    factor, sub, div = (float(v) for v in g[f"{case}_meta"])
    part = hl.stats(x)
    tot = hl.stats_finalize(part, x.numel()).cpu()
    ref = g[f"{case}_in"].double()
    assert abs(tot[0].item() - ref.sum().item()) < 1e-6 * ref.abs().sum().item()
    assert abs(tot[1].item() - (ref * ref).sum().item()) < 1e-9 * (ref * ref).sum().item() + 1e-9
    hl.scale_noise_(x, factor, True, part)
    close(x, g[f"{case}_out"])
    # the data-dependent branches must match the reference's decisions
    dec = {}
    orc.scale_noise(g[f"{case}_in"].clone(), factor, normalized=True, decisions=dec)
    assert (dec["sub"], dec["div"]) == (bool(sub), bool(div))


def test_scale_noise_unnormalized_and_rows(hl, golden):
    g = golden("scale_noise")
    x = dev(g["dims_in"])
    hl.scale_noise_(x, 1.9, False, None)
    close(x, g["unnorm_out"])
    x = dev(g["dims_in"])
    b, c, h, w = x.shape
    hl.scale_noise_rows_(x, b * c, h * w, 0.7)
    close(x, g["dims_out"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 1023, 65536 + 3, 4 * 128 * 128 * 8])
def test_scale_noise_sizes_and_alignment(hl, n):
    torch.manual_seed(n)
    src = torch.randn(n + 1) * 1.5 + 0.25
    for off in (0, 1):  # off=1 -> 4-byte aligned only: scalar path
        x = dev(src)[off:off + n]
        want = orc.scale_noise(src[off:off + n].clone(), 0.9, normalized=True)
        part = hl.stats(x)
        hl.scale_noise_(x, 0.9, True, part)
        if n == 1:
            assert torch.isnan(want).all() == torch.isnan(x.cpu()).all() or torch.allclose(want, x.cpu(), equal_nan=True)
        else:
            close(x, want, rtol=2e-5, atol=2e-6)


def test_scale_noise_global_stats_override(hl):
    """n_total / partials from a wider tensor (the cross-rank all-reduce variant)."""
    torch.manual_seed(1)
    full = torch.randn(4, 4, 16, 16) * 0.8 + 0.1
    want = orc.scale_noise(full.clone(), 1.0, normalized=True)
    xf = dev(full)
    tot = hl.stats_finalize(hl.stats(xf), full.numel())
    shard = dev(full[2:])
    part = torch.zeros(hl.NPART * 2, dtype=torch.float64, device="cuda")
    part[:2] = tot[:2]
    hl.scale_noise_(shard, 1.0, True, part, n_total=full.numel())
    close(shard, want[2:])


# ------------------------------------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("mode", ["lerp", "inject", "subtract_b"])
@pytest.mark.parametrize("t", [0.0, 0.3, 0.5, 0.75, 1.0, -0.2, 1.4])
def test_blend_scalar(hl, mode, t):
    torch.manual_seed(5)
    a, b = torch.randn(3, 4, 9, 7), torch.randn(3, 4, 9, 7)
    close(hl.blend(mode, dev(a), dev(b), t), orc.blend(mode, a, b, t))


@pytest.mark.parametrize("mode", ["lerp", "inject"])
def test_blend_tensor_weight(hl, mode):
    torch.manual_seed(6)
    a, b = torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8)
    w = torch.rand(2, 4, 8, 8)
    close(hl.blend(mode, dev(a), dev(b), dev(w)), orc.blend(mode, a, b, w))
    w1 = torch.rand(1, 1, 8, 8)
    close(hl.blend(mode, dev(a), dev(b), dev(w1)), orc.blend(mode, a, b, w1))


def test_axpby_and_mask_mix(hl):
    torch.manual_seed(7)
    y, x = torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8)
    close(hl.axpby_(dev(y), 1.0, dev(x), 1.0), y + x)
    close(hl.axpby_(dev(y), 1.0, dev(x), 0.37), y + x * 0.37)
    close(hl.axpby_(dev(y), -0.5, dev(x), 2.0), y * -0.5 + x * 2.0)
    mask = torch.rand(2, 1, 8, 8)
    want = y * (torch.ones_like(mask) - mask) + x * mask
    got = hl.mask_mix(dev(y), dev(x), dev(mask.expand(2, 4, 8, 8).contiguous()))
    close(got, want)
    m1 = torch.rand(1, 1, 8, 8)
    close(hl.mask_mix(dev(y), dev(x), dev(m1)), y * (1 - m1) + x * m1)


def test_minmax_rows(hl):
    torch.manual_seed(8)
    x = torch.randn(6, 1000)
    lo, hi = hl.minmax_rows(dev(x), 6, 1000)
    assert torch.equal(lo.cpu(), x.amin(1)) and torch.equal(hi.cpu(), x.amax(1))


# ------------------------------------------------------------------------------------------------ momentum kernels
def make_cfg(hl, c: orc.MomentumCfg, st: orc.MomentumState, step: int, h_present: bool, h_fresh=False):
    cfg = hl.MomentumCfg()
    cfg.momentum = c.momentum
    cfg.hist_ratio, cfg.hist_scale, cfg.md_scale = st.ratios
    cfg.mode = hl.MODE_IDS[c.mode]
    cfg.momentum_blend = hl.BLEND_IDS[st.mblend]
    cfg.history_blend = hl.BLEND_IDS[st.hblend]
    cfg.use_momentum = int(st.check_step(step))
    hist_ok = st.check_step(step, is_history=True)
    cfg.update_hist = int(c.momentum_hist != 1 and hist_ok)
    cfg.init_kind = hl.INIT_IDS[c.init] if (not h_present and hist_ok and c.init in ("SAMPLE", "SAMPLE_NORM")) else 0
    cfg.h_in_fresh = int(h_fresh)
    return cfg


CFGS = [
    orc.MomentumCfg(),
    orc.MomentumCfg(mode="CLASSIC"),
    orc.MomentumCfg(mode="DENOISED"),
    orc.MomentumCfg(direction=-0.5),
    orc.MomentumCfg(mode="CLASSIC", momentum=0.8, momentum_hist=0.5, direction=1.5),
    orc.MomentumCfg(mode="DENOISED", init="SAMPLE"),
    orc.MomentumCfg(init="SAMPLE_NORM"),
    orc.MomentumCfg(blend_mode="inject", momentum=0.3, momentum_hist=0.4),
    orc.MomentumCfg(momentum_blend_mode="subtract_b", history_blend_mode="inject", momentum=0.2, momentum_hist=0.3),
    orc.MomentumCfg(momentum=1.0),
    orc.MomentumCfg(momentum_hist=1.0, init="SAMPLE"),
    orc.MomentumCfg(momentum_start_step=1, momentum_end_step=1, always_update_history=False),
]


@pytest.mark.parametrize("cfg", CFGS, ids=lambda c: f"{c.mode}-{c.init}-{c.blend_mode}-m{c.momentum}-h{c.momentum_hist}-d{c.direction}")
def test_momentum_euler_kernel_three_steps(hl, cfg):
    """Three consecutive fused steps (no history -> history created -> history updated) vs the oracle."""
    torch.manual_seed(9)
    shape = (2, 4, 8, 8)
    x = torch.randn(shape) * 5
    st = orc.MomentumState(cfg)
    xd, hd = dev(x), None
    sig = [torch.tensor(s) for s in (7.0, 4.0, 2.0, 1.0)]
    for step in range(3):
        den = x * 0.5 + torch.tanh(x) * 0.1
        want = st.euler_step(step, x, den, sig[step], sig[step + 1])
        kc = make_cfg(hl, cfg, orc.MomentumState(cfg), step, hd is not None)
        dt = (sig[step + 1] - sig[step]).item()
        xd, hd = hl.momentum_euler(xd, dev(den), hd, kc, sig[step].item(), dt)
        close(xd, want, rtol=2e-5, atol=2e-5)
        assert (hd is None) == (st.h is None)
        if hd is not None:
            close(hd, st.h, rtol=2e-5, atol=2e-5)
        x = want


def test_momentum_euler_with_noise(hl):
    torch.manual_seed(10)
    cfg = orc.MomentumCfg()
    x, den, nz = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    st = orc.MomentumState(cfg)
    want = st.euler_step(0, x, den, torch.tensor(3.0), torch.tensor(2.5)) + nz * (1.1 * 0.7)
    kc = make_cfg(hl, cfg, st, 0, False)
    got, h = hl.momentum_euler(dev(x), dev(den), None, kc, 3.0, -0.5, noise=dev(nz), noise_scale=1.1 * 0.7)
    close(got, want)
    close(h, st.h)


# ------------------------------------------------------------------------------------------------ Philox generators
def test_philox_normal_moments_and_shard_invariance(hl):
    n = 1 << 22
    x = hl.philox_normal((n,), "cuda", seed=1234, stream_id=3)
    m, s = x.mean().item(), x.std().item()
    assert abs(m) < 4 / math.sqrt(n) and abs(s - 1) < 4 / math.sqrt(2 * n)
    kurt = ((x - m) ** 4).mean().item() / s**4
    assert abs(kurt - 3.0) < 0.02
    # same values regardless of how the range is split across calls ("ranks"), incl. odd offsets
    for cut in (n // 2, 4096 * 5, 4 * 1001, 4 * 1001 + 1, 7):
        a = hl.philox_normal((cut,), "cuda", 1234, 3, 0)
        b = hl.philox_normal((n - cut,), "cuda", 1234, 3, cut)
        assert torch.equal(torch.cat((a, b)), x)
    assert not torch.equal(hl.philox_normal((1024,), "cuda", 1234, 4), x[:1024])  # stream id matters
    assert not torch.equal(hl.philox_normal((1024,), "cuda", 1235, 3), x[:1024])  # seed matters


def test_philox_uniform_range_and_stats_partials(hl):
    n = 1 << 20
    part = hl.new_partials("cuda")
    u = hl.philox_uniform((n,), "cuda", seed=7, stream_id=0, partials=part)
    assert u.min().item() >= 0.0 and u.max().item() < 1.0
    assert abs(u.mean().item() - 0.5) < 4 * math.sqrt(1 / 12 / n)
    tot = hl.stats_finalize(part, n).cpu()
    assert abs(tot[0].item() - u.double().sum().item()) < 1e-7 * n  # per-group fp32 partial sums feed the fp64 accumulators
    assert abs(tot[1].item() - (u.double() ** 2).sum().item()) < 1e-7 * n
    v = hl.philox_uniform((n,), "cuda", seed=7, stream_id=0, sub=0.5, mul=3.46, add=0.0)
    close(v, (u - 0.5) * 3.46)


# ------------------------------------------------------------------------------------------------ Perlin
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_perlin_replay_golden(hl, golden, tag):
    g = golden("perlin")
    mode = str(g[f"{tag}_blend"])
    terms = hl.perlin_terms(dev(g[f"{tag}_angles"]), mode)
    want_terms = torch.stack([orc.perlin_term(a, mode) for a in g[f"{tag}_angles"]])
    close(terms, want_terms, rtol=1e-5, atol=2e-6)  # device sinf/cosf vs torch CPU
    part = hl.new_partials("cuda")
    raw = hl.perlin_apply(dev(g[f"{tag}_base"]), terms, 2.0, part)
    close(raw, g[f"{tag}_raw"], rtol=1e-5, atol=2e-6)
    hl.scale_noise_(raw, 1.0, True, part)
    close(raw, g[f"{tag}_out"], rtol=2e-5, atol=5e-6)


def test_perlin_generate_matches_apply_on_device_draws(hl):
    shape = (3, 4, 16, 20)
    torch.manual_seed(0)
    angles = torch.rand(2, 4, 17, 21) * 2 * math.pi
    terms = hl.perlin_terms(dev(angles))
    u = hl.philox_uniform(shape, "cuda", seed=99, stream_id=5, elem_offset=64)
    want = hl.perlin_apply(u, terms, 2.0)
    p1, p2 = hl.new_partials("cuda"), hl.new_partials("cuda")
    got = hl.perlin_generate(shape, terms, 2.0, seed=99, stream_id=5, elem_offset=64, partials=p1)
    assert torch.equal(got, want)
    hl.stats(got, p2)
    t1, t2 = hl.stats_finalize(p1, got.numel()).cpu(), hl.stats_finalize(p2, got.numel()).cpu()
    assert torch.allclose(t1, t2, rtol=1e-7, atol=1e-9)  # fused kernel folds fp32 4-element partials into fp64


# ------------------------------------------------------------------------------------------------ resampling / pyramid
@pytest.mark.parametrize("mode", ["bilinear", "nearest-exact", "area"])
@pytest.mark.parametrize("src_hw,dst_hw", [((3, 3), (32, 32)), ((32, 32), (32, 32)), ((1, 1), (16, 24)), ((7, 5), (24, 40)),
                                            ((48, 64), (16, 16)), ((20, 36), (8, 12))])
def test_resample_matches_interpolate(hl, mode, src_hw, dst_hw):
    torch.manual_seed(11)
    src = torch.randn(2, 3, *src_hw)
    dst0 = torch.randn(2, 3, *dst_hw)
    want = dst0 + orc.resize(src, dst_hw[1], dst_hw[0], mode).mul_(0.7)
    got = hl.resample_acc_(dev(dst0), dev(src), 0.7, mode, True)
    close(got, want, rtol=1e-5, atol=2e-6)
    got = hl.resample_acc_(dev(dst0), dev(src), 1.0, mode, False)
    close(got, orc.resize(src, dst_hw[1], dst_hw[0], mode), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_pyramid_replay_golden(hl, golden, tag):
    g = golden("pyramid")
    noise = dev(g[f"{tag}_base"])
    nl = int(g[f"{tag}_nlevels"])
    discount, mode = float(g[f"{tag}_discount"]), str(g[f"{tag}_mode"])
    part = hl.new_partials("cuda")
    for i in range(nl):
        hl.resample_acc_(noise, dev(g[f"{tag}_level{i}"]), discount**i, mode, True, part if i == nl - 1 else None)
    close(noise, g[f"{tag}_raw"], rtol=1e-5, atol=3e-6)
    hl.scale_noise_(noise, 1.0, True, part)
    close(noise, g[f"{tag}_out"], rtol=2e-5, atol=5e-6)


@pytest.mark.parametrize("shape,mode", [((2, 4, 32, 32), "bilinear"), ((3, 4, 64, 64), "bilinear"), ((2, 4, 128, 128), "bilinear"),
                                        ((2, 4, 64, 64), "nearest-exact"), ((2, 4, 64, 64), "area"), ((2, 3, 32, 48), "bilinear")])
def test_pyramid_generate_is_sum_of_its_levels(hl, shape, mode):
    """Generate mode = base draw * sqrt(1 + w0^2) (the full-resolution level folded in: sum of two independent normals)
    + sum of the resampled small levels.  All of them run in the LDS-staged plane kernel (32x32 and 32x48 planes share RNG tiles with their neighbours)."""
    planes = shape[0] * shape[1]
    H, W = shape[-2:]
    sizes = [(H, W), (H // 3 + 1, W // 3 - 1), (2, 3), (1, 1)]
    weights = [1.0, 0.7, 0.49, 0.343]
    seed = 42
    small = [hl.philox_normal((planes, h, w), "cuda", seed, 10 + i) for i, (h, w) in enumerate(sizes[1:])]
    levels = [(None, H, W, weights[0])] + [(t, h, w, wt) for t, (h, w), wt in zip(small, sizes[1:], weights[1:])]
    part = hl.new_partials("cuda")
    got = hl.pyramid_generate(shape, "cuda", levels, mode, seed, 0, 0, part)
    want = hl.philox_normal(shape, "cuda", seed, 0) * math.sqrt(1.0 + weights[0] ** 2)
    for t, wt in zip(small, weights[1:]):
        hl.resample_acc_(want, t, wt, mode, True)
    close(got, want, rtol=2e-6, atol=2e-6)
    tot = hl.stats_finalize(part, got.numel()).cpu()
    assert abs(tot[0].item() - got.double().sum().item()) < 1e-7 * got.numel()
    # shard invariance: the second latent alone (elem_offset = one latent, level grids sliced) reproduces its slice
    per = shape[1] * H * W
    lv2 = [(None, H, W, weights[0])] + [(t[shape[1]:2 * shape[1]].contiguous(), h, w, wt) for t, (h, w), wt in zip(small, sizes[1:], weights[1:])]
    one = hl.pyramid_generate((1, *shape[1:]), "cuda", lv2, mode, seed, 0, per, None)
    assert torch.equal(one[0], got[1])


def test_pyramid_flat_kernel_for_levels_beyond_lds(hl):
    """A level grid too large for the plane kernel's LDS budget (126 x 126 floats) takes the flat global-gather kernel: same values."""
    shape = (2, 4, 128, 128)
    seed = 9
    big = hl.philox_normal((8, 126, 126), "cuda", seed, 10)
    small = hl.philox_normal((8, 5, 7), "cuda", seed, 11)
    levels = [(None, 128, 128, 1.0), (big, 126, 126, 0.7), (small, 5, 7, 0.49)]
    part = hl.new_partials("cuda")
    got = hl.pyramid_generate(shape, "cuda", levels, "bilinear", seed, 0, 0, part)
    want = hl.philox_normal(shape, "cuda", seed, 0) * math.sqrt(2.0)
    hl.resample_acc_(want, big, 0.7, "bilinear", True)
    hl.resample_acc_(want, small, 0.49, "bilinear", True)
    close(got, want, rtol=2e-6, atol=2e-6)
    tot = hl.stats_finalize(part, got.numel()).cpu()
    assert abs(tot[1].item() - (got.double() ** 2).sum().item()) < 1e-6 * tot[1].item()


# ------------------------------------------------------------------------------------------------ power-law rFFT noise
@pytest.mark.parametrize("tag", ["cfg2", "b", "c", "d", "e", "np2", "np2_rot", "odd", "sdxl_portrait"])
def test_power_noise_replay_golden(hl, golden, tag):
    g = golden("power_noise")
    z = torch.view_as_complex(g[f"{tag}_z"].contiguous())
    shape = tuple(g[f"{tag}_out"].shape)
    filt = g[f"{tag}_filter"]
    mixer = g[f"{tag}_mixer"]
    identity = torch.equal(mixer, torch.eye(mixer.shape[0]))
    part = hl.new_partials("cuda")
    out = hl.power_irfft2(dev(z), dev(filt.reshape(filt.shape[-2:])), shape, partials=part if identity else None)
    if not identity:
        out = hl.channel_mix(out, dev(mixer), part)
    close(out, g[f"{tag}_pre"], rtol=0, atol=FFT_ATOL)
    hl.scale_noise_(out, 1.0, bool(g[f"{tag}_normalized"]), part)
    close(out, g[f"{tag}_out"], rtol=0, atol=2 * FFT_ATOL)


@pytest.mark.parametrize("hw", [(16, 16), (32, 32), (64, 64), (128, 128), (64, 128), (128, 64), (32, 64), (64, 32),
                                (256, 128), (128, 256), (256, 64), (64, 256)])
def test_power_irfft2_all_supported_shapes(hl, hw):
    torch.manual_seed(13)
    h, w = hw
    z = torch.randn(3, 2, h, w // 2 + 1, dtype=torch.complex64)
    filt = torch.rand(h, w // 2 + 1) + 0.5
    want = torch.fft.irfft2(z * filt, s=(h, w), norm="ortho")
    got = hl.power_irfft2(dev(z), dev(filt), (3, 2, h, w))
    close(got, want, rtol=0, atol=FFT_ATOL)


@pytest.mark.parametrize("hw", [(16, 16), (32, 32), (64, 64), (128, 128), (64, 128), (128, 64), (32, 64), (64, 32),
                                (256, 128), (128, 256), (256, 64), (64, 256)])
def test_spectral_filter_all_supported_shapes(hl, hw):
    """Forward r2c + filter + inverse c2r, all in LDS, against torch.fft on the host (oracle.spectral_filter)."""
    torch.manual_seed(17)
    h, w = hw
    x = torch.randn(3, 2, h, w)
    filt = torch.rand(h, w // 2 + 1) + 0.5
    part = hl.new_partials("cuda")
    got = hl.spectral_filter(dev(x), dev(filt), part)
    want = orc.spectral_filter(x, filt)
    close(got, want, rtol=0, atol=FFT_ATOL)
    tot = hl.stats_finalize(part, got.numel()).cpu()
    assert abs(tot[0].item() - got.double().sum().item()) < 1e-6 * got.numel()
    # unit filter: the transform pair is the identity
    close(hl.spectral_filter(dev(x), dev(torch.ones(h, w // 2 + 1))), x, rtol=0, atol=FFT_ATOL)


@pytest.mark.parametrize("planes", [1, 5, 513, 1027])
def test_spectral_filter_128_kernel_of_its_own(hl, planes):
    """Round 5: 128 x 128 planes run spectral_filter128_kernel (paired forward row pass, fused forward / filter / inverse column pass).
    A filter with signs and zeros whose kx = 0 and kx = W/2 columns differ and are not even in ky exercises the packed column's
    weights; plane counts that leave the persistent workgroups unequal shares exercise the loop split at the prefetch; an input that is
    aligned to 8 bytes only is the ABI's contract (the kernel loads 16 bytes at a time)."""
    torch.manual_seed(planes)
    filt = torch.randn(128, 65)
    filt[3, 0] = 0.0
    filt[:, 64] *= 2.0
    filt[7, 9] = 0.0
    flat = torch.randn(planes * 128 * 128 + 2, device="cuda")
    x = flat[2:].view(planes, 128, 128)
    assert x.data_ptr() % 16 == 8
    part = hl.new_partials("cuda")
    got = hl.spectral_filter(x, dev(filt), part)
    want = torch.fft.irfft2(torch.fft.rfft2(x.double()) * filt.double().cuda(), s=(128, 128))
    assert (got.double() - want).abs().max().item() < 2e-6 * want.abs().max().item()
    tot = hl.stats_finalize(part, got.numel()).cpu()
    assert abs(tot[0].item() - got.double().sum().item()) < 1e-6 * got.numel() * max(1.0, got.abs().max().item())
    assert abs(tot[1].item() / (got.double() ** 2).sum().item() - 1.0) < 1e-6
    assert torch.equal(got, hl.spectral_filter(x.clone(), dev(filt)))  # the statistics variant stores the same values, whatever the alignment


def test_spectral_filter_full_batch_linearity_and_std_scale(hl):
    """BASELINE size (512 x 4 x 128 x 128): linearity F(a x + b y) = a F(x) + b F(y), and x *= mul / std."""
    torch.manual_seed(2)
    x = torch.randn(512, 4, 128, 128, device="cuda")
    y = torch.randn(512, 4, 128, 128, device="cuda")
    filt = (torch.rand(128, 65) + 0.25).cuda()
    fx, fy = hl.spectral_filter(x, filt), hl.spectral_filter(y, filt)
    part = hl.new_partials("cuda")
    fz = hl.spectral_filter(0.5 * x - 2.0 * y, filt, part)
    close(fz, 0.5 * fx - 2.0 * fy, rtol=0, atol=4 * FFT_ATOL)
    sd = fz.double().std().item()
    hl.std_scale_(fz, 3.0, part)
    assert abs(fz.double().std().item() - 3.0) < 1e-5
    assert sd > 0


def test_power_generate_equals_replay_of_device_draws(hl):
    shape = (5, 4, 128, 128)
    h, w = shape[-2:]
    filt = dev(torch.rand(h, w // 2 + 1) + 0.25)
    z = hl.power_spectrum(shape, "cuda", seed=77, stream_id=9, plane_offset=12)  # as if three latents preceded this shard
    assert abs(z.real.std().item() - math.sqrt(0.5)) < 3e-3 and abs(z.imag.mean().item()) < 6e-3  # 166k samples: sigma of the mean = 1.7e-3
    p1 = hl.new_partials("cuda")
    got = hl.power_irfft2(None, filt, shape, seed=77, stream_id=9, plane_offset=12, partials=p1)
    want = hl.power_irfft2(z, filt, shape)
    # the generate path takes the filter value under the radius' square root (|z f| = sqrt(-ln2 f^2 log2 u), one rounding), the replay
    # multiplies the dumped z by f (three): the two agree to the last bits of every spectrum value, not bit for bit (round 5)
    close(got, want, rtol=0, atol=GEN_VS_REPLAY_ATOL)
    ref = torch.fft.irfft2(z.cpu() * filt.cpu(), s=(h, w), norm="ortho")
    close(got, ref, rtol=0, atol=FFT_ATOL)
    tot = hl.stats_finalize(p1, got.numel()).cpu()
    assert abs(tot[1].item() - (got.double() ** 2).sum().item()) < 1e-5 * tot[1].item()
    # shard invariance: planes [12, 32) drawn as [12, 20) + [20, 32)
    a = hl.power_spectrum((2, 4, h, w), "cuda", seed=77, stream_id=9, plane_offset=12)
    b = hl.power_spectrum((3, 4, h, w), "cuda", seed=77, stream_id=9, plane_offset=20)
    assert torch.equal(torch.cat((a, b)), z)
    # 600 latents = 600 RNG groups (> 512): one workgroup per GROUP; its first two latents alone: one workgroup per PLANE
    # (fast-forwarding the group streams) -- same values
    big = hl.power_spectrum((600, 4, 32, 32), "cuda", seed=9, stream_id=1)
    assert torch.equal(hl.power_spectrum((2, 4, 32, 32), "cuda", seed=9, stream_id=1), big[:2])
    assert torch.equal(hl.power_spectrum((3, 4, 32, 32), "cuda", seed=9, stream_id=1, plane_offset=4 * 597), big[597:])
    # channel counts that are not a multiple of 4 use per-plane streams (group 1), still shard-invariant
    z3 = hl.power_spectrum((4, 3, 32, 32), "cuda", seed=1, stream_id=0, plane_offset=3)
    a3 = hl.power_spectrum((1, 3, 32, 32), "cuda", seed=1, stream_id=0, plane_offset=3)
    b3 = hl.power_spectrum((3, 3, 32, 32), "cuda", seed=1, stream_id=0, plane_offset=6)
    assert torch.equal(torch.cat((a3, b3)), z3)
    assert abs(z3.real.std().item() - math.sqrt(0.5)) < 1e-2
    # an offset that splits an RNG group is refused
    with pytest.raises(hl.SonarHipError):
        hl.power_spectrum((2, 4, h, w), "cuda", seed=77, stream_id=9, plane_offset=2)


def test_power_spectrum_draws_are_unit_complex_normals(hl):
    """Generate-mode spectrum elements z = rho e^{i theta} (23-bit radius word, 16-bit angle word, multiply-with-carry streams
    seeded by Philox): first moments, E|z|^2 = 1, E|z|^4 = 2, uncorrelated parts, uniform angle, no correlation between
    neighbouring elements / planes / the two halves of an angle word, and the expected tail."""
    z = hl.power_spectrum((64, 4, 128, 128), "cuda", seed=2024, stream_id=3).to(torch.complex128)  # 2.1 M elements
    n = z.numel()
    re, im = z.real.flatten(), z.imag.flatten()
    tol = 5.0 / math.sqrt(n)  # 5 sigma of a unit-variance mean
    assert abs(re.mean().item()) < tol and abs(im.mean().item()) < tol
    assert abs(re.var().item() - 0.5) < 2 * tol and abs(im.var().item() - 0.5) < 2 * tol
    assert abs((re * im).mean().item()) < tol
    a2 = re * re + im * im
    assert abs(a2.mean().item() - 1.0) < 2 * tol            # exponential(1): mean 1
    assert abs((a2 * a2).mean().item() - 2.0) < 10 * tol     # second moment 2
    assert abs(((re ** 4).mean() / re.var() ** 2).item() - 3.0) < 20 * tol  # Gaussian kurtosis of a component
    ang = torch.atan2(im, re)
    hist = torch.histc(ang, bins=64, min=-math.pi, max=math.pi) / n * 64
    assert (hist - 1).abs().max().item() < 6 * math.sqrt(64 / n)
    for shift in (1, 65, 128 * 65):  # next column, next row, next plane
        assert abs((re[:-shift] * re[shift:]).mean().item()) < tol and abs((re[:-shift] * im[shift:]).mean().item()) < tol
    half = z.shape[-2] // 2  # partner rows ky and ky + H/2 share one angle word (low / high half)
    top, bot = z[..., :half, 1:64], z[..., half:, 1:64]
    assert abs((top.real * bot.real).mean().item()) < 2 * tol and abs((top.imag * bot.imag).mean().item()) < 2 * tol
    assert a2.max().item() < 23 * math.log(2) + 1e-6 and a2.max().item() > 11.0  # rho^2 <= 23 ln 2; 2.1 M draws reach ~14.5


@pytest.mark.parametrize("hw", [(128, 128), (64, 64), (32, 64)])
@pytest.mark.parametrize("factor", [1.0, 0.6])
def test_power_noise_fused_normalisation(hl, hw, factor):
    """sonar_power_noise_f32 (Parseval statistics + one write) == generate, then scale_noise (two sweeps)."""
    h, w = hw
    shape = (6, 4, h, w)
    torch.manual_seed(3)
    filt = dev(torch.rand(h, w // 2 + 1) * 1.5 + 0.1)
    part = hl.new_partials("cuda")
    two_pass = hl.power_irfft2(None, filt, shape, seed=5, stream_id=2, plane_offset=8, partials=part)
    actual = hl.stats_finalize(part, two_pass.numel()).cpu()
    hl.scale_noise_(two_pass, factor, True, part)
    fused = hl.power_noise(filt, shape, seed=5, stream_id=2, plane_offset=8, factor=factor)
    close(fused, two_pass, rtol=2e-5, atol=2e-5)
    assert abs(fused.std().item() - factor) < 2e-4 * factor + 1e-5
    # the Parseval statistics themselves
    ws = hl.new_partials("cuda")
    hl._check(hl.load().sonar_power_noise_f32(filt.data_ptr(), fused.data_ptr(), 24, h, w, 5, 2, 8, 4, 1.0, 2.5, ws.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "power_noise")
    pars = hl.stats_finalize(ws, fused.numel()).cpu()
    assert abs(pars[1].item() - actual[1].item()) < 2e-5 * actual[1].item()
    assert abs(pars[0].item() - actual[0].item()) < 1e-3 * math.sqrt(actual[1].item())


@pytest.mark.parametrize("factor", [1.0, 1.7])
def test_perlin_and_pyramid_fused_normalisation(hl, factor):
    shape = (4, 4, 32, 32)
    torch.manual_seed(0)
    terms = hl.perlin_terms(dev(torch.rand(2, 4, 33, 33) * 2 * math.pi))
    part = hl.new_partials("cuda")
    two = hl.perlin_generate(shape, terms, 2.0, seed=11, stream_id=4, elem_offset=4096 * 3, partials=part)
    hl.scale_noise_(two, factor, True, part)
    one = hl.perlin_noise(shape, terms, 2.0, 11, 4, 4096 * 3, factor)
    close(one, two, rtol=1e-5, atol=1e-6)
    small = [hl.philox_normal((16, hh, ww), "cuda", 42, 10 + i) for i, (hh, ww) in enumerate(((9, 9), (2, 2)))]
    levels = [(None, 32, 32, 1.0), (small[0], 9, 9, 0.7), (small[1], 2, 2, 0.49)]
    part = hl.new_partials("cuda")
    two = hl.pyramid_generate(shape, "cuda", levels, "bilinear", 42, 0, 4096, part)
    hl.scale_noise_(two, factor, True, part)
    one = hl.pyramid_noise(shape, "cuda", levels, "bilinear", 42, 0, 4096, factor)
    close(one, two, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("planes", [(3, 4), (64, 4), (300, 4)])
def test_pyramid_fused_call_reduces_only_the_partials_its_planes_own(hl, planes):
    """sonar_pyramid_noise_f32 hands its normalising pass the number of partial pairs the plane kernel's workgroups own (12 / 256 / 1024
    here) instead of all SONAR_NPART: the rest are zeros, so the decision and the values are the two-step path's bit for bit
    (generate with statistics, then sonar_scale_noise_f32 over every pair) -- also with levels drawn in the kernel."""
    b, c = planes
    shape = (b, c, 64, 64)  # whole RNG tiles per plane: the plane kernel runs
    n = b * c
    small = [hl.philox_normal((n, hh, ww), "cuda", 42, 10 + i) for i, (hh, ww) in enumerate(((21, 21), (5, 5)))]
    for levels in ([(None, 64, 64, 1.0), (small[0], 21, 21, 0.7), (small[1], 5, 5, 0.49)], hl.AutoLevels(64, 64, 10, 0.7, 42, 3)):
        part = hl.new_partials("cuda")
        two = hl.pyramid_generate(shape, "cuda", levels, "bilinear", 42, 3, 4096 * 8, part)
        hl.scale_noise_(two, 1.3, True, part)
        one = hl.pyramid_noise(shape, "cuda", levels, "bilinear", 42, 3, 4096 * 8, 1.3)
        assert torch.equal(one, two)
        assert abs(one.std().item() - 1.3) < 1e-3


GENERAL_PLANES = [(104, 152), (152, 104), (96, 96), (72, 88), (90, 50), (34, 38), (20, 12), (2, 4), (168, 96), (128, 160), (192, 192), (512, 64), (26, 1024),
                  # register codelets of every length 2..16 in either pass (round 3): 11 x 2 / 13 x 2, 15 x 2 / 14, 9 x 2 / 10, 12 x 2 / 9 x 2,
                  # 12 x 12 / 8 x 7, 12 x 10 / 10 x 8, 11 x 8 / 13 x 4, 16 x 11 / 8 x 5, 14 x 9 / 9 x 5, one-pass lengths, 16 x 15 / 15 x 2
                  (22, 26), (30, 28), (18, 20), (24, 36), (144, 112), (120, 160), (88, 104), (176, 80), (126, 90), (14, 10), (6, 12), (240, 60)]


@pytest.mark.parametrize("hw", GENERAL_PLANES)
def test_power_irfft2_general_size_planes(hl, hw):
    """Latents that are not powers of two (832 x 1216 px -> 104 x 152 ...): the general-size kernels (two-factor table-twiddle DFTs in
    LDS; 34 x 38 has the prime factors 17 and 19) against torch.fft on the host -- replay, forward + filter + inverse, unit filter."""
    torch.manual_seed(23)
    h, w = hw
    assert hl.load().sonar_power_plane_kind(h, w) == 2
    z = torch.randn(3, 2, h, w // 2 + 1, dtype=torch.complex64)
    filt = torch.rand(h, w // 2 + 1) + 0.5
    want = torch.fft.irfft2(z * filt, s=(h, w), norm="ortho")
    part = hl.new_partials("cuda")
    got = hl.power_irfft2(dev(z), dev(filt), (3, 2, h, w), partials=part)
    close(got, want, rtol=0, atol=FFT_ATOL)
    tot = hl.stats_finalize(part, got.numel()).cpu()
    assert abs(tot[1].item() - (got.double() ** 2).sum().item()) < 1e-5 * tot[1].item() + 1e-9
    x = torch.randn(3, 2, h, w)
    close(hl.spectral_filter(dev(x), dev(filt)), orc.spectral_filter(x, filt), rtol=0, atol=FFT_ATOL)
    close(hl.spectral_filter(dev(x), dev(torch.ones(h, w // 2 + 1))), x, rtol=0, atol=FFT_ATOL)


@pytest.mark.parametrize("hw", [(104, 152), (96, 96), (34, 38), (90, 50)])
def test_power_generate_general_size_planes(hl, hw):
    """Generate mode on general-size planes: equals the replay of its own dumped spectrum, unit complex normals, shard invariant,
    fused Parseval normalisation == generate + scale_noise."""
    h, w = hw
    shape = (5, 4, h, w)
    filt = dev(torch.rand(h, w // 2 + 1) + 0.25)
    z = hl.power_spectrum(shape, "cuda", seed=77, stream_id=9, plane_offset=12)
    assert abs(z.real.std().item() - math.sqrt(0.5)) < 6e-3 and abs(z.imag.mean().item()) < 1e-2
    assert abs((z.abs() ** 2).mean().item() - 1.0) < 1.5e-2
    p1 = hl.new_partials("cuda")
    got = hl.power_irfft2(None, filt, shape, seed=77, stream_id=9, plane_offset=12, partials=p1)
    close(got, hl.power_irfft2(z, filt, shape), rtol=0, atol=GEN_VS_REPLAY_ATOL)  # last-bit differences per spectrum value (see above)
    close(got, torch.fft.irfft2(z.cpu() * filt.cpu(), s=(h, w), norm="ortho"), rtol=0, atol=FFT_ATOL)
    a = hl.power_spectrum((2, 4, h, w), "cuda", seed=77, stream_id=9, plane_offset=12)
    b = hl.power_spectrum((3, 4, h, w), "cuda", seed=77, stream_id=9, plane_offset=20)
    assert torch.equal(torch.cat((a, b)), z)
    big = hl.power_spectrum((300, 4, h, w), "cuda", seed=9, stream_id=1) if h * w < 5000 else None  # whole-group units vs per-plane units
    if big is not None:
        assert torch.equal(hl.power_spectrum((2, 4, h, w), "cuda", seed=9, stream_id=1), big[:2])
    for factor in (1.0, 0.6):
        part = hl.new_partials("cuda")
        two_pass = hl.power_irfft2(None, filt, shape, seed=5, stream_id=2, plane_offset=8, partials=part)
        hl.scale_noise_(two_pass, factor, True, part)
        fused = hl.power_noise(filt, shape, seed=5, stream_id=2, plane_offset=8, factor=factor)
        close(fused, two_pass, rtol=2e-5, atol=2e-5)
        assert abs(fused.std().item() - factor) < 3e-4


@pytest.mark.parametrize("latents", [530, 519, 1030])
def test_power_general_size_mixed_units_draw_the_same_planes(hl, latents):
    """Round 5: with more RNG groups than resident workgroups the general-size plane kernel takes whole groups for its full rounds and the
    planes of the remaining groups one by one (csrc/power_core.h, group_units / GroupWalk) -- a work decomposition, not part of the stream
    definition: the planes are the ones a small launch at the same offsets draws (per-plane units), the ones the spectrum dump replays."""
    h, w = 104, 152
    filt = dev(torch.rand(h, w // 2 + 1) + 0.25)
    shape = (latents, 4, h, w)
    got = hl.power_irfft2(None, filt, shape, seed=21, stream_id=4)
    for first, count in ((0, 3), (511, 2), (latents - 4, 4), (latents - 1, 1)):
        part = hl.power_irfft2(None, filt, (count, 4, h, w), seed=21, stream_id=4, plane_offset=4 * first)
        assert torch.equal(part, got[first : first + count]), first
    z = hl.power_spectrum((6, 4, h, w), "cuda", seed=21, stream_id=4, plane_offset=4 * (latents - 6))
    close(got[latents - 6 :], hl.power_irfft2(z, filt, (6, 4, h, w)), rtol=0, atol=GEN_VS_REPLAY_ATOL)
    # the statistics variant of the same launch: the same planes, partials = the tensor's sums
    p1 = hl.new_partials("cuda")
    assert torch.equal(hl.power_irfft2(None, filt, shape, seed=21, stream_id=4, partials=p1), got)
    tot = hl.stats_finalize(p1, got.numel()).cpu()
    assert abs(tot[0].item() - got.double().sum().item()) < 1e-6 * got.numel() and abs(tot[1].item() / (got.double() ** 2).sum().item() - 1.0) < 1e-6
    # the normalised call (statistics kernel with its own units + the final pass) has unit variance over the whole tensor
    out = hl.power_noise(filt, shape, seed=21, stream_id=4, plane_offset=0, factor=1.0)
    assert abs(out.double().std().item() - 1.0) < 1e-4 and abs(out.double().mean().item()) < 1e-4


def test_power_general_size_full_batch(hl):
    """512 SDXL-portrait latents (104 x 152): normalised generation has unit variance, zero mean, flat-filter output is white."""
    shape = (512, 4, 104, 152)
    filt = dev(torch.ones(104, 77))
    out = hl.power_noise(filt, shape, seed=3, stream_id=0, plane_offset=0, factor=1.0)
    assert abs(out.std().item() - 1.0) < 1e-4 and abs(out.mean().item()) < 1e-3
    row_corr = (out[:, :, :, 1:] * out[:, :, :, :-1]).mean().item()
    col_corr = (out[:, :, 1:, :] * out[:, :, :-1, :]).mean().item()
    assert abs(row_corr) < 2e-3 and abs(col_corr) < 2e-3


def test_planes_beyond_the_lds_kernels_take_the_direct_passes(hl):
    """Odd sizes and half-spectra larger than LDS are not the LDS kernels' (kind 0 at the C ABI); the host routes them through the
    direct DFT passes (kind 3), up to 2048 x 2048; beyond that it raises."""
    assert hl.load().sonar_power_plane_kind(96, 161) == 0 and hl.load().sonar_power_plane_kind(256, 256) == 4  # 4: generated in column blocks
    assert hl.power_plane_kind(95, 160) == 3 and hl.power_plane_kind(256, 256) == 4 and hl.power_plane_kind(128, 128) == 1
    assert hl.power_plane_kind(1024, 256) == 3  # a block of columns of that height does not fit either
    assert hl.power_plane_kind(4096, 64) == 0 and not hl.power_supported(64, 4096)
    out = hl.power_irfft2(None, dev(torch.ones(95, 81)), (1, 4, 95, 160), seed=3, stream_id=1)  # odd height
    assert tuple(out.shape) == (1, 4, 95, 160) and bool(torch.isfinite(out).all()) and abs(out.std().item() - 1.0) < 0.02
    z = torch.randn(2, 256, 129, dtype=torch.complex64, device="cuda")
    filt = torch.rand(256, 129, device="cuda") + 0.5
    got = hl.power_irfft2(z, filt, (2, 1, 256, 256)).reshape(2, 256, 256)  # half-spectrum larger than LDS
    want = torch.fft.irfft2(z * filt, s=(256, 256), norm="ortho")
    torch.testing.assert_close(got, want, rtol=0, atol=2e-5 * float(want.abs().max()))
    with pytest.raises(hl.SonarHipError):
        hl.power_irfft2(None, dev(torch.ones(4096, 33)), (1, 1, 4096, 64))


def test_channel_mix(hl, golden):
    g = golden("power_filter")
    torch.manual_seed(14)
    x = torch.randn(3, 4, 8, 8)
    for key in ("mixer_0.25", "mixer_-0.2", "mixer_partial"):
        m = g[key]
        want = (m @ x.swapaxes(0, 1).reshape(4, -1)).reshape(4, 3, 8, 8).swapaxes(1, 0)
        close(hl.channel_mix(dev(x), dev(m)), want, rtol=1e-5, atol=2e-6)


def test_cpu_tensor_is_rejected(hl):
    with pytest.raises(hl.SonarHipError):
        hl.stats(torch.zeros(16))


# ------------------------------------------------------------------------------------------------ C-ABI error behaviour
def test_c_abi_error_codes_and_messages(hl):
    """Every entry point returns 0 or a negative SONAR_ERR_* code and leaves a message in sonar_last_error(); nothing is
    launched on a refused call (the output buffer keeps its contents)."""
    lib = hl.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.full((64,), 7.0, device="cuda")
    assert lib.sonar_stats_f32(None, 64, None, st) == hl.ERR_ARG
    assert b"sonar_stats_f32" in lib.sonar_last_error()
    filt = torch.ones(25 * 13, device="cuda")
    out = torch.full((2, 25, 24), 7.0, device="cuda")
    part = hl.new_partials("cuda")
    rc = lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), 2, 25, 24, 1, 0, 0, 1, part.data_ptr(), st)  # odd height
    assert rc == hl.ERR_UNSUPPORTED and b"unsupported plane 25 x 24" in lib.sonar_last_error()
    rc = lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), 6, 32, 32, 1, 0, 2, 4, part.data_ptr(), st)
    assert rc == hl.ERR_ARG and b"multiples of the RNG group" in lib.sonar_last_error()
    assert lib.sonar_dwt2_ws_bytes(1, 8, 8, 8, 9, 4, 0) == -1  # mode out of range
    assert lib.sonar_wcfg_fused_ws_bytes(4, 64, 64, 13, 8, 1, 8, 1, 8) == -1  # more levels than supported
    torch.cuda.synchronize()
    assert torch.all(out == 7.0) and torch.all(x == 7.0)
    with pytest.raises(hl.SonarHipError, match="code -2"):
        hl.power_irfft2(None, torch.ones(4096, 13, device="cuda"), (2, 1, 4096, 24))  # beyond the direct passes too (lines of at most 2048)
    with pytest.raises(hl.SonarHipError):
        hl.stats(torch.zeros(8))  # host tensor


# ------------------------------------------------------------------------------------------------ Perlin lattice drawn in-kernel
def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Reference Philox4x32-10 on numpy uint64 lanes (Salmon et al.); returns the four 32-bit words."""
    import numpy as np

    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    mask = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & mask for v in (c0, c1, c2, c3))
    for r in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        kk0, kk1 = np.uint64((k0 + r * W0) & 0xFFFFFFFF), np.uint64((k1 + r * W1) & 0xFFFFFFFF)
        c0, c1, c2, c3 = (hi1 ^ c1 ^ kk0) & mask, lo1, (hi0 ^ c3 ^ kk1) & mask, lo0
    return c0, c1, c2, c3


@pytest.mark.parametrize("blend_mode", ["lerp", "inject"])
def test_perlin_lattice_kernel_against_a_host_replay_of_its_draws(hl, blend_mode):
    """sonar_perlin_lattice_f32 draws angle(it, c, gy, gx) = 2 pi u, u = word[it % 4] >> 8 * 2^-24 of
    Philox4x32-10(counter = (lattice point lo, hi, it // 4, stream), key = seed): replayed here with numpy, then the oracle's term."""
    import numpy as np

    C, H, W, iters, seed, stream = 2, 5, 7, 6, 0x1234567890ABCDEF, 9
    got = hl.perlin_lattice(iters, C, H, W, "cuda", blend_mode, seed, stream).cpu()[0]
    pts = np.arange(C * (H + 1) * (W + 1), dtype=np.uint64)
    angles = np.zeros((iters, C, H + 1, W + 1), dtype=np.float32)
    for g in range(0, iters, 4):
        words = _philox4x32_10(pts & np.uint64(0xFFFFFFFF), pts >> np.uint64(32), np.full_like(pts, g // 4), np.full_like(pts, stream),
                               seed & 0xFFFFFFFF, seed >> 32)
        for k in range(4):
            if g + k < iters:
                u = (words[k] >> np.uint64(8)).astype(np.float32) * np.float32(2.0**-24)
                angles[g + k] = (u * np.float32(2 * math.pi)).reshape(C, H + 1, W + 1)
    want = sum(orc.perlin_term(torch.from_numpy(angles[i]), blend_mode) for i in range(iters))
    close(got, want, rtol=0, atol=3e-6 * iters)
    # statistics of a big lattice: the term of one iteration has zero mean
    big = hl.perlin_lattice(1, 4, 128, 128, "cuda", "lerp", 7, 0)
    assert abs(big.mean().item()) < 5e-3 and big.std().item() > 0.05


def test_pyramid_in_kernel_level_grids(hl):
    """Small level grids drawn by the plane kernel (NULL pointer, size below the latent's): deterministic, shard-invariant,
    right variance; shapes the plane kernel cannot run report 'unsupported' (None) so the host passes explicit grids."""
    shape = (6, 4, 64, 64)
    levels = [(None, 64, 64, 1.0), (None, 21, 20, 0.7), (None, 3, 3, 0.49)]
    a = hl.pyramid_generate(shape, "cuda", levels, "bilinear", 11, 5)
    assert torch.equal(a, hl.pyramid_generate(shape, "cuda", levels, "bilinear", 11, 5))
    tail = hl.pyramid_generate((4, 4, 64, 64), "cuda", levels, "bilinear", 11, 5, 2 * 4 * 64 * 64)
    assert torch.equal(tail, a[2:])
    other = hl.pyramid_generate(shape, "cuda", levels, "bilinear", 12, 5)
    assert abs((other * a).mean().item()) < 0.02
    # variance: 1 + 1 (folded full-resolution level) + the interpolated small levels' share (< their weights squared)
    v = a.double().var().item()
    assert 2.0 < v < 2.0 + 0.49 + 0.24 + 0.05
    # coarse structure really is there: 8x8 block means carry more variance than white noise of variance v would (v / 64)
    blocks = a.reshape(6, 4, 8, 8, 8, 8).mean(dim=(3, 5))
    assert blocks.double().var().item() > 3 * v / 64
    # planes that do not fill an RNG tile (32 x 32) or straddle tiles (104 x 152) run in the plane kernel too, shard-invariant
    for hh, ww in ((32, 32), (104, 152)):
        lv = [(None, hh, ww, 1.0), (None, hh // 3, ww // 3, 0.7)]
        b = hl.pyramid_generate((3, 4, hh, ww), "cuda", lv, "bilinear", 11, 5)
        assert torch.equal(hl.pyramid_generate((2, 4, hh, ww), "cuda", lv, "bilinear", 11, 5, 4 * hh * ww), b[1:])
        assert 2.0 < b.double().var().item() < 2.6
    assert hl.pyramid_generate((2, 4, 32, 30), "cuda", [(None, 32, 30, 1.0), (None, 9, 9, 0.7)], "bilinear", 11, 5) is None  # W % 4 != 0


@pytest.mark.parametrize("uniform", [False, True])
@pytest.mark.parametrize("n,offset", [(8 * 4 * 64 * 64, 0), (4096 * 3 + 5, 4096 * 2), (1000, 12)])
def test_fused_normalised_fill(hl, uniform, n, offset):
    """sonar_philox_noise_f32 (statistics by re-drawing, one write) == the plain fill followed by stats + scale_noise."""
    kw = dict(sub=0.5, mul=3.46, add=0.1) if uniform else {}
    for factor in (1.0, 0.7):
        part = hl.new_partials("cuda")
        two = hl.philox_uniform((n,), "cuda", 5, 3, offset, partials=part, **kw) if uniform else hl.philox_normal((n,), "cuda", 5, 3, offset, part)
        hl.scale_noise_(two, factor, True, part)
        one = hl.philox_noise(uniform, (n,), "cuda", 5, 3, offset, factor, **kw)
        close(one, two, rtol=1e-5, atol=1e-6)
        assert abs(one.std().item() - factor) < (2.5 / math.sqrt(n) + 5e-3) * factor  # inside the band nothing is rescaled


@pytest.mark.parametrize("shape", [(3, 9, 7), (2, 135, 30), (5, 16, 21), (2, 33, 64), (1, 1, 5), (2, 6, 1), (1, 250, 250),
                                   # even widths run the LDS line transforms pass by pass (round 3): 16 x 16 codelets, a direct second pass (512 = 16 x
                                   # 32), the longest lines either way, a length with the factors 13 and 20
                                   (1, 256, 256), (2, 384, 512), (1, 2048, 40), (1, 34, 2048), (1, 300, 520),
                                   # odd widths as full-length complex lines: 135 = 15 x 9, a prime, a prime above the codelets, the longest
                                   (3, 135, 135), (2, 24, 241), (2, 16, 37), (1, 6, 2047)])
def test_direct_dft_passes_match_torch_fft(hl, shape):
    """sonar_dft_rows_r2c / cols / rows_c2r (the route of planes the LDS FFT kernels do not take: odd sizes, big planes) against
    torch.fft: rfft2, irfft2 of a filtered spectrum (norm='ortho', imaginary parts of the DC / Nyquist columns ignored), and the
    spectral filter irfft2(rfft2(x) * f).  fp32 sums: 2e-5 of the result's peak."""
    planes, H, W = shape
    K = W // 2 + 1
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(planes, H, W, device="cuda", generator=g)
    filt = torch.rand(H, K, device="cuda", generator=g) + 0.5
    assert hl.power_plane_kind(H, W) in (3, 4)  # 4: only GENERATED noise leaves the direct passes (test_gpu_round4.py)
    lib = hl.load()
    a = torch.empty(planes, H, K, dtype=torch.complex64, device="cuda")
    b = torch.empty_like(a)
    assert lib.sonar_dft_rows_r2c_f32(x.data_ptr(), a.data_ptr(), planes * H, W, None) == 0
    assert lib.sonar_dft_cols_f32(a.data_ptr(), None, b.data_ptr(), planes, H, K, 0, None) == 0
    want = torch.fft.rfft2(x)
    torch.testing.assert_close(b, want, rtol=0, atol=2e-5 * float(want.abs().max()))
    z = torch.randn(planes, H, K, dtype=torch.complex64, device="cuda", generator=g)
    part = hl.new_partials("cuda")
    got = hl.power_irfft2(z, filt, (planes, 1, H, W), partials=part).reshape(planes, H, W)
    want = torch.fft.irfft2(z * filt, s=(H, W), norm="ortho")
    torch.testing.assert_close(got, want, rtol=0, atol=2e-5 * max(1.0, float(want.abs().max())))
    sums = part.view(-1, 2).sum(0)
    torch.testing.assert_close(sums, torch.stack([got.double().sum(), (got.double() ** 2).sum()]), rtol=1e-9, atol=1e-6)
    got = hl.spectral_filter(x, filt)
    want = torch.fft.irfft2(torch.fft.rfft2(x, norm="ortho") * filt, s=(H, W), norm="ortho")
    torch.testing.assert_close(got, want, rtol=0, atol=2e-5 * max(1.0, float(want.abs().max())))


def test_power_noise_on_an_odd_plane(pkg, hl):
    """1080-line video: 135 x 240 latents.  Generate mode takes the direct passes (white noise -> rfft2 x filter -> irfft2, the
    reference's own route): unit statistics after normalisation, two shards == the whole."""
    import importlib

    pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
    item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                             common_mode=0.0, channel_correlation="1")

    def run(b0, b, normalized):
        torch.manual_seed(4)
        with ng.shard_offset(b0):
            x = torch.zeros(b, 4, 135, 240, device="cuda")
            return item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=normalized)(None, None)

    whole = run(0, 4, False)
    assert torch.equal(torch.cat([run(0, 2, False), run(2, 2, False)]), whole)
    normed = run(0, 4, True)
    assert bool(torch.isfinite(normed).all()) and abs(normed.std().item() - 1.0) < 1e-3 and abs(normed.mean().item()) < 5e-3
    # pink: neighbouring pixels correlate
    assert float((whole[..., 1:] * whole[..., :-1]).mean() / whole.var()) > 0.2
