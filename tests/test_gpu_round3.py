"""-m gpu, round 3: the pipelined power-noise kernel and its look-ahead statistics."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(pkg):
    import types

    pkg.hip_lib.load()
    return types.SimpleNamespace(hl=pkg.hip_lib, powernoise=importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise"),
                                 noise=importlib.import_module("comfyui_sonar_amd.py.noise"),
                                 noise_generation=importlib.import_module("comfyui_sonar_amd.py.noise_generation"),
                                 utils=importlib.import_module("comfyui_sonar_amd.py.utils"))


def _item(pn, factor=1.0, channels="1,1,1,1,1,1"):
    return pn.PowerNoiseItem(factor, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                             common_mode=0.0, channel_correlation=channels)


def _filter(api, shape):
    return _item(api.powernoise).make_filter(shape).to("cuda", torch.float32).reshape(shape[-2], shape[-1] // 2 + 1).contiguous()


@pytest.mark.parametrize("shape", [(512, 4, 128, 128), (96, 4, 128, 128), (171, 3, 128, 128), (65, 4, 128, 128), (300, 4, 128, 128)])
def test_pipelined_power_kernel_equals_the_phase_serial_kernel(api, shape):
    """The pipelined generate kernel (drawing team / transforming team, planes > 256) and the phase-serial kernel serve ONE stream
    definition: run on the same (seed, stream) -- sonar_power_pipeline(0) selects the phase-serial kernel for every batch size -- they
    write the same bits, raw, with statistics, and normalised.  The phase-serial kernel itself is pinned to the spectrum the streams dump
    and torch's irfft2 (sampled planes)."""
    hl = api.hl
    lib = hl.load()
    filt = _filter(api, shape)
    planes = shape[0] * shape[1]

    def run_all():
        part = hl.new_partials("cuda")
        return (hl.power_irfft2(None, filt, shape, seed=11, stream_id=5, plane_offset=0),
                hl.power_irfft2(None, filt, shape, seed=11, stream_id=6, plane_offset=0, partials=part), part,
                hl.power_noise(filt, shape, seed=11, stream_id=5, plane_offset=0, factor=0.75))

    assert lib.sonar_power_pipeline(-1) == 1
    piped = run_all()
    assert lib.sonar_power_pipeline(0) == 1
    try:
        # beyond 256 work units the look-ahead belongs to the pipelined kernel (the phase-serial kernel has its own below that)
        assert lib.sonar_power_noise_ahead_ok(planes, 128, 128, hl.rng_group_for(shape)) == 0
        serial = run_all()
    finally:
        lib.sonar_power_pipeline(1)
    raw, _, part, norm = piped
    assert torch.equal(raw, serial[0]) and torch.equal(piped[1], serial[1]) and torch.equal(norm, serial[3])
    # the statistics partials are reduced in each kernel's own grouping: the same totals to fp64 rounding
    torch.testing.assert_close(part.view(-1, 2).sum(0), serial[2].view(-1, 2).sum(0), rtol=1e-12, atol=0)
    spec = hl.power_spectrum(shape, "cuda", seed=11, stream_id=5)
    for pl in (0, 1, 255, 256, planes // 2 + 3, planes - 1):
        z = spec.reshape(planes, 128, 65)[pl] * filt
        want = torch.fft.irfft2(z, s=(128, 128), norm="ortho")
        got = serial[0].reshape(planes, 128, 128)[pl]
        assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item(), pl
    d = raw.double()
    want = ((d - d.mean()) / d.std()).float() * 0.75
    torch.testing.assert_close(norm, want, rtol=2e-5, atol=2e-6)
    # two shards of the batch are the batch (stream keys are global plane groups; only for latent counts a shard cut can halve)
    if shape[0] % 2 == 0:
        half = shape[0] // 2
        a = hl.power_irfft2(None, filt, (half, *shape[1:]), seed=11, stream_id=5, plane_offset=0)
        b = hl.power_irfft2(None, filt, (half, *shape[1:]), seed=11, stream_id=5, plane_offset=half * shape[1])
        assert torch.equal(torch.cat([a, b]), raw)


def test_rng_groups_beyond_the_lookahead(api):
    """The C ABI takes any RNG group up to 8 planes (the host passes 1 or 4): the look-ahead statistics cover four planes per unit, so
    larger unsplit groups must be refused by sonar_power_noise_ahead_ok / _ahead_f32 and served by the two-launch form, whose
    statistics cover every plane; beyond 8 planes per group nothing is launched."""
    hl = api.hl
    lib = hl.load()
    shape = (256, 8, 128, 128)
    planes = shape[0] * shape[1]
    filt = _filter(api, shape)
    assert lib.sonar_power_noise_ahead_ok(planes, 128, 128, 4) == 1      # 512 units of four planes
    assert lib.sonar_power_noise_ahead_ok(planes, 128, 128, 8) == 0      # 256 units of eight planes: beyond the look-ahead's four
    assert lib.sonar_power_noise_ahead_ok(64 * 8, 128, 128, 8) == 1      # 64 groups: single planes as units, any group size
    out = torch.empty(shape, device="cuda")
    ws, nws = hl.new_partials("cuda"), hl.new_partials("cuda")
    st = hl._stream()
    rc = lib.sonar_power_noise_ahead_f32(filt.data_ptr(), out.data_ptr(), planes, 128, 128, 3, 9, 0, 8, 1.0, 2.5, ws.data_ptr(), 0, 10, nws.data_ptr(), st)
    assert rc == hl.ERR_UNSUPPORTED
    raw = torch.empty(shape, device="cuda")
    assert lib.sonar_power_irfft2_f32(None, filt.data_ptr(), raw.data_ptr(), planes, 128, 128, 3, 9, 0, 8, None, st) == 0
    assert lib.sonar_power_noise_f32(filt.data_ptr(), out.data_ptr(), planes, 128, 128, 3, 9, 0, 8, 1.0, 2.5, ws.data_ptr(), st) == 0
    d = raw.double()
    torch.testing.assert_close(out, ((d - d.mean()) / d.std()).float(), rtol=2e-5, atol=2e-6)
    assert abs(out.std().item() - 1.0) < 1e-4
    assert lib.sonar_power_noise_f32(filt.data_ptr(), out.data_ptr(), planes, 128, 128, 3, 9, 0, 16, 1.0, 2.5, ws.data_ptr(), st) == hl.ERR_UNSUPPORTED


@pytest.mark.parametrize("shape", [(512, 4, 128, 128), (128, 4, 128, 128), (100, 3, 128, 128),
                                   # launch-bound batches: the phase-serial kernel, the next call's statistics in extra workgroups of its launch
                                   (64, 4, 128, 128), (1, 4, 128, 128), (16, 3, 128, 128), (5, 16, 128, 128), (8, 4, 64, 64), (3, 4, 128, 64), (2, 4, 32, 32),
                                   # general-size planes (SDXL's portrait buckets ...) at launch-bound sizes: the same in power_irfft2_any_kernel
                                   (4, 4, 104, 152), (1, 4, 96, 96), (8, 4, 72, 120), (3, 3, 112, 144), (1, 16, 80, 80)])
def test_lookahead_statistics_are_the_statistics_kernels(api, shape):
    """The statistics a call leaves for the next stream id == what the statistics launch of that call computes (same pairs, bit for bit:
    one unit per slot, the same order of additions), and a call that uses them writes the same tensor."""
    hl = api.hl
    lib = hl.load()
    filt = _filter(api, shape)
    planes = shape[0] * shape[1]
    H, W = shape[-2:]
    group = hl.rng_group_for(shape)
    assert lib.sonar_power_noise_ahead_ok(planes, H, W, group) == 1
    st = torch.cuda.current_stream().cuda_stream

    def call(stream_id, ws, have, nxt, nws):
        out = torch.empty(shape, device="cuda")
        rc = lib.sonar_power_noise_ahead_f32(filt.data_ptr(), out.data_ptr(), planes, H, W, 3, stream_id, 0, group, 1.0, 2.5, ws.data_ptr(), int(have),
                                             nxt, 0 if nws is None else nws.data_ptr(), st)
        assert rc == 0, lib.sonar_last_error()
        return out

    ws7, ws8, ws8b = (torch.full((2 * hl.NPART,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3))
    out7 = call(7, ws7, False, 8, ws8)           # statistics launch for stream 7, look-ahead for stream 8
    out8_plain = call(8, ws8b, False, 9, None)    # statistics launch for stream 8
    assert torch.equal(ws8, ws8b) and not bool(torch.isnan(ws8).any())
    out8_ahead = call(8, ws8, True, 9, None)      # no statistics launch: the look-ahead's pairs
    assert torch.equal(out8_ahead, out8_plain)
    assert torch.equal(out7, hl.power_noise(filt, shape, seed=3, stream_id=7, plane_offset=0, factor=1.0))


def test_lookahead_in_the_sampler_hits_and_misses(api):
    """A sampler called step after step predicts its next stream id (hit: no statistics launch); another device draw in between shifts
    the stream ids once (one miss; a one-off step is not adopted).  Every call equals the call without look-ahead."""
    pn, hl, ng = api.powernoise, api.hl, api.noise_generation
    x = torch.zeros(96, 4, 128, 128, device="cuda")
    sig = (torch.tensor(14.6), torch.tensor(10.0))
    states = []
    real = hl.PowerLookahead

    class Spy(real):
        def __init__(self):
            super().__init__()
            states.append(self)

    hl.PowerLookahead = Spy
    try:
        torch.manual_seed(21)
        ns = _item(pn).make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
        got = [ns(*sig).clone() for _ in range(3)]
        ng.DeviceRNG.take()                      # somebody else draws: the stream ids shift by one
        got += [ns(*sig).clone() for _ in range(2)]
    finally:
        hl.PowerLookahead = real
    look = states[-1]
    assert (look.hits, look.misses) == (3, 2)   # calls 2, 3 and 5 hit; call 1 (nothing known) and call 4 (shifted) miss
    torch.manual_seed(21)
    filt = _filter(api, tuple(x.shape))
    seed, first = ng.DeviceRNG.take(0)
    streams = [first, first + 1, first + 2, first + 4, first + 5]
    for g, stream in zip(got, streams):
        assert torch.equal(g, hl.power_noise(filt, tuple(x.shape), seed=seed, stream_id=stream, plane_offset=0, factor=1.0))


def test_cfg5_at_the_flux_latents_size_against_the_reference(api, golden):
    """BASELINE.json cfg5 at 2 x 16 x 128 x 128 against the REAL reference (tests/golden/make_golden.py gen_cfg5_full): the scheduled
    power-law + Perlin + third-source chain, normalised, through two SonarDPMPPSDE steps with momentum (four noise calls), replay mode.
    The third source is Gaussian (the configured Brownian one needs torchsde, absent from the image): chain arithmetic, scheduling and
    normalisation at the full size are pinned here, not only in the 32 x 32 miniature.  Compared: every fourth pixel of the result and
    of one noise call, and the fp64 plane sums / sums of squares of the whole tensors."""
    g = golden("cfg5_full")
    N, pn = api.noise, api.powernoise
    S = importlib.import_module("comfyui_sonar_amd.py.sonar")

    def fake_model(x, sigma, **_kw):  # tests/golden/make_golden.py fake_model
        s = sigma.reshape(-1, *([1] * (x.ndim - 1)))
        return x * 0.5 + torch.tanh(x) * (0.1 * s / (1.0 + s))

    inner = N.CustomNoiseChain()
    inner.add(_item(pn, 0.5, "1"))
    inner.add(N.CustomNoiseItem(0.3, noise_type="perlin"))
    inner.add(N.CustomNoiseItem(0.2, noise_type="gaussian"))
    fallback = N.CustomNoiseChain()
    fallback.add(N.CustomNoiseItem(1.0, noise_type="gaussian"))
    chain = N.CustomNoiseChain()
    chain.add(N.ScheduledNoise(1.0, noise=inner, start_sigma=20.0, end_sigma=0.0, normalize=None, fallback_noise=fallback))
    torch.manual_seed(71)
    x0 = (torch.randn(2, 16, 128, 128) * 10.0).cuda()
    sigmas = g["sigmas"]
    torch.manual_seed(72)
    ns = chain.make_noise_sampler(x0, sigmas[sigmas > 0].min(), sigmas.max(), seed=5, cpu=True, normalized=True)
    calls = []

    def spy(s, sn):
        out = ns(s, sn)
        calls.append(out.clone())
        return out

    out = S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas[:3], {"seed": 5}, None, True, None, dict(momentum=0.95), 1.0, 1.0, spy)
    assert len(calls) == 4
    for got, tag, tol in ((calls[1], "noise", 5e-5), (out, "out", 3e-4)):
        torch.testing.assert_close(got[..., ::4, ::4].cpu(), g[f"{tag}_sub"], rtol=tol, atol=tol)
        d = got.double()
        torch.testing.assert_close(d.sum(dim=(-2, -1)).cpu(), g[f"{tag}_plane_sums"], rtol=0, atol=128 * 128 * tol * 0.05)
        torch.testing.assert_close((d * d).sum(dim=(-2, -1)).cpu(), g[f"{tag}_plane_sq"], rtol=2 * tol, atol=0)


def test_reductions_over_dimensions_that_are_not_trailing(api, golden):
    """normalize_to_scale / scale_noise(normalize_dims=) / guidance_shift with any `dim` tuple (py/utils.py:97-99,452-470,
    py/sonar.py:372-377) against the reference: the row kernels on a transposed copy.  normalize_to_scale stays bit-exact (min / max and
    an elementwise rescale); the mean / std rows carry the usual 1e-5."""
    g = golden("nontrailing")
    U = api.utils
    S = importlib.import_module("comfyui_sonar_amd.py.sonar")
    t = g["t"].cuda()
    assert torch.equal(U.normalize_to_scale(t.clone(), -1.0, 1.0, dim=(1,)).cpu(), g["nts_c"])
    assert torch.equal(U.normalize_to_scale(t.clone(), 0.0, 2.0, dim=(0, 2)).cpu(), g["nts_bh"])
    tn = g["sn_in"].cuda()
    torch.testing.assert_close(U.scale_noise(tn.clone(), 0.8, normalize_dims=(1,)).cpu(), g["sn_c"], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(U.scale_noise(tn.clone(), 1.25, normalize_dims=(0, 3)).cpu(), g["sn_bw"], rtol=1e-5, atol=2e-6)
    ref = g["gs_ref"].cuda()
    torch.testing.assert_close(S.SonarGuidanceMixin.guidance_shift(tn.clone(), ref.clone(), dim=(1,)).cpu(), g["gs_c"], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(S.SonarGuidanceMixin.guidance_shift(tn.clone(), ref.clone(), dim=(0, 2, 3)).cpu(), g["gs_bhw"], rtol=1e-5, atol=2e-6)
    # trailing dims still take the direct route (no copy) and agree with the general one
    torch.testing.assert_close(U.scale_noise(tn.clone(), 0.8, normalize_dims=(-2, -1)), U.scale_noise(tn.clone(), 0.8, normalize_dims=(2, 3)))


def test_rfft2_on_any_plane_size(api):
    for shape in ((3, 32, 32), (2, 26, 38), (2, 9, 15), (1, 40, 56)):
        x = torch.randn(*shape, device="cuda")
        want = torch.fft.rfft2(x)
        got = api.hl.rfft2(x)
        assert (got - want).abs().max().item() <= 3e-5 * want.abs().max().item(), shape


def test_the_spectral_gain_is_uploaded_once_and_follows_its_parameters(api):
    """OneF / GreenTest build their gain with host arithmetic and a blocking upload: kept per generator, rebuilt when a parameter it
    depends on changes (the values of a call do not depend on whether the gain was cached)."""
    ng = api.noise_generation
    x = torch.zeros((2, 4, 64, 64), device="cuda")
    gen = ng.OneFNoiseGenerator(x, cpu=False, alpha=-0.5)
    g0 = gen.device_gain()
    assert gen.device_gain() is g0
    fresh = ng._half_gain(gen.spectral_gain().to(torch.float32)).to("cuda")
    assert torch.equal(g0, fresh)
    torch.manual_seed(5)
    first = gen.generate()  # cached gain
    torch.manual_seed(5)
    again = ng.OneFNoiseGenerator(x, cpu=False, alpha=-0.5).generate()  # fresh gain
    assert torch.equal(first, again)
    gen.alpha = 0.5
    g1 = gen.device_gain()
    assert g1 is not g0 and not torch.equal(g1, g0)
    assert torch.equal(g1, ng._half_gain(gen.spectral_gain().to(torch.float32)).to("cuda"))
    gen.update_x(torch.zeros((2, 4, 32, 64), device="cuda"))
    assert gen.device_gain().shape == (32, 33)


@pytest.mark.parametrize("rows,inner", [(4, 65536), (3, 8192), (5, 8190), (7, 1000), (2, 65537), (64, 16384)])
def test_row_kernels_on_long_and_short_rows(api, rows, inner):
    """One workgroup per row: 1024 threads and 16-byte loads for long aligned rows, the scalar walk otherwise -- same results."""
    hl = api.hl
    g = torch.Generator(device="cpu").manual_seed(rows * 131 + inner)
    x = torch.randn(rows, inner, generator=g).cuda()
    mean, std = hl.rowstats(x, rows, inner)
    torch.testing.assert_close(mean.cpu(), x.double().mean(1).float().cpu(), rtol=0, atol=1e-7)
    torch.testing.assert_close(std.cpu(), x.double().std(1).float().cpu(), rtol=2e-7, atol=0)
    lo, hi = hl.minmax_rows(x, rows, inner)
    assert torch.equal(lo, x.amin(1)) and torch.equal(hi, x.amax(1))
    peak = hl.amax_mid(x.reshape(rows, inner, 1), rows, inner, 1, True)
    assert torch.equal(peak.reshape(-1), x.abs().amax(1))
    q = hl.abs_quantile_rows(x, rows, inner, 0.75)
    want = torch.quantile(x.abs().cpu(), 0.75, dim=1)
    torch.testing.assert_close(q.cpu().reshape(-1), want, rtol=1e-6, atol=0)


@pytest.mark.parametrize("mode", ["nearest-exact", "nearest", "bilinear", "bicubic"])
@pytest.mark.parametrize("kind", ["pyramid_old", "highres"])
def test_shrunk_levels_draw_only_the_taps_they_read(api, mode, kind):
    """sonar_levels_sampled_f32 against its definition: the levels written out whole from the same keys (sonar_level_normal_f32), shrunk
    by the resampler and summed -- the reference's loops (py/noise_generation.py:517-606).  PyramidOld's power-of-two levels and
    HighresPyramid's odd ratios (the resampler's own index rules either way); accumulation onto a base; shards of a batch add up."""
    hl = api.hl
    shape = (3, 4, 24, 40)
    if kind == "pyramid_old":
        levels = [(24 * (2 << i), 40 * (2 << i), 0.8**i, 0.5**i) for i in range(4)]
    else:
        levels = [(24, 40, 1.0, 1.0), (61, 93, 0.7, 1.0), (157, 331, 0.49, 1.0), (360, 600, 0.343, 1.0)]
    got = hl.levels_sampled(shape, "cuda", levels, mode, 77, 5, plane_offset=8)
    want = torch.zeros(shape, device="cuda")
    for i, (lh, lw, wt, sd) in enumerate(levels):
        level = hl.level_normal((3, 4, lh, lw), "cuda", sd, 77, 5 + i, plane_offset=8)
        assert abs(level.std().item() - sd) < 0.03 * sd
        hl.resample_acc_(want, level, wt, mode, True)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-6)
    base = torch.randn(shape, device="cuda")
    onto = hl.levels_sampled(shape, "cuda", levels, mode, 77, 5, plane_offset=8, out=base.clone())
    torch.testing.assert_close(onto, base + got, rtol=1e-5, atol=2e-6)
    a = hl.levels_sampled((1, 4, 24, 40), "cuda", levels, mode, 77, 5, plane_offset=8)
    b = hl.levels_sampled((2, 4, 24, 40), "cuda", levels, mode, 77, 5, plane_offset=12)
    assert torch.equal(torch.cat([a, b]), got)


def test_area_mode_draws_the_block_means(api):
    """The mean of a block of independent normals is one normal of the mean's variance: drawn as such for levels of whole multiples of the
    output size (same distribution as pooling the levels); other ratios' windows overlap -- refused, the caller pools drawn levels."""
    hl = api.hl
    levels = [(64 * (2 << i), 64 * (2 << i), 0.8**i, 0.5**i) for i in range(4)]
    area = hl.levels_sampled((8, 4, 64, 64), "cuda", levels, "area", 77, 5)
    pooled = torch.zeros((8, 4, 64, 64), device="cuda")
    for i, (lh, lw, wt, sd) in enumerate(levels):
        hl.resample_acc_(pooled, hl.level_normal((8, 4, lh, lw), "cuda", sd, 77, 5 + i), wt, "area", True)
    want_sd = sum((wt * sd / (2 << i)) ** 2 for i, (_, _, wt, sd) in enumerate(levels)) ** 0.5
    assert abs(area.std().item() / want_sd - 1.0) < 0.01 and abs(pooled.std().item() / want_sd - 1.0) < 0.01 and abs(area.mean().item()) < 0.01 * want_sd
    assert abs((area[..., 1:] * area[..., :-1]).mean().item()) / want_sd**2 < 0.01  # neighbours are independent
    assert hl.levels_sampled((1, 4, 64, 64), "cuda", [(100, 130, 1.0, 1.0)], "area", 77, 5) is None
    assert hl.levels_sampled((1, 4, 8, 8), "cuda", [(16, 16, 1.0, 1.0)] * 17, "bilinear", 77, 5) is None


def test_shrunk_level_samplers_on_device(api):
    """The registry types end to end: finite, PyramidOld shard invariant, the area variants (block means / pooled levels from the same keys)."""
    ng = api.noise_generation
    SIG = (torch.tensor(14.6), torch.tensor(10.0))

    def gen(name, b0, b):
        torch.manual_seed(9)
        with ng.shard_offset(b0):
            x = torch.zeros((b, 4, 32, 32), device="cuda")
            return api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=False)(*SIG)

    for name in ("pyramid_old", "pyramid_old_area"):
        whole = gen(name, 0, 4)
        assert bool(torch.isfinite(whole).all()) and 0.5 < float(whole.std()) < 3.0
        assert torch.equal(torch.cat([gen(name, 0, 2), gen(name, 2, 2)]), whole)
    for name in ("highres_pyramid", "highres_pyramid_area"):  # their uniform base is normalised per call tensor: shards differ by design
        whole = gen(name, 0, 4)
        assert bool(torch.isfinite(whole).all()) and 0.3 < float(whole.std()) < 5.0
