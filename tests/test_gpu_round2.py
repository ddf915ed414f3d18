"""-m gpu: rows finished in round 2 against goldens captured from the real reference (tests/golden/make_golden.py:
gen_pyramid_variants, gen_ffilter, gen_cfg_exact), and every BASELINE.json configuration at its FULL size.

Tolerances: replay-mode elementwise / resampling rows rtol 2e-5 atol 2e-5 (HighresPyramid / PyramidOld sum levels drawn at up to
32x the latent resolution and averaged down: accumulation order differs from ATen's); FFT rows 2e-5 of the output peak; sampler
traces rtol / atol 1e-4 on |x| ~ 10 (20 steps for cfg1: 3e-4)."""
import importlib
import math
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SIG = (torch.tensor(14.6), torch.tensor(10.0))
PYRAMID_VARIANTS = ("highres_pyramid", "highres_pyramid_area", "pyramid_old", "pyramid_old_area", "pyramid_mix", "pyramid_mix_area")


@pytest.fixture(scope="module")
def api(pkg):
    pkg.hip_lib.load()
    mods = {m: importlib.import_module(f"comfyui_sonar_amd.py.{m}") for m in ("utils", "noise_generation", "noise", "sonar", "wavelet_cfg")}
    mods["powernoise"] = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
    mods["freeu"] = importlib.import_module("comfyui_sonar_amd.py.nodes.freeu_extreme")
    return types.SimpleNamespace(**mods, hl=pkg.hip_lib)


def close(a, b, rtol=2e-5, atol=2e-5):
    torch.testing.assert_close(a.detach().cpu().float(), b.detach().cpu().float(), rtol=rtol, atol=atol)


def near(a, b, rel=2e-5):
    b = b.detach().cpu().float()
    torch.testing.assert_close(a.detach().cpu().float(), b, rtol=0, atol=rel * float(b.abs().max()))


def replay(api, name, shape, seed, normalized, **kw):
    x = torch.zeros(shape, device="cuda")
    torch.manual_seed(seed)
    ns = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=seed, cpu=True, factor=1.0, normalized=normalized, **kw)
    return ns(*SIG)


# ------------------------------------------------------------------------------------------------ every registry type on odd shapes
def _sweep_cases():
    import json
    import numpy as np
    from tests.conftest import GOLDEN

    g = np.load(f"{GOLDEN}/shape_sweep.npz", allow_pickle=False)
    return g, json.loads(str(g["meta_json"]))


@pytest.mark.parametrize("key", sorted(_sweep_cases()[1]))
def test_registry_types_on_odd_shapes(api, key):
    """Every NoiseType the path covers, replay mode, against the reference's own output on shapes the other fixtures do not have
    (1 / 3 / 5 / 16 channels, H != W, odd sizes, a 5-D video latent; tests/golden/make_golden.py gen_shape_sweep).  fp32 op sequences
    through FFTs and resamplers: 2e-5 of the output's peak."""
    g, meta = _sweep_cases()
    m = meta[key]
    assert m["error"] is None
    want = torch.from_numpy(g[key])
    got = replay(api, m["type"], tuple(m["shape"]), m["seed"], m["normalized"])
    assert got.is_cuda and tuple(got.shape) == tuple(want.shape) and got.dtype == torch.float32
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-5 * max(1.0, float(want.abs().max())))


from tests.golden import sweep_cases as _sc  # noqa: E402


@pytest.mark.parametrize("name", sorted(_sc.WRAPPERS))
def test_item_wrappers_on_odd_shapes(api, golden, name):
    """Random / Repeated / Channel / RippleFiltered / PerDim / Scheduled / Blended wrappers (py/noise.py) over inner chains, several calls
    each, on odd shapes and a 5-D video latent, against the reference's outputs (tests/golden/make_golden.py gen_wrapper_sweep)."""
    want = golden("wrapper_sweep")[name]
    item, shape, seed, calls = _sc.build(api.noise, api.utils, name)
    x = torch.zeros(shape, device="cuda")
    torch.manual_seed(seed)
    ns = item.make_noise_sampler(x, 0.03, 14.6, seed=seed, cpu=True, normalized=True)
    for k in range(calls):
        s, sn = _sc.SIGMAS[k % len(_sc.SIGMAS)]
        got = ns(torch.tensor(s), torch.tensor(sn))
        assert got.is_cuda and tuple(got.shape) == tuple(want[k].shape)
        torch.testing.assert_close(got.cpu(), want[k], rtol=2e-5, atol=2e-5 * max(1.0, float(want[k].abs().max())), equal_nan=True)


@pytest.mark.parametrize("name", sorted(_sc.SAMPLERS))
def test_samplers_on_odd_shapes(api, golden, name):
    """SonarEuler / SonarEulerAncestral / SonarDPMPPSDE end to end with registry noise (replay mode) on odd shapes, other channel counts
    and a 5-D video latent: every step's x against the reference's run (tests/golden/make_golden.py gen_sampler_sweep)."""
    want = golden("sampler_sweep")[name]
    trace = _sc.run_sampler(api.sonar, api.noise, name, "cuda")
    assert len(trace) == want.shape[0]
    for i, t in enumerate(trace):
        assert t.is_cuda
        torch.testing.assert_close(t.cpu(), want[i], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("name", sorted(_sc.ADVANCED))
def test_advanced_items_on_odd_shapes(api, golden, name):
    """ModulatedNoise (intensity / frequency), GuidedNoise (reference latent resized by non-integer ratios), AdvancedWaveletNoise,
    PowerFilterNoiseItem and PowerNoiseItem on odd planes (spectral work through the direct DFT passes) against the reference's outputs
    (tests/golden/make_golden.py gen_advanced_sweep).  FFT / resampler chains in fp32: 4e-5 of the output's peak."""
    want = golden("advanced_sweep")[name]
    outs = _sc.run_advanced(api.noise, api.powernoise, api.utils, name, "cuda")
    assert len(outs) == want.shape[0]
    for got, w in zip(outs, want):
        assert got.is_cuda and tuple(got.shape) == tuple(w.shape)
        torch.testing.assert_close(got.cpu(), w, rtol=4e-5, atol=4e-5 * max(1.0, float(w.abs().max())))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_precision_and_strided_latents(api, dtype):
    """The reference draws and steps in the latent's dtype and takes any strides.  Here a half / bfloat16 latent gets fp32 noise rounded once
    (chains, registry samplers, generate and replay mode) and samplers that carry fp32 between the steps and return the latent's dtype;
    channels-last, transposed and sliced latents give the result of their contiguous copies."""
    N, S = api.noise, api.sonar
    g = torch.Generator().manual_seed(1)
    x32 = (torch.randn(2, 4, 16, 24, generator=g) * 10).cuda()
    big = torch.zeros(2, 4, 32, 27, device="cuda", dtype=dtype)
    big[:, :, ::2, 1:-2] = x32.to(dtype)
    layouts = {"contiguous": x32.to(dtype), "channels_last": x32.to(dtype).contiguous(memory_format=torch.channels_last),
               "transposed": x32.to(dtype).transpose(2, 3).contiguous().transpose(2, 3), "sliced": big[:, :, ::2, 1:-2]}
    sigmas = torch.cat((torch.linspace(14.6, 0.03, 5), torch.zeros(1)))
    model = lambda x, sigma, **_k: x * 0.5  # noqa: E731
    for name in ("gaussian", "perlin", "brownian", "onef_pinkish"):
        for cpu in (True, False):
            if name == "brownian" and cpu:
                continue
            torch.manual_seed(5)
            want = N.get_noise_sampler(name, x32, 0.03, 14.6, seed=5, cpu=cpu, normalized=True)(torch.tensor(10.0), torch.tensor(7.0))
            for tag, xv in layouts.items():
                torch.manual_seed(5)
                got = N.get_noise_sampler(name, xv, 0.03, 14.6, seed=5, cpu=cpu, normalized=True)(torch.tensor(10.0), torch.tensor(7.0))
                assert got.dtype == dtype and tuple(got.shape) == tuple(xv.shape)
                assert torch.equal(got, want.to(dtype)), (name, cpu, tag)
    chain = N.CustomNoiseChain()
    chain.add(N.CustomNoiseItem(0.5, noise_type="perlin"))
    chain.add(N.CustomNoiseItem(0.5, noise_type="pyramid"))
    torch.manual_seed(6)
    want = chain.make_noise_sampler(x32, 0.03, 14.6, seed=6, cpu=False, normalized=True)(torch.tensor(10.0), torch.tensor(7.0))
    torch.manual_seed(6)
    got = chain.make_noise_sampler(layouts["sliced"], 0.03, 14.6, seed=6, cpu=False, normalized=True)(torch.tensor(10.0), torch.tensor(7.0))
    assert got.dtype == dtype and torch.equal(got, want.to(dtype))
    for kind in ("euler", "ancestral", "dpmpp"):
        outs = {}
        for tag, xv in layouts.items():
            torch.manual_seed(3)
            ns = N.get_noise_sampler("gaussian", xv, 0.03, 14.6, seed=3, cpu=True, normalized=True)
            if kind == "euler":
                outs[tag] = S.SonarEuler.sampler(model, xv, sigmas, {"seed": 3}, None, True, ns, None, {})
            elif kind == "ancestral":
                outs[tag] = S.SonarEulerAncestral.sampler(model, xv, sigmas, {"seed": 3}, None, True, None, {}, 0.8, 1.1, ns)
            else:
                outs[tag] = S.SonarDPMPPSDE.sampler(model, xv, sigmas, {"seed": 3}, None, True, None, {}, 0.9, 1.05, ns)
            assert outs[tag].dtype == dtype and bool(torch.isfinite(outs[tag]).all())
            assert torch.equal(outs[tag], outs["contiguous"]), (kind, tag)


def _tiny_cases():
    import json
    import numpy as np
    from tests.conftest import GOLDEN

    g = np.load(f"{GOLDEN}/tiny_sweep.npz", allow_pickle=False)
    return g, json.loads(str(g["meta_json"]))


@pytest.mark.parametrize("key", sorted(_tiny_cases()[1]))
def test_registry_types_on_degenerate_latents(api, key):
    """Single pixels, 1-pixel rows and 2 x 2 / 3 x 3 planes through every registry type, replay mode: the reference's
    output where it has one (NaN where a one-element std is NaN), the reference's refusal where it refuses (same exception type)."""
    g, meta = _tiny_cases()
    m = meta[key]
    shape = tuple(m["shape"])
    if m["error"] is not None:
        with pytest.raises(Exception) as exc:
            replay(api, m["type"], shape, m["seed"], m["normalized"])
        assert type(exc.value).__name__ == m["error"], (exc.value, m["message"])
        return
    want = torch.from_numpy(g[key])
    got = replay(api, m["type"], shape, m["seed"], m["normalized"])
    assert got.is_cuda and tuple(got.shape) == tuple(want.shape)
    peak = float(want[torch.isfinite(want)].abs().max()) if bool(torch.isfinite(want).any()) else 1.0
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-5 * max(1.0, peak), equal_nan=True)


def _node_cases():
    import json
    import numpy as np
    from tests.conftest import GOLDEN

    g = np.load(f"{GOLDEN}/node_sweep.npz", allow_pickle=False)
    return g, json.loads(str(g["meta_json"]))


@pytest.mark.parametrize("key", sorted(_node_cases()[1]))
def test_chain_nodes_with_default_sockets(pkg, api, key):
    """Every node that returns a noise chain, run with the defaults of its own sockets (tests/golden/node_abi.json) over a gaussian base chain
    and sampled twice on a 10 x 14 latent, against the same graph run through the reference's nodes (gen_node_sweep).  Nodes outside the
    path raise NotImplementedError here; the reference's ValueError refusals are refused the same way."""
    import importlib
    import json
    from tests.conftest import GOLDEN

    g, meta = _node_cases()
    abi = json.load(open(f"{GOLDEN}/node_abi.json"))
    mappings = importlib.import_module("comfyui_sonar_amd.py.nodes.registry").NODE_CLASS_MAPPINGS
    if mappings[key].__name__.startswith("OffPath_"):
        with pytest.raises(NotImplementedError):
            _sc.run_node(mappings, abi, key, "cuda")
        return
    m = meta[key]
    if m["error"] is not None:
        if m["error"] != "ValueError":
            pytest.skip(f"the reference refused for a reason outside the path (missing package, its own socket table): {m['message']}")
        with pytest.raises(ValueError):
            _sc.run_node(mappings, abi, key, "cuda")
        return
    want = torch.from_numpy(g[key.replace(" ", "_")])
    outs = _sc.run_node(mappings, abi, key, "cuda")
    assert outs is not None and len(outs) == want.shape[0]
    for got, w in zip(outs, want):
        assert got.is_cuda and tuple(got.shape) == tuple(w.shape)
        torch.testing.assert_close(got.cpu(), w, rtol=4e-5, atol=4e-5 * max(1.0, float(w.abs().max())))


# ------------------------------------------------------------------------------------------------ row U: every F.interpolate mode of scale_samples
def test_scale_samples_every_mode(api, golden):
    """py/utils.py:58-67 against the reference's outputs: bilinear, nearest-exact, nearest, area, bicubic, adaptive_avg_pool2d; enlarging,
    shrinking, non-integer ratios, same size, down to one pixel.  fp32 resampling weights: 2e-6 of the output peak + 2e-6."""
    g = golden("resample_modes")
    seen = 0
    for key in g:
        if not key.startswith("scale_"):
            continue
        _, tag, size, mode = key.split("_", 3)
        h, w = (int(v) for v in size.split("x"))
        want = g[key]
        got = api.utils.scale_samples(g[f"src_{tag}"].cuda(), w, h, mode=mode)
        assert got.is_cuda and tuple(got.shape) == tuple(want.shape)
        torch.testing.assert_close(got.cpu(), want, rtol=0, atol=2e-6 * float(want.abs().max()) + 2e-6, msg=lambda m, key=key: f"{key}: {m}")
        seen += 1
    assert seen == 9 * 6
    with pytest.raises(NotImplementedError):
        api.utils.scale_samples(g["src_a"].cuda(), 8, 8, mode="bislerp")


@pytest.mark.parametrize("mode", ["bicubic", "nearest"])
@pytest.mark.parametrize("name,seed,normalized", [("pyramid", 63, True), ("pyramid_old", 64, False), ("highres_pyramid", 65, True)])
def test_pyramid_levels_through_bicubic_and_nearest(api, golden, name, seed, normalized, mode):
    want = golden("resample_modes")[f"{name}_{mode}"]
    close(replay(api, name, tuple(want.shape), seed, normalized, upscale_mode=mode), want)
    if name == "pyramid":  # device draws with these modes: the unfused kernels, shard-invariant like the fused ones
        ng = api.noise_generation

        def gen(b0, b):
            torch.manual_seed(9)
            with ng.shard_offset(b0):
                x = torch.zeros((b, 4, 32, 32), device="cuda")
                return api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=False, upscale_mode=mode)(*SIG)

        whole = gen(0, 4)
        assert bool(torch.isfinite(whole).all()) and 0.8 < float(whole.std()) < 3.0
        assert torch.equal(torch.cat([gen(0, 2), gen(2, 2)]), whole)


@pytest.mark.parametrize("method", ["linear", "euler"])
def test_guided_noise_resizes_its_reference(api, golden, method):
    """py/noise.py:581-588: a reference latent of another size goes through bicubic, align_corners=True."""
    g = golden("resample_modes")
    item = api.noise.GuidedNoise(1.0, guidance_factor=0.4, ref_latent=g["guided_ref"].cuda(), method=method, normalize_noise=None,
                                 normalize_result=None, noise=None)
    ns = item.make_noise_sampler(g["guided_x"].cuda(), 0.03, 14.6, seed=96, cpu=True, normalized=True)
    close(ns(torch.tensor(9.0), torch.tensor(6.0)), g[f"guided_{method}"])


# ------------------------------------------------------------------------------------------------ row Y: pyramid variants
@pytest.mark.parametrize("normalized", [False, True])
@pytest.mark.parametrize("name", PYRAMID_VARIANTS)
def test_pyramid_variants_replay(api, golden, name, normalized):
    g = golden("pyramid_variants")
    want = g[f"{name}_{int(normalized)}"]
    close(replay(api, name, tuple(want.shape), 61, normalized), want)


def test_highres_pyramid_video_latent(api, golden):
    want = golden("pyramid_variants")["video_highres"]
    close(replay(api, "highres_pyramid", tuple(want.shape), 62, True), want)


@pytest.mark.parametrize("name", ["highres_pyramid", "pyramid_old", "pyramid_mix"])
def test_pyramid_variants_generate_mode(api, name):
    """Device draws: unit statistics after normalisation, shard invariance (two halves of a batch == the whole)."""
    ng = api.noise_generation
    shape = (4, 4, 32, 32)

    def gen(b0, b, normalized=False):
        torch.manual_seed(9)
        with ng.shard_offset(b0):
            x = torch.zeros((b, *shape[1:]), device="cuda")
            return api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=normalized)(*SIG)

    whole = gen(0, 4)
    assert bool(torch.isfinite(whole).all())
    if name == "pyramid_old":  # the other two normalise parts per call tensor (uniform base / each mixed generator): shards differ by design
        assert torch.equal(torch.cat([gen(0, 2), gen(2, 2)]), whole)
    normed = gen(0, 4, True)
    thr = 2.5 / math.sqrt(normed.numel())
    assert abs(normed.std().item() - 1.0) < thr + 1e-4 and abs(normed.mean().item()) < thr + 1e-4


# ------------------------------------------------------------------------------------------------ 8f-2: FreeU-Extreme ffilter
def test_ffilter_matches_reference(api, golden):
    g = golden("ffilter")
    PF = api.powernoise.PowerFilter
    specs = {"a": (dict(alpha=1.0, max_freq=0.7071), 1.0), "b": (dict(alpha=-0.5, min_freq=0.05, max_freq=0.5), 0.7),
             "c": (dict(alpha=2.0, max_freq=0.7071, stretch=1.5, rotate=20.0), 1.0)}
    cache = {}
    for tag, (fkw, nf) in specs.items():
        x = g[f"{tag}_x"].cuda()
        out = api.freeu.ffilter(x, PF(**fkw), normalization_factor=nf, cfg_idx=tag, filter_cache=cache)
        assert out.dtype == x.dtype and out.shape == x.shape
        near(out, g[f"{tag}_out"])
    assert set(cache) == {("a", torch.Size([32, 32])), ("b", torch.Size([16, 64])), ("c", torch.Size([40, 56]))}
    assert all(v.is_cuda for v in cache.values())
    # a cache hit wins over the filter argument (py/nodes/freeu_extreme.py:13-16)
    near(api.freeu.ffilter(g["a_x"].cuda(), PF(alpha=3.0), cfg_idx="a", filter_cache=cache), g["a_cached_out"])
    half = api.freeu.ffilter(g["a_x"].cuda().half(), PF(alpha=1.0, max_freq=0.7071), cfg_idx=0, filter_cache={})
    assert half.dtype == torch.float16
    near(half, g["a_half_out"], rel=2e-3)  # fp16 input and output rounding
    for kw in (dict(), dict(cfg_idx=1), dict(filter_cache={})):  # the reference fails without a cache key
        with pytest.raises(UnboundLocalError):
            api.freeu.ffilter(g["a_x"].cuda(), PF(alpha=1.0), **kw)
    with pytest.raises(api.hl.SonarHipError):
        api.freeu.ffilter(g["a_x"], PF(alpha=1.0), cfg_idx=0, filter_cache={})


# ------------------------------------------------------------------------------------------------ BASELINE cfg1, exactly
def test_cfg1_exact(api, golden):
    """x0 = randn(1,4,64,64, seed 3) * 14.6, linspace(14.6, 0.03, 20) + [0], den = 0.5 x, SonarEuler(momentum = 0.95)."""
    g = golden("cfg_exact")
    torch.manual_seed(3)
    x0 = torch.randn(1, 4, 64, 64) * 14.6
    assert torch.equal(x0, g["cfg1_x0"])
    trace = []
    out = api.sonar.SonarEuler.sampler(lambda x, sigma, **_k: x * 0.5, x0.cuda(), g["cfg1_sigmas"], {"seed": 0}, lambda d: trace.append(d["x"].clone()), True,
                                       None, None, dict(momentum=0.95))
    assert len(trace) == 20
    for i, t in enumerate(trace):
        close(t, g["cfg1_trace"][i], rtol=3e-4, atol=3e-4)
    close(out, g["cfg1_out"], rtol=3e-4, atol=3e-4)
    close(replay(api, "gaussian", (1, 4, 64, 64), 3, True), g["cfg1_noise"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kind", ["euler", "ancestral", "dpmpp"])
def test_rand_history_init_keeps_the_reference_draw_order(api, golden, kind):
    """init = RAND draws the history from the global generator inside the first momentum step, before that step's noise."""
    g = golden("cfg_exact")
    S = api.sonar
    x1, sig7 = g["rand_x0"].cuda(), g["rand_sigmas"]

    def fake_model(x, sigma, **_kw):
        s = sigma.reshape(-1, *([1] * (x.ndim - 1)))
        return x * 0.5 + torch.tanh(x) * (0.1 * s / (1.0 + s))

    trace = []
    cb = lambda d: trace.append(d["x"].clone())  # noqa: E731
    torch.manual_seed(17)
    ns = api.noise.get_noise_sampler("gaussian", torch.zeros_like(x1), 0.03, 14.6, seed=17, cpu=True, factor=1.0, normalized=True)
    kw = dict(init="RAND", momentum=0.9)
    if kind == "euler":
        S.SonarEuler.sampler(fake_model, x1.clone(), sig7, {"seed": 0}, cb, True, None, None, kw)
    elif kind == "ancestral":
        S.SonarEulerAncestral.sampler(fake_model, x1.clone(), sig7, {"seed": 0}, cb, True, None, kw, 0.8, 1.1, ns)
    else:
        S.SonarDPMPPSDE.sampler(fake_model, x1.clone(), sig7, {"seed": 0}, cb, True, None, kw, 0.9, 1.05, ns)
    want = g[f"rand_{kind}"]
    assert len(trace) == want.shape[0]
    for i, t in enumerate(trace):
        close(t, want[i], rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------------------------------------ BASELINE cfg2 at full size
def power_item(api):
    return api.powernoise.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                                         common_mode=0.0, channel_correlation="1,1,1,1,1,1")


def test_cfg2_headline_full_size(api):
    """512 x 4 x 128 x 128 through sonar_power_noise_f32 (the benchmarked call): unit statistics, 8 sampled planes equal
    irfft2(dumped spectrum x filter) normalised with the tensor's own statistics, and the two 256-latent shards equal the whole."""
    ng, hl = api.noise_generation, api.hl
    item = power_item(api)
    shape = (512, 4, 128, 128)
    x = torch.zeros(shape, device="cuda")
    torch.manual_seed(2024)
    state = torch.cuda.get_rng_state()
    whole = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)(None, None)
    assert whole.shape == shape and bool(torch.isfinite(whole).all())
    assert abs(whole.std().item() - 1.0) < 1e-4 and abs(whole.mean().item()) < 1e-4
    # the same draws, un-normalised, and the spectrum behind them
    torch.cuda.set_rng_state(state)
    raw = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=False)(None, None)
    torch.cuda.set_rng_state(state)
    seed, stream = ng.DeviceRNG.take()
    filt = item.make_filter(shape).to("cuda")[0, 0]
    planes = [0, 1, 777, 1024, 1500, 2045, 2046, 2047]
    for p in planes:
        b, c = divmod(p, 4)
        # dump the RNG group (4 planes) that holds plane p
        z = hl.power_spectrum((1, 4, 128, 128), x.device, seed=seed, stream_id=stream, plane_offset=b * 4)
        want = torch.fft.irfft2(z[0, c] * filt, s=(128, 128), norm="ortho")
        torch.testing.assert_close(raw[b, c], want, rtol=0, atol=2e-5 * float(want.abs().max()))
    mean, std = raw.mean().item(), raw.std().item()
    torch.testing.assert_close(whole[300], (raw[300] - (mean if abs(mean) > 2.5 / math.sqrt(raw.numel()) else 0.0)) / std, rtol=1e-5, atol=1e-5)
    # shards: ranks 0 and 1 of a 2-GPU job draw latents [0, 256) and [256, 512) of the same logical batch
    parts = []
    for b0 in (0, 256):
        torch.cuda.set_rng_state(state)
        with ng.shard_offset(b0):
            parts.append(item.make_noise_sampler(x[:256], None, None, seed=None, cpu=False, normalized=False)(None, None))
    assert torch.equal(torch.cat(parts), raw)


# ------------------------------------------------------------------------------------------------ BASELINE cfg3 at full size
@pytest.mark.parametrize("name", ["perlin", "pyramid"])
def test_cfg3_replay_at_sdxl_size(api, golden, name):
    want = golden("cfg_exact")[f"cfg3_{name}"]
    close(replay(api, name, (2, 4, 128, 128), 0, True), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", ["perlin", "pyramid"])
def test_cfg3_generate_full_batch(api, name):
    """64 x 4 x 128 x 128 drawn on device: unit statistics, two 32-latent shards == the whole, the Perlin lattice term is shared."""
    ng = api.noise_generation
    shape = (64, 4, 128, 128)

    def gen(b0, b, normalized):
        torch.manual_seed(31)
        with ng.shard_offset(b0):
            x = torch.zeros((b, *shape[1:]), device="cuda")
            return api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=normalized)(*SIG)

    raw = gen(0, 64, False)
    assert torch.equal(torch.cat([gen(0, 32, False), gen(32, 32, False)]), raw)
    normed = gen(0, 64, True)
    assert bool(torch.isfinite(normed).all())
    thr = 2.5 / math.sqrt(normed.numel())
    assert abs(normed.std().item() - 1.0) < thr + 1e-4 and abs(normed.mean().item()) < thr + 1e-4
    if name == "perlin":  # two latents differ only by their uniform base / div_fac: |diff| <= 0.5
        assert float((raw[0] - raw[1]).abs().max()) <= 0.5 + 1e-6


# ------------------------------------------------------------------------------------------------ BASELINE cfg4 at full size
def test_cfg4_full_batch_matches_small_batch(api):
    """256 x 4 x 128 x 128 WaveletCFG (placeholder rule, fp64 and fp32): per-plane arithmetic, so the first 2 latents of the full batch
    equal a 2-latent call (the 2 x 4 x 128 x 128 case is pinned to the reference in tests/test_gpu_wavelet_golden.py)."""
    from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel

    torch.manual_seed(12)
    cond, uncond, x = (torch.randn(256, 4, 128, 128, device="cuda") for _ in range(3))
    opts = {"transformer_options": {"sample_sigmas": SAMPLE_SIGMAS["karras12"]}}

    def args(n):
        return {"input": x[:n], "cond_scale": 7.0, "cond": x[:n] - cond[:n], "uncond": x[:n] - uncond[:n], "cond_denoised": cond[:n],
                "uncond_denoised": uncond[:n], "sigma": torch.full((n,), 7.0, device="cuda"), "model": FakeModel(), "model_options": opts}

    for hp in (True, False):
        fn = api.wavelet_cfg.WaveletCFG(existing_cfg=None, rules=api.wavelet_cfg.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=3.0),
                                                                                                 high_precision_mode=hp))
        full, small = fn(args(256)), fn(args(2))
        assert full.shape == x.shape and bool(torch.isfinite(full).all())
        assert torch.equal(full[:2], small)


# ------------------------------------------------------------------------------------------------ BASELINE cfg5: one rank's shard
def test_cfg5_full_shard_one_dpmpp_step(api):
    """128 x 16 x 128 x 128 (one GPU's share of the 1024-latent Flux batch): scheduled power-law + Perlin + Brownian chain in generate
    mode, one SonarDPMPPSDE step with momentum: finite, the chain's noise is unit-variance, and two 64-latent shards equal the whole."""
    N, S, ng, pn = api.noise, api.sonar, api.noise_generation, api.powernoise

    def chain_of(*items):
        c = N.CustomNoiseChain()
        for it in items:
            c.add(it)
        return c

    def build():
        power = pn.PowerNoiseItem(0.5, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                                  common_mode=0.0, channel_correlation="1")
        inner = chain_of(power, N.CustomNoiseItem(0.3, noise_type="perlin"), N.CustomNoiseItem(0.2, noise_type="brownian"))
        fallback = chain_of(N.CustomNoiseItem(1.0, noise_type="gaussian"))
        return chain_of(N.ScheduledNoise(1.0, noise=inner, start_sigma=20.0, end_sigma=0.0, normalize=None, fallback_noise=fallback))

    sigmas = torch.tensor([10.0, 7.0, 0.0])
    torch.manual_seed(1)
    x0 = torch.randn(128, 16, 128, 128, device="cuda") * 10.0

    def run(xs, b0):
        torch.manual_seed(8)
        with ng.shard_offset(b0):
            ns = build().make_noise_sampler(xs, 0.5, 10.0, seed=3, cpu=False, normalized=True)
            sample = ns(torch.tensor(8.0), torch.tensor(6.0)).clone()
            out = S.SonarDPMPPSDE.sampler(lambda x, sigma, **_k: x * 0.5, xs.clone(), sigmas, {"seed": 3}, None, True, None,
                                          dict(momentum=0.95), 1.0, 1.0, ns)
        return sample, out

    sample, out = run(x0, 0)
    assert out.shape == x0.shape and bool(torch.isfinite(out).all())
    assert abs(sample.std().item() - 1.0) < 2e-3 and abs(sample.mean().item()) < 2e-3
    # each shard normalises its own call tensor (== the reference called with B/2), so compare un-normalised parts:
    # the Perlin lattice, the Brownian paths and the power spectra are keyed by global latent index
    parts = []
    for b0 in (0, 64):
        torch.manual_seed(8)
        with ng.shard_offset(b0):
            item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                                     common_mode=0.0, channel_correlation="1")
            parts.append(item.make_noise_sampler(x0[b0:b0 + 64], None, None, seed=None, cpu=False, normalized=False)(None, None))
    torch.manual_seed(8)
    item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                             common_mode=0.0, channel_correlation="1")
    assert torch.equal(torch.cat(parts), item.make_noise_sampler(x0, None, None, seed=None, cpu=False, normalized=False)(None, None))


def test_normalize_to_scale(api, golden):
    """py/utils.py:450-469 against the reference: min / max reduction + the rescale kernel, each step rounded as the reference's tensor ops
    are -- bit-exact, including a constant group (denominator eps only)."""
    g = golden("resample_modes")
    x = g["nts_in"].cuda()
    assert torch.equal(api.utils.normalize_to_scale(x.clone(), 0.0, 1.0).cpu(), g["nts_default"])
    assert torch.equal(api.utils.normalize_to_scale(x.clone(), -1.5, 2.0, dim=(-2, -1)).cpu(), g["nts_hw"])
    assert torch.equal(api.utils.normalize_to_scale(x.clone(), 0.25, 0.5, dim=(-4, -3, -2, -1), eps=1e-3).cpu(), g["nts_all"])
    assert torch.equal(api.utils.normalize_to_scale(x.clone(), 0.1, 0.3).cpu(), g["nts_inexact"])  # the span is rounded once, from double


# ------------------------------------------------------------------------------------------------ chains: generators that fold into the running sum
@pytest.mark.parametrize("n_shape", [(3, 4, 64, 64), (2, 3, 5, 7)])
def test_accumulating_generators_match_generate_then_axpby(api, n_shape):
    """sonar_{philox_normal,perlin_generate,brownian_bridge}_acc_f32: y <- y * a + x * b with x the values the plain entry point would
    have written -- bit-identical to plain + sonar_axpby_f32 (products rounded separately, a multiplier of 1 skipped), statistics of
    the new y included.  Shapes with and without the vector path."""
    hl = api.hl
    g = torch.Generator(device="cuda").manual_seed(4)
    y0 = torch.randn(n_shape, device="cuda", generator=g)
    n = y0.numel()
    per = n // n_shape[0]
    offs = 8 * per  # a shard that starts at latent 8
    for a, b in ((1.0, 1.0), (0.5, 0.3), (1.0, 0.2), (-1.25, 1.0)):
        # Gaussian
        x = hl.philox_normal(n_shape, "cuda", 1234, 7, offs)
        want = hl.axpby_(y0.clone(), a, x, b)
        part = hl.new_partials("cuda")
        got = hl.philox_normal_acc_(y0.clone(), a, b, 1234, 7, offs, part)
        assert torch.equal(got, want)
        s = part.view(-1, 2).sum(0)
        assert abs(s[0].item() - want.double().sum().item()) < 1e-6 * n and abs(s[1].item() - (want.double() ** 2).sum().item()) < 1e-6 * n
        # Perlin
        c, h, w = n_shape[1:]
        terms = hl.perlin_lattice(2, c, h, w, "cuda", "lerp", 99, 3)
        x = hl.perlin_generate(n_shape, terms, 2.0, 99, 2, offs, None)
        want = hl.axpby_(y0.clone(), a, x, b)
        part = hl.new_partials("cuda")
        got = hl.perlin_generate_acc_(y0.clone(), a, b, terms, 2.0, 99, 2, offs, part)
        assert torch.equal(got, want)
        s = part.view(-1, 2).sum(0)
        assert abs(s[0].item() - want.double().sum().item()) < 1e-6 * n
        # Brownian bridge between two kept tensors, differenced against one of them
        wa, wb = torch.randn(n_shape, device="cuda", generator=g), torch.randn(n_shape, device="cuda", generator=g)
        kw = dict(base_a=wa, fa=0.3, base_b=wb, fb=0.7, prev=wa, scale=1.7)
        x, w = hl.brownian_bridge(n_shape, "cuda", [5], [0.4], 77, offs, None, **kw)
        want = hl.axpby_(y0.clone(), a, x, b)
        part = hl.new_partials("cuda")
        yy = y0.clone()
        w2 = hl.brownian_bridge_acc_(yy, a, b, [5], [0.4], 77, offs, None, **kw, partials=part)
        assert torch.equal(yy, want) and torch.equal(w2, w)
        s = part.view(-1, 2).sum(0)
        assert abs(s[1].item() - (want.double() ** 2).sum().item()) < 1e-6 * n
        # no terms at all: the increment between two kept tensors
        x, _ = hl.brownian_bridge(n_shape, "cuda", [], [], 77, offs, None, base_b=wb, fb=1.0, prev=wa, scale=-0.5, want_w=False)
        assert torch.equal(x, (wb - wa) * -0.5)


def test_chain_of_folding_generators_equals_the_unfused_chain(api, monkeypatch):
    """A chain of power-law + Perlin + Brownian + Gaussian items (cfg5's mix): items after the first fold into the running sum.  Same seeds,
    folding switched off -> the same tensor, bit for bit, normalised or not."""
    N, pn = api.noise, api.powernoise

    def build():
        chain = N.CustomNoiseChain()
        chain.add(pn.PowerNoiseItem(0.5, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                                    common_mode=0.0, channel_correlation="1"))
        chain.add(N.CustomNoiseItem(0.3, noise_type="perlin"))
        chain.add(N.CustomNoiseItem(0.2, noise_type="brownian"))
        chain.add(N.CustomNoiseItem(0.1, noise_type="gaussian"))
        return chain

    x = torch.zeros(4, 4, 64, 64, device="cuda")
    steps = [(10.0, 7.0), (7.0, 4.0), (7.0, 5.5), (4.0, 1.0)]
    for normalized in (True, False):
        runs = []
        for fold in (True, False):
            with monkeypatch.context() as m:
                if not fold:
                    m.delattr(N.NoiseSampler, "accumulate")
                torch.manual_seed(21)
                ns = build().make_noise_sampler(x, 0.03, 14.6, seed=5, cpu=False, normalized=normalized)
                outs = []
                for s, sn in steps:
                    out = ns(torch.tensor(s), torch.tensor(sn))
                    outs.append((out.clone(), api.utils.pop_stats(out) is not None))
                runs.append(outs)
        for (a, tag_a), (b, _tag_b) in zip(*runs):
            assert torch.equal(a, b)
            assert bool(torch.isfinite(a).all()) and (not normalized or abs(a.std().item() - 1.1) < 3e-3)  # scale_noise(factor = sum |factor_i|)
            assert tag_a or not normalized  # the normalised result carries its statistics (derived from the sum's, no sweep)


@pytest.mark.parametrize("n_shape", [(3, 4, 64, 64), (2, 16, 128, 128), (2, 3, 5, 7), (2, 4, 40, 56)])
def test_brownian_kernel_hosts_the_previous_items_fold(api, n_shape):
    """sonar_brownian_bridge_chain_f32: the previous chain item's fold (Gaussian draw or Perlin, ``hip_lib.FoldPrefix``) evaluated inside
    the Brownian kernel's pass over the running sum == that item's own accumulating launch followed by the Brownian one, bit for bit
    (y and W), statistics of the final y included.  Latents that are not whole generator tiles: the prefix is applied by its own launch."""
    hl = api.hl
    g = torch.Generator(device="cuda").manual_seed(8)
    y0 = torch.randn(n_shape, device="cuda", generator=g)
    n, per = y0.numel(), y0.numel() // n_shape[0]
    offs = 8 * per
    c, h, w = n_shape[1:]
    terms = hl.perlin_lattice(3, c, h, w, "cuda", "lerp", 99, 3)
    wa, wb = torch.randn(n_shape, device="cuda", generator=g), torch.randn(n_shape, device="cuda", generator=g)
    routes = [dict(ids=[5], coefs=[0.4], base_a=wa, fa=0.3, base_b=wb, fb=0.7, prev=wa, scale=1.7),   # bridge between two kept tensors
              dict(ids=[], coefs=[], base_b=wb, fb=1.0, prev=wa, scale=-0.5),                          # both ends kept
              dict(ids=[3, 9, 11], coefs=[0.2, -0.7, 1.1], scale=0.9)]                                # a short expansion, W(t_lo) = 0
    for kind in ("normal", "perlin"):
        for (a1, b1), b2, route in zip(((0.5, 0.3), (1.0, 1.0), (-1.25, 0.2)), (0.2, 1.0, 0.6), routes):
            def prefix(y):
                if kind == "normal":
                    return hl.FoldPrefix(hl.PREFIX_NORMAL, y, a1, b1, 1234, 7, offs)
                return hl.FoldPrefix(hl.PREFIX_PERLIN, y, a1, b1, 99, 2, offs, terms=terms, div_fac=2.0)

            kw = {k: v for k, v in route.items() if k not in ("ids", "coefs")}
            want = y0.clone()
            prefix(want).apply()
            if kind == "normal":
                assert torch.equal(want, hl.philox_normal_acc_(y0.clone(), a1, b1, 1234, 7, offs))
            part_w = hl.new_partials("cuda")
            w_want = hl.brownian_bridge_acc_(want, 1.0, b2, route["ids"], route["coefs"], 77, offs, None, **kw, partials=part_w)
            got = y0.clone()
            pre = prefix(got)
            assert pre.hosted(got, offs, (wa, wb))  # (the Brownian wrapper adds its own condition: latents of whole generator tiles)
            part_g = hl.new_partials("cuda")
            w_got = hl.brownian_bridge_acc_(got, 1.0, b2, route["ids"], route["coefs"], 77, offs, None, **kw, partials=part_g, pre=pre)
            assert pre.consumed
            pre.apply()  # must be a no-op now
            assert torch.equal(got, want), (kind, route["ids"])
            assert torch.equal(w_got, w_want)
            sw, sg = part_w.view(-1, 2).sum(0), part_g.view(-1, 2).sum(0)
            torch.testing.assert_close(sg, sw, rtol=1e-12, atol=1e-9 * n)
    # per-latent seeds have no tile route: the prefix goes first, by its own launch
    if per % 4:
        return  # (that route needs latents of whole 4-element groups)
    seeds = torch.arange(1, n_shape[0] + 1, dtype=torch.int64, device="cuda")
    want = y0.clone()
    hl.philox_normal_acc_(want, 0.5, 0.3, 1234, 7, offs)
    hl.brownian_bridge_acc_(want, 1.0, 0.2, [5], [0.4], 0, offs, seeds, want_w=False)
    got = y0.clone()
    hl.brownian_bridge_acc_(got, 1.0, 0.2, [5], [0.4], 0, offs, seeds, want_w=False, pre=hl.FoldPrefix(hl.PREFIX_NORMAL, got, 0.5, 0.3, 1234, 7, offs))
    assert torch.equal(got, want)


@pytest.mark.parametrize("n_shape", [(3, 4, 64, 64), (2, 16, 128, 128), (2, 4, 40, 56), (3, 3, 20, 12)])
def test_pyramid_kernel_folds_and_hosts_the_previous_items_fold(api, n_shape):
    """sonar_pyramid_generate_acc_f32: y <- y * a + pyramid * b == sonar_pyramid_generate_f32 + sonar_axpby_f32, and with a Gaussian / Perlin
    prefix (also a FRESH one: the chain's first item, y not read at all) == that item's own launch first.  Bit for bit; planes that
    straddle generator tiles included."""
    hl = api.hl
    g = torch.Generator(device="cuda").manual_seed(9)
    y0 = torch.randn(n_shape, device="cuda", generator=g)
    b, c, h, w = n_shape
    per = c * h * w
    offs = 8 * per
    levels = [(None, h, w, 1.0), (None, max(1, h // 3), max(1, w // 3), 0.7), (None, max(1, h // 11), max(1, w // 11), 0.49)]
    terms = hl.perlin_lattice(3, c, h, w, "cuda", "lerp", 99, 3)
    x = hl.pyramid_generate(n_shape, "cuda", levels, "bilinear", 4321, 11, offs)
    assert x is not None
    for a, bb in ((1.0, 1.0), (0.5, 0.3), (-1.25, 0.2)):
        want = hl.axpby_(y0.clone(), a, x, bb)
        part = hl.new_partials("cuda")
        got = y0.clone()
        assert hl.pyramid_generate_acc_(got, a, bb, levels, "bilinear", 4321, 11, offs, part)
        assert torch.equal(got, want)
        # (groups of four are summed in fp32 before they enter the fp64 sums, in each kernel's own grouping)
        torch.testing.assert_close(part.view(-1, 2).sum(0), hl.stats(want).view(-1, 2).sum(0), rtol=1e-7, atol=1e-7 * y0.numel())
    for kind in ("normal", "perlin"):
        for fresh in (False, True):
            a1, b1, b2 = (1.0, 1.0, 0.7) if fresh else (0.5, 0.3, 0.2)

            def prefix(y):
                if kind == "normal":
                    return hl.FoldPrefix(hl.PREFIX_NORMAL, y, a1, b1, 1234, 7, offs, fresh=fresh)
                return hl.FoldPrefix(hl.PREFIX_PERLIN, y, a1, b1, 99, 2, offs, terms=terms, div_fac=2.0, fresh=fresh)

            want = torch.full(n_shape, float("nan"), device="cuda") if fresh else y0.clone()
            prefix(want).apply()
            if fresh:
                raw = hl.philox_normal(n_shape, "cuda", 1234, 7, offs) if kind == "normal" else hl.perlin_generate(n_shape, terms, 2.0, 99, 2, offs)
                assert torch.equal(want, raw)
            ymul = 0.6 if fresh else 1.0  # fresh: the first item's factor rides in the hosting fold
            hl.pyramid_generate_acc_(want, ymul, b2, levels, "bilinear", 4321, 11, offs)
            got = torch.full(n_shape, float("nan"), device="cuda") if fresh else y0.clone()
            pre = prefix(got)
            assert pre.hosted(got, offs)
            assert hl.pyramid_generate_acc_(got, ymul, b2, levels, "bilinear", 4321, 11, offs, pre=pre) and pre.consumed
            assert torch.equal(got, want), (kind, fresh)
    # a fresh prefix in the Brownian kernel
    if per % 4096 == 0:
        wa = torch.randn(n_shape, device="cuda", generator=g)
        want = hl.perlin_generate(n_shape, terms, 2.0, 99, 2, offs)
        hl.brownian_bridge_acc_(want, 0.6, 0.2, [5], [0.4], 77, offs, None, base_a=wa, fa=0.3, prev=wa, scale=1.7, want_w=False)
        got = torch.full(n_shape, float("nan"), device="cuda")
        pre = hl.FoldPrefix(hl.PREFIX_PERLIN, got, 1.0, 1.0, 99, 2, offs, terms=terms, div_fac=2.0, fresh=True)
        hl.brownian_bridge_acc_(got, 0.6, 0.2, [5], [0.4], 77, offs, None, base_a=wa, fa=0.3, prev=wa, scale=1.7, want_w=False, pre=pre)
        assert pre.consumed and torch.equal(got, want)


@pytest.mark.parametrize("n_shape", [(3, 4, 64, 64), (2, 16, 128, 128), (3, 3, 20, 12), (2, 3, 5, 7)])
def test_gaussian_and_perlin_kernels_host_the_previous_items_fold(api, n_shape):
    """sonar_philox_normal_chain_f32 / sonar_perlin_generate_chain_f32: the pair kernel (host item + the previous item riding along, also
    as the chain's first) == the two accumulating launches, bit for bit; a shard offset and latents that are not whole tiles included.
    Shapes without whole 4-element groups: the prefix goes first by its own launch (same result)."""
    hl = api.hl
    g = torch.Generator(device="cuda").manual_seed(10)
    y0 = torch.randn(n_shape, device="cuda", generator=g)
    b, c, h, w = n_shape
    per = c * h * w
    offs = 8 * per
    t_host = hl.perlin_lattice(2, c, h, w, "cuda", "lerp", 55, 4)
    t_pre = hl.perlin_lattice(3, c, h, w, "cuda", "lerp", 99, 3)
    for host in ("normal", "perlin"):
        for kind in ("normal", "perlin"):
            for fresh in (False, True):
                a1, b1 = (1.0, 1.0) if fresh else (0.5, 0.3)
                ymul, b2 = (0.6, 0.7) if fresh else (1.0, 0.2)

                def prefix(y):
                    if kind == "normal":
                        return hl.FoldPrefix(hl.PREFIX_NORMAL, y, a1, b1, 1234, 7, offs, fresh=fresh)
                    return hl.FoldPrefix(hl.PREFIX_PERLIN, y, a1, b1, 99, 2, offs, terms=t_pre, div_fac=2.0, fresh=fresh)

                def fold(y, part, pre):
                    if host == "normal":
                        return hl.philox_normal_acc_(y, ymul, b2, 4321, 9, offs, part, pre=pre)
                    return hl.perlin_generate_acc_(y, ymul, b2, t_host, 1.5, 55, 3, offs, part, pre=pre)

                want = torch.full(n_shape, float("nan"), device="cuda") if fresh else y0.clone()
                prefix(want).apply()
                pw = hl.new_partials("cuda")
                fold(want, pw, None)
                got = torch.full(n_shape, float("nan"), device="cuda") if fresh else y0.clone()
                pre = prefix(got)
                pg = hl.new_partials("cuda")
                fold(got, pg, pre)
                assert pre.consumed
                assert torch.equal(got, want), (host, kind, fresh)
                torch.testing.assert_close(pg.view(-1, 2).sum(0), pw.view(-1, 2).sum(0), rtol=1e-6, atol=1e-6 * y0.numel())


@pytest.mark.parametrize("items", [("gaussian", "perlin", "brownian"), ("gaussian", "gaussian", "brownian"), ("power", "perlin", "brownian", "perlin", "brownian"),
                                   ("gaussian", "perlin", "gaussian", "brownian", "gaussian"), ("perlin", "pyramid"), ("pyramid", "perlin"),
                                   ("perlin", "brownian"), ("gaussian", "pyramid", "perlin", "pyramid", "gaussian"), ("power", "gaussian", "pyramid"),
                                   ("gaussian", "perlin"), ("perlin", "gaussian"), ("gaussian", "gaussian", "perlin"), ("perlin", "perlin"),
                                   ("power", "perlin", "gaussian")])
def test_chain_with_hosted_folds_equals_the_plain_chain(api, monkeypatch, items):
    """Chains whose Brownian item follows a Gaussian / Perlin item: that item is not launched at all, the Brownian kernel applies it.  Same
    seeds with the hosting switched off (every item folds by itself) and with folding switched off -> the same tensors."""
    N, pn = api.noise, api.powernoise

    def build():
        chain = N.CustomNoiseChain()
        for k, name in enumerate(items):
            f = (0.5, 0.3, 0.2, 0.15, 0.1)[k]
            if name == "power":
                chain.add(pn.PowerNoiseItem(f, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                                            mix=1.0, common_mode=0.0, channel_correlation="1"))
            else:
                chain.add(N.CustomNoiseItem(f, noise_type=name))
        return chain

    x = torch.zeros(4, 4, 64, 64, device="cuda")
    steps = [(10.0, 7.0), (7.0, 4.0), (7.0, 5.5), (4.0, 1.0)]
    hosted = []
    real = api.hl.FoldPrefix.hosted
    for normalized in (True, False):
        runs = []
        for variant in ("hosted", "folded", "plain"):
            with monkeypatch.context() as m:
                if variant == "hosted":
                    m.setattr(api.hl.FoldPrefix, "hosted", lambda self, *a, **k: hosted.append(real(self, *a, **k)) or hosted[-1])
                elif variant == "folded":
                    m.setattr(N.NoiseSampler, "accepts_prefix", property(lambda self: False))
                else:
                    m.delattr(N.NoiseSampler, "accumulate")
                torch.manual_seed(21)
                ns = build().make_noise_sampler(x, 0.03, 14.6, seed=5, cpu=False, normalized=normalized)
                runs.append([ns(torch.tensor(s), torch.tensor(sn)).clone() for s, sn in steps])
        for a, b, c in zip(*runs):
            assert bool(torch.isfinite(a).all())
            if normalized:
                # the kernel that writes the sum last also reduces its statistics, in its own order of fp32 groups and fp64 partial
                # sums: the three routes agree on the sum bit for bit (normalized=False below) but may round the normalisation's
                # mean / std differently in the last bits
                torch.testing.assert_close(a, b, rtol=0, atol=2e-6)
                torch.testing.assert_close(b, c, rtol=0, atol=2e-6)
            else:
                assert torch.equal(a, b) and torch.equal(b, c)
    assert (hosted and all(hosted)) or items == ("pyramid", "perlin")  # (nothing in that chain can host)


@pytest.mark.parametrize("items", [("perlin", "pyramid"), ("power", "perlin", "brownian"), ("gaussian", "perlin", "gaussian")])
def test_hosted_chains_shard_like_their_items(api, items):
    """cfg3 / cfg5 chains at SDXL size, drawn on device: the batch 16 chain == its two 8-latent shards (global element keys in the
    hosting kernels too: ``elem_offset`` of the prefix and of the host are the shard's), over several sampler steps."""
    N, pn, ng = api.noise, api.powernoise, api.noise_generation

    def run(b0, b):
        chain = N.CustomNoiseChain()
        for k, name in enumerate(items):
            f = (0.5, 0.3, 0.2)[k]
            if name == "power":
                chain.add(pn.PowerNoiseItem(f, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                                            mix=1.0, common_mode=0.0, channel_correlation="1"))
            else:
                chain.add(N.CustomNoiseItem(f, noise_type=name))
        torch.manual_seed(77)
        with ng.shard_offset(b0):
            x = torch.zeros(b, 4, 128, 128, device="cuda")
            ns = chain.make_noise_sampler(x, 0.03, 14.6, seed=9, cpu=False, normalized=False)
            return [ns(torch.tensor(s), torch.tensor(sn)).clone() for s, sn in ((10.0, 7.0), (7.0, 4.0), (7.0, 5.5))]

    whole = run(0, 16)
    for step, (lo, hi) in enumerate(zip(run(0, 8), run(8, 8))):
        assert torch.equal(torch.cat([lo, hi]), whole[step]), step


def test_chain_routes_agree_on_random_shapes(api, monkeypatch):
    """Randomised differential run (fixed seed): random latent shapes -- odd sizes, 5-D video latents, tiny planes, shard offsets that
    are not multiples of four elements -- and random item lists; hosted, folded and plain routes give the same unnormalised sum, bit
    for bit.  (`scratch/fuzz_chains.py` is the long-running form.)"""
    import random

    N, pn, ng = api.noise, api.powernoise, api.noise_generation
    rnd = random.Random(7)
    kinds = ["gaussian", "perlin", "pyramid", "brownian", "uniform", "power", "laplacian"]
    for it in range(60):
        b, c = rnd.randint(1, 4), rnd.choice([1, 3, 4, 16])
        h, w = rnd.choice([(8, 8), (16, 24), (32, 32), (64, 64), (20, 12), (18, 30), (7, 9), (40, 56), (33, 17), (4, 4)])
        frames = rnd.choice([0, 0, 0, 3])
        shape = (b, c, frames, h, w) if frames else (b, c, h, w)
        items = [rnd.choice(kinds) for _ in range(rnd.randint(2, 4))]
        if h % 2 or w % 2 or frames:
            items = [k if k != "power" else "gaussian" for k in items]
        factors = [rnd.choice([1.0, 0.5, 0.3, -0.7, 0.25]) for _ in items]
        offset = rnd.choice([0, 0, 2, 5])

        def build():
            chain = N.CustomNoiseChain()
            for f, name in zip(factors, items):
                if name == "power":
                    chain.add(pn.PowerNoiseItem(f, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0,
                                                pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1"))
                else:
                    chain.add(N.CustomNoiseItem(f, noise_type=name))
            return chain

        x = torch.zeros(shape, device="cuda")
        runs = []
        for variant in ("hosted", "folded", "plain"):
            with monkeypatch.context() as m:
                if variant == "folded":
                    m.setattr(N.NoiseSampler, "accepts_prefix", property(lambda self: False))
                elif variant == "plain":
                    m.delattr(N.NoiseSampler, "accumulate")
                torch.manual_seed(1000 + it)
                with ng.shard_offset(offset):
                    ns = build().make_noise_sampler(x, 0.03, 14.6, seed=5 + it, cpu=False, normalized=False)
                    runs.append([ns(torch.tensor(s), torch.tensor(sn)).clone() for s, sn in ((10.0, 7.0), (7.0, 4.0))])
        for a, b_, c_ in zip(*runs):
            assert bool(torch.isfinite(a).all()) and torch.equal(a, b_) and torch.equal(b_, c_), (it, shape, items, factors, offset)


def test_brownian_shards_with_odd_latents(api):
    """Latents of 3 x 33 x 17 elements: a shard's first element need not be a multiple of four; the general Brownian kernel owns GLOBAL
    groups of four, so the shards of a batch still add up to the whole."""
    ng = api.noise_generation

    def run(b0, b):
        with ng.shard_offset(b0):
            x = torch.zeros(b, 3, 33, 17, device="cuda")
            ns = api.noise.get_noise_sampler("brownian", x, 0.03, 14.6, seed=11, cpu=False, normalized=False)
            return [ns(torch.tensor(s), torch.tensor(sn)).clone() for s, sn in ((10.0, 7.0), (7.0, 4.0), (7.0, 5.5))]

    whole = run(0, 5)
    for step, parts in enumerate(zip(run(0, 1), run(1, 2), run(3, 2))):
        assert torch.equal(torch.cat(parts), whole[step]), step


# ------------------------------------------------------------------------------------------------ noise whose normalisation rides in the step kernel
@pytest.mark.parametrize("case", ["shift_and_scale", "scale_only", "as_is"])
def test_step_kernels_apply_a_pending_normalisation_like_scale_noise(api, case):
    """sonar_norm_decision_f32 + the `noise_norm` argument of the three sampler-step kernels == scale_noise(normalized=True) first, then the
    plain kernels: bit for bit (same decision, same subtract / divide / multiply sequence).  Also sonar_apply_norm_f32."""
    hl = api.hl
    g = torch.Generator(device="cuda").manual_seed(12)
    shape = (4, 4, 64, 64)
    noise = torch.randn(shape, device="cuda", generator=g)
    if case == "shift_and_scale":
        noise = noise * 1.7 + 0.4
    elif case == "scale_only":
        noise = noise * 0.6
    else:
        noise = (noise - noise.mean()) / noise.std()
    factor = 1.0 if case == "as_is" else 1.3
    x, den, md1 = (torch.randn(shape, device="cuda", generator=g) for _ in range(3))
    n = noise.numel()
    want_noise = hl.scale_noise_(noise.clone(), factor, True, hl.stats(noise))
    norm = hl.norm_decision(hl.stats(noise), n, factor)
    assert torch.equal(hl.apply_norm_(noise.clone(), norm), want_noise)
    cfg = hl.MomentumCfg()
    cfg.momentum, cfg.hist_ratio, cfg.hist_scale, cfg.md_scale = 0.95, 0.75, 1.0, 1.0
    cfg.mode, cfg.use_momentum, cfg.update_hist, cfg.init_kind = 0, 1, 1, 0
    h = torch.randn(shape, device="cuda", generator=g)
    a = hl.momentum_euler(x, den, h, cfg, 3.0, -0.5, noise=noise, noise_scale=0.7, noise_norm=norm)
    b = hl.momentum_euler(x, den, h, cfg, 3.0, -0.5, noise=want_noise, noise_scale=0.7)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    a = hl.dpmpp_stage1(x, den, h, cfg, 3.0, -0.3, 0.8, False, noise=noise, noise_scale=0.7, noise_norm=norm)
    b = hl.dpmpp_stage1(x, den, h, cfg, 3.0, -0.3, 0.8, False, noise=want_noise, noise_scale=0.7)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    a = hl.dpmpp_stage2(x, den, md1, h, cfg, 2.0, -0.4, 0.6, 1.0, False, noise=noise, noise_scale=0.7, noise_norm=norm)
    b = hl.dpmpp_stage2(x, den, md1, h, cfg, 2.0, -0.4, 0.6, 1.0, False, noise=want_noise, noise_scale=0.7)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("sampler", ["dpmpp_sde", "euler_ancestral"])
def test_samplers_with_deferred_normalisation_equal_the_plain_run(api, sampler, monkeypatch):
    """A chain's noise handed to the step kernels with its normalisation pending (`deferred`) gives the same trajectory as the chain
    normalising first: cfg5's mix (power-law + Perlin + Brownian), 6 steps."""
    N, pn, S = api.noise, api.powernoise, api.sonar

    def run(defer):
        chain = N.CustomNoiseChain()
        chain.add(pn.PowerNoiseItem(0.5, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                                    common_mode=0.0, channel_correlation="1"))
        chain.add(N.CustomNoiseItem(0.3, noise_type="perlin"))
        chain.add(N.CustomNoiseItem(0.2, noise_type="brownian"))
        torch.manual_seed(31)
        x = torch.randn(2, 4, 64, 64, device="cuda") * 10
        sigmas = torch.cat([torch.linspace(14.6, 0.5, 7), torch.zeros(1)])
        ns = chain.make_noise_sampler(x, 0.5, 14.6, seed=3, cpu=False, normalized=True)
        assert hasattr(ns, "deferred")
        if not defer:
            plain = ns
            ns = lambda s, sn: plain(s, sn)  # noqa: E731  (no `deferred` attribute: the sampler takes the finished tensor)
        model = lambda t, sigma, **_k: api.hl.mul_scalar(t, 0.5)  # noqa: E731
        if sampler == "dpmpp_sde":
            return S.SonarDPMPPSDE.sampler(model, x, sigmas, {"seed": 3}, None, True, None, dict(momentum=0.95), 1.0, 1.0, ns)
        return S.SonarEulerAncestral.sampler(model, x, sigmas, {"seed": 3}, None, True, None, dict(momentum=0.95), 1.0, 1.0, ns)

    a, b = run(True), run(False)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_perlin_fast_path_equals_the_general_kernel(api):
    """perlin_generate_kernel's fast path (one summed term table, tile-aligned latents: term vectors requested four iterations ahead)
    against the general loop, reached with a second, all-zero term table: raw values + statistics, the statistics-only pass and the
    normalising pass, and the folding form."""
    hl = api.hl
    _perlin_fast_vs_general(hl, (6, 4, 64, 64), 4)      # latents of whole RNG tiles
    _perlin_fast_vs_general(hl, (5, 4, 104, 152), 3)    # 63232 elements per latent: tiles straddle latents, the shard ends inside a tile
    _perlin_fast_vs_general(hl, (3, 2, 8, 8), 1)        # latents smaller than one 256-element step


def _perlin_fast_vs_general(hl, shape, first_latent):
    c, h, w = shape[1:]
    terms = hl.perlin_lattice(2, c, h, w, "cuda", "lerp", 99, 3)
    two = torch.cat([terms, torch.zeros_like(terms)]).contiguous()
    offs = first_latent * c * h * w
    pa, pb = hl.new_partials("cuda"), hl.new_partials("cuda")
    a = hl.perlin_generate(shape, terms, 2.0, 7, 5, offs, pa)
    b = hl.perlin_generate(shape, two, 2.0, 7, 5, offs, pb)
    # values bit-equal; the fast path sums a tile's 64 values per lane in fp32 before its fp64 partials: statistics agree to ~1e-7
    assert torch.equal(a, b)
    torch.testing.assert_close(pa.view(-1, 2).sum(0), pb.view(-1, 2).sum(0), rtol=2e-6, atol=0)
    torch.testing.assert_close(hl.perlin_noise(shape, terms, 2.0, 7, 5, offs, 1.25), hl.perlin_noise(shape, two, 2.0, 7, 5, offs, 1.25), rtol=1e-6, atol=1e-6)
    y = torch.randn(shape, device="cuda")
    assert torch.equal(hl.perlin_generate_acc_(y.clone(), 0.5, 0.3, terms, 2.0, 7, 5, offs), hl.perlin_generate_acc_(y.clone(), 0.5, 0.3, two, 2.0, 7, 5, offs))


def test_gaussian_normalised_single_pass(api):
    """sonar_philox_noise_f32 for N(0,1) with factor 1: one pass stores the draws and their statistics, scale_noise's kernel decides on
    the device.  Same values as draw + scale_noise; with the default thresholds the draws of a large tensor come back untouched, with
    zero thresholds they are shifted and scaled to mean 0 / std 1; other factors keep the two-pass (draw twice, write once) route."""
    hl = api.hl
    shape = (64, 4, 128, 128)
    offs = 3 * 4 * 128 * 128
    raw = hl.philox_normal(shape, "cuda", 11, 5, offs)
    for thr in (2.5, 0.0):
        got = hl.philox_noise(False, shape, "cuda", 11, 5, offs, 1.0, threshold_std_devs=thr)
        want = hl.scale_noise_(raw.clone(), 1.0, True, hl.stats(raw), threshold_std_devs=thr)
        torch.testing.assert_close(got, want, rtol=0, atol=1e-6)
        if thr == 0.0:
            assert abs(got.mean().item()) < 1e-6 and abs(got.std().item() - 1.0) < 1e-6 and not torch.equal(got, raw)
    two_pass = hl.philox_noise(False, shape, "cuda", 11, 5, offs, 1.5, threshold_std_devs=0.0)
    torch.testing.assert_close(two_pass, hl.scale_noise_(raw.clone(), 1.5, True, hl.stats(raw), threshold_std_devs=0.0), rtol=1e-6, atol=1e-6)


def test_euler_ancestral_with_pyramid_noise_deferred_equals_plain(api):
    """A single pyramid generator behind ``get_noise_sampler(normalized=True)``: the sampler step applies the normalisation its fused
    path would have spent a pass on; same trajectory."""
    N, S = api.noise, api.sonar

    def run(defer):
        torch.manual_seed(41)
        x = torch.randn(4, 4, 64, 64, device="cuda") * 10
        sigmas = torch.cat([torch.linspace(14.6, 0.5, 6), torch.zeros(1)])
        ns = N.get_noise_sampler("pyramid", x, 0.5, 14.6, seed=3, cpu=False, normalized=True)
        assert hasattr(ns, "deferred")
        if defer:
            _noise, norm = ns.deferred(torch.tensor(14.6), torch.tensor(10.0))
            assert norm is not None  # the pyramid path really defers
            torch.manual_seed(41)
            x = torch.randn(4, 4, 64, 64, device="cuda") * 10
            ns = N.get_noise_sampler("pyramid", x, 0.5, 14.6, seed=3, cpu=False, normalized=True)
        else:
            inner = ns
            ns = lambda s, sn: inner(s, sn)  # noqa: E731
        return S.SonarEulerAncestral.sampler(lambda t, sigma, **_k: api.hl.mul_scalar(t, 0.5), x, sigmas, {"seed": 3}, None, True, None,
                                             dict(momentum=0.95), 1.0, 1.0, ns)

    a, b = run(True), run(False)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
