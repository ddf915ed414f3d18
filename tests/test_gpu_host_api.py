"""Drop-in API parity (-m gpu): the reference's plugin interface (get_noise_sampler / CustomNoiseChain /
PowerNoiseItem / Sonar*.sampler) driven exactly like the reference's callers drive it, with the latent on
the MI355X, against golden outputs captured from the real reference (tests/golden/make_golden.py).

Replay mode (``cpu=True``, the reference default): base draws come from the torch CPU generator in the
reference's order, so outputs must match the reference's CPU path within fp32 tolerance:
  generators / composition: rtol 2e-5, atol 5e-6;  power noise (FFT): atol 4e-5;
  multi-step samplers (error accumulates over 7 steps incl. a tanh model): rtol 1e-4, atol 1e-4.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

SIG = (torch.tensor(14.6), torch.tensor(10.0))


@pytest.fixture(scope="module")
def api(pkg):
    import importlib

    pkg.hip_lib.load()
    mods = {m: importlib.import_module(f"comfyui_sonar_amd.py.{m}") for m in ("utils", "noise_generation", "noise", "sonar")}
    mods["powernoise"] = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
    import types

    return types.SimpleNamespace(**mods, hl=pkg.hip_lib)


def close(a, b, rtol=2e-5, atol=5e-6):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), rtol=rtol, atol=atol)


def run_type(api, name, shape, seed, normalized, **kw):
    x = torch.zeros(shape, device="cuda")
    torch.manual_seed(seed)
    ns = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=seed, cpu=True, factor=1.0, normalized=normalized, **kw)
    out = ns(*SIG)
    assert out.is_cuda and out.shape == x.shape and out.dtype == torch.float32
    return out


# ------------------------------------------------------------------------------------------------ registry types
@pytest.mark.parametrize("name", ["gaussian", "uniform"])
@pytest.mark.parametrize("normalized", [False, True])
def test_basic_types(api, golden, name, normalized):
    g = golden("basic_types")
    close(run_type(api, name, (2, 4, 8, 8), 21, normalized), g[f"{name}_{int(normalized)}"])


@pytest.mark.parametrize("name", ["laplacian", "power_old"])
@pytest.mark.parametrize("normalized", [False, True])
def test_laplacian_and_power_old_types(api, golden, name, normalized):
    """py/noise_generation.py:789-802 and 1259-1287 against the reference's outputs (replay), plus generate-mode properties."""
    g = golden("basic_types")
    close(run_type(api, name, (3, 4, 8, 8), 22, normalized), g[f"{name}_{int(normalized)}"], rtol=2e-5, atol=2e-5)
    x = torch.zeros(64, 4, 64, 64, device="cuda")
    out = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=1, cpu=False, normalized=False)(*SIG)
    if name == "laplacian":  # var = 1/16 + 2 scale^2, excess kurtosis well above a normal's
        v = out.double().var().item()
        assert abs(v - (1.0 / 16 + 2.0)) < 0.02 and abs(out.mean().item()) < 5e-3
        assert (out.double() ** 4).mean().item() / v**2 > 4.5
    else:  # every plane standardised
        planes = out.reshape(-1, 64 * 64).double()
        assert planes.mean(dim=1).abs().max().item() < 1e-5 and (planes.std(dim=1) - 1.0).abs().max().item() < 1e-5


def test_studentt_type(api, golden):
    """py/noise_generation.py:652-677 against the reference (replay: normal + torch._standard_gamma base draws, per-latent quantile clamp,
    sign-preserving power), the quantile kernel against torch.quantile at SDXL size, and the on-device draws' distribution."""
    g = golden("basic_types")
    for normalized in (False, True):
        close(run_type(api, "studentt", (3, 4, 8, 8), 23, normalized), g[f"studentt_{int(normalized)}"], rtol=2e-5, atol=2e-5)
    close(run_type(api, "studentt", (3, 4, 8, 8), 23, False, df=3, quantile_fac=0.9, pow_fac=0.75, scale=0.5, loc=0.1, nq_fac=0.8),
          g["studentt_df3"], rtol=2e-5, atol=2e-5)
    torch.manual_seed(5)
    x = torch.randn(6, 4 * 128 * 128)
    x[1, :1000] = 0.25  # repeated values around a rank
    for q in (0.75, 0.5, 0.999, 0.0, 1.0, 0.123):
        want = torch.quantile(x.abs(), q, dim=-1)
        got = api.hl.abs_quantile_rows(x.cuda(), 6, x.shape[1], q)
        close(got, want, rtol=1e-6, atol=1e-7)
    xs = torch.zeros(32, 4, 64, 64, device="cuda")
    out = api.noise.get_noise_sampler("studentt", xs, 0.03, 14.6, seed=1, cpu=False, normalized=False)(*SIG)
    # |t_1| * 0.2 clamped at its 75 % quantile (0.2 * tan(3 pi / 8) = 0.483), then sqrt: a quarter of the values sit at +-sqrt(0.483)
    lim = (0.2 * math.tan(3 * math.pi / 8)) ** 0.5
    assert abs(out.abs().max().item() - lim) < 0.02 * lim
    at_lim = (out.abs() > 0.999 * out.abs().amax(dim=(1, 2, 3), keepdim=True)).float().mean().item()
    assert abs(at_lim - 0.25) < 0.01 and abs(out.mean().item()) < 5e-3
    with pytest.raises(NotImplementedError):
        api.noise.get_noise_sampler("studentt", xs, 0.03, 14.6, seed=1, cpu=False, normalized=False, df=2.5)(*SIG)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_perlin_type(api, golden, tag):
    g = golden("perlin")
    shape = tuple(g[f"{tag}_base"].shape)
    kw = {} if str(g[f"{tag}_blend"]) == "lerp" else {"blend_mode": str(g[f"{tag}_blend"])}
    close(run_type(api, "perlin", shape, int(g[f"{tag}_seed"]), False, **kw), g[f"{tag}_raw"])
    close(run_type(api, "perlin", shape, int(g[f"{tag}_seed"]), True, **kw), g[f"{tag}_out"])


@pytest.mark.parametrize("tag,name", [("a", "pyramid"), ("b", "pyramid"), ("c", "pyramid_discount5"), ("d", "pyramid_area")])
def test_pyramid_type(api, golden, tag, name):
    g = golden("pyramid")
    shape = tuple(g[f"{tag}_base"].shape)
    close(run_type(api, name, shape, int(g[f"{tag}_seed"]), False), g[f"{tag}_raw"])
    close(run_type(api, name, shape, int(g[f"{tag}_seed"]), True), g[f"{tag}_out"])


def test_noise_type_registry_is_complete(api):
    NT = api.noise_generation.NoiseType
    assert len(NT) == 38 and set(api.noise.NOISE_SAMPLERS) == set(NT)
    assert next(NT.get_names()) == "gaussian"
    with pytest.raises(NotImplementedError):
        api.noise.get_noise_sampler("collatz", torch.zeros(1, 4, 8, 8, device="cuda"), 0.1, 1.0)
    with pytest.raises(ValueError):
        api.noise.get_noise_sampler("brownian", torch.zeros(1, 4, 8, 8, device="cuda"), None, None)


def test_cpu_latent_fails_loudly(api):
    with pytest.raises(api.hl.SonarHipError):
        api.noise.get_noise_sampler("gaussian", torch.zeros(1, 4, 8, 8), 0.1, 1.0)


# ------------------------------------------------------------------------------------------------ power noise
POWER_KW = {
    "cfg2": {},
    "b": {"alpha": 0.5},
    "c": {"alpha": 2.0, "common_mode": 0.25},
    "d": {"alpha": 1.0, "min_freq": 0.1, "max_freq": 0.4},
    "e": {"alpha": 0.0, "mix": 0.5},
    "np2": {"alpha": 1.0},  # 40 x 56 and 52 x 76 planes: the general-size kernels
    "np2_rot": {"alpha": 1.5, "rotate": 20.0, "stretch": 1.5, "common_mode": 0.1},
    "odd": {"alpha": 1.0, "common_mode": 0.1},  # 27 x 35: the direct DFT passes
}


def power_item(api, **kw):
    args = dict(time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                common_mode=0.0, channel_correlation="1,1,1,1,1,1")
    args.update(kw)
    return api.powernoise.PowerNoiseItem(1.0, **args)


@pytest.mark.parametrize("tag", list(POWER_KW))
def test_power_noise_item(api, golden, tag):
    g = golden("power_noise")
    item = power_item(api, **POWER_KW[tag])
    shape = tuple(g[f"{tag}_out"].shape)
    x = torch.zeros(shape, device="cuda")
    torch.manual_seed(int(g[f"{tag}_seed"]))
    ns = item.make_noise_sampler(x, None, None, seed=None, cpu=True, normalized=bool(g[f"{tag}_normalized"]))
    close(ns(None, None), g[f"{tag}_out"], rtol=0, atol=4e-5)
    # host-built filter: same op sequence as the reference (bit-identical on the machine that made the
    # fixtures; other CPUs may differ in the last bit of exp/pow)
    torch.testing.assert_close(item.make_filter(shape), g[f"{tag}_filter"], rtol=2e-6, atol=1e-30)


def test_power_filter_and_mixer_host_setup(api, golden):
    g = golden("power_filter")
    PF = api.powernoise.PowerFilter
    cases = {"white": dict(alpha=0.0), "pink": dict(alpha=1.0), "band": dict(alpha=1.0, min_freq=0.1, max_freq=0.4),
             "rot_stretch": dict(alpha=1.0, rotate=30.0, stretch=2.0), "pnorm1": dict(alpha=1.0, pnorm=1.0)}
    for name, kw in cases.items():
        torch.testing.assert_close(PF(**kw).build((1, 4, 32, 32)), g[f"{name}_32x32_raw"], rtol=2e-6, atol=1e-30)
    m = api.powernoise.ChannelMixer(4, 0.25, torch.ones(6)).mixer
    torch.testing.assert_close(m, g["mixer_0.25"], rtol=1e-6, atol=1e-7)


def test_power_noise_device_mode_statistics_and_shards(api):
    """cpu=False: spectrum drawn in-kernel.  Check unit variance, the 1/f spectral slope, and that a batch
    generated as two shards equals the batch generated at once."""
    item = power_item(api)
    shape = (8, 4, 128, 128)
    x = torch.zeros(shape, device="cuda")
    torch.manual_seed(123)
    full = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=False)(None, None)
    spec = torch.fft.rfft2(full, norm="ortho").abs().square().mean(dim=(0, 1)).cpu()
    filt = item.make_filter(shape)[0, 0]
    ratio = (spec[1:40, 1:40] / filt[1:40, 1:40].square()).mean().item()
    assert abs(ratio - 1.0) < 0.1  # E|Z f|^2 = f^2 for unit complex normal Z
    parts = []
    for b0 in (0, 4):
        with api.noise_generation.shard_offset(b0):
            xs = torch.zeros((4, *shape[1:]), device="cuda")
            ns = item.make_noise_sampler(xs, None, None, seed=None, cpu=False, normalized=False)
            torch.manual_seed(123)  # every "rank" seeds alike -> same stream ids
            parts.append(ns(None, None))
    torch.manual_seed(123)
    whole = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=False)(None, None)
    assert torch.equal(torch.cat(parts), whole)
    normed = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)(None, None)
    assert abs(normed.std().item() - 1.0) < 1e-4 and abs(normed.mean().item()) < 5e-3


# ------------------------------------------------------------------------------------------------ composition
def chain_of(api, *items):
    c = api.noise.CustomNoiseChain()
    for it in items:
        c.add(it)
    return c


def item(api, name, f):
    return api.noise.CustomNoiseItem(f, noise_type=api.noise_generation.NoiseType[name.upper()])


def test_chain_and_rescaled(api, golden):
    g = golden("composition")
    x = torch.zeros(2, 4, 16, 16, device="cuda")
    chain = chain_of(api, item(api, "gaussian", 0.6), item(api, "uniform", -0.3), item(api, "perlin", 0.5))
    assert chain.factor == pytest.approx(1.4)
    for tag, ch in (("chain", chain), ("chain_rescaled", chain.rescaled(1.0))):
        torch.manual_seed(31)
        out = ch.make_noise_sampler(x, 0.03, 14.6, seed=31, cpu=True, normalized=True)(torch.tensor(5.0), torch.tensor(4.0))
        close(out, g[f"{tag}_out"])
        assert [i.factor for i in ch.items] == pytest.approx(list(g[f"{tag}_factors"]))


def test_composite(api, golden):
    g = golden("composition")
    x = torch.zeros(2, 4, 16, 16, device="cuda")
    comp = api.noise.CompositeNoise(0.8, dst_noise=chain_of(api, item(api, "gaussian", 1.0)), src_noise=chain_of(api, item(api, "uniform", 1.0)),
                                    normalize_dst=True, normalize_src=True, normalize_result=True, mask=g["comp_mask"])
    torch.manual_seed(32)
    out = comp.make_noise_sampler(x, 0.03, 14.6, seed=32, cpu=True, normalized=True)(torch.tensor(5.0), torch.tensor(4.0))
    close(out, g["comp_out"])


def test_blended(api, golden):
    g = golden("composition")
    x = torch.zeros(2, 4, 16, 16, device="cuda")
    bl = api.noise.BlendedNoise(1.2, normalize=True, blend_function=api.utils.BLENDING_MODES["lerp"],
                                custom_noise_1=chain_of(api, item(api, "gaussian", 1.0)), custom_noise_2=chain_of(api, item(api, "uniform", 1.0)),
                                noise_2_percent=0.3)
    torch.manual_seed(33)
    close(bl.make_noise_sampler(x, 0.03, 14.6, seed=33, cpu=True, normalized=True)(torch.tensor(5.0), torch.tensor(4.0)), g["blend_out"])
    blm = api.noise.BlendedNoise(1.0, normalize=True, blend_function=api.utils.BLENDING_MODES["inject"],
                                 custom_noise_1=chain_of(api, item(api, "gaussian", 1.0)), custom_noise_2=chain_of(api, item(api, "uniform", 1.0)),
                                 custom_noise_mask=chain_of(api, item(api, "gaussian", 1.0)), noise_2_percent=0.1)
    torch.manual_seed(34)
    close(blm.make_noise_sampler(x, 0.03, 14.6, seed=34, cpu=True, normalized=True)(torch.tensor(5.0), torch.tensor(4.0)), g["blendmask_out"])


def test_scheduled(api, golden):
    g = golden("composition")
    x = torch.zeros(2, 4, 16, 16, device="cuda")
    sch = api.noise.ScheduledNoise(1.0, noise=chain_of(api, item(api, "perlin", 1.0)), start_sigma=10.0, end_sigma=2.0, normalize=True,
                                   fallback_noise=chain_of(api, item(api, "gaussian", 1.0)))
    ns = sch.make_noise_sampler(x, 0.03, 14.6, seed=35, cpu=True, normalized=True)
    torch.manual_seed(35)
    close(ns(torch.tensor(5.0), torch.tensor(4.0)), g["sched_in"])
    close(ns(torch.tensor(12.0), torch.tensor(11.0)), g["sched_out"])
    with pytest.raises(ValueError):
        ns(None, None)
    # no fallback + normalisation outside the window -> NaN, like the reference
    bare = api.noise.ScheduledNoise(1.0, noise=chain_of(api, item(api, "gaussian", 1.0)), start_sigma=10.0, end_sigma=2.0, normalize=True)
    out = bare.make_noise_sampler(x, 0.03, 14.6, seed=1, cpu=True, normalized=True)(torch.tensor(12.0), torch.tensor(11.0))
    assert torch.isnan(out).all()


# ------------------------------------------------------------------------------------------------ samplers
def fake_model(x, sigma, **_kw):
    s = sigma.reshape(-1, *([1] * (x.ndim - 1)))
    return x * 0.5 + torch.tanh(x) * (0.1 * s / (1.0 + s))


MOMENTUM_CASES = {
    "new_default": dict(),
    "classic": dict(momentum_mode="CLASSIC"),
    "denoised": dict(momentum_mode="DENOISED"),
    "new_negdir": dict(direction=-0.5),
    "classic_dir15": dict(momentum_mode="CLASSIC", momentum=0.8, momentum_hist=0.5, direction=1.5),
    "denoised_sample": dict(momentum_mode="DENOISED", init="SAMPLE"),
    "new_sample_norm": dict(init="SAMPLE_NORM"),
    "new_sample": dict(init="SAMPLE", momentum=0.7),
    "steps_gated": dict(momentum_start_step=2, momentum_end_step=4, always_update_history=False),
    "steps_gated_hist": dict(momentum_start_step=2, momentum_end_step=4),
    "inject_blend": dict(blend_mode="inject", momentum=0.3, momentum_hist=0.4),
    "mixed_blend": dict(momentum_blend_mode="subtract_b", history_blend_mode="inject", momentum=0.2, momentum_hist=0.3),
    "no_momentum": dict(momentum=1.0),
    "hist_frozen": dict(momentum_hist=1.0, init="SAMPLE"),
    "low_weight": dict(momentum=0.4, momentum_hist=0.2),
}


@pytest.mark.parametrize("kind", ["euler", "ancestral", "dpmpp"])
@pytest.mark.parametrize("name", list(MOMENTUM_CASES))
def test_sampler_traces(api, golden, kind, name):
    g = golden("momentum")
    S = api.sonar
    x0, sigmas, bank = g["x0"].cuda(), g["sigmas"], g["noise_bank"]
    it = iter(bank)

    def ns(_s, _sn):
        return next(it).cuda()

    trace = []
    cb = lambda d: trace.append(d["x"].clone())  # noqa: E731
    extra = {"seed": 0}
    kw = dict(MOMENTUM_CASES[name])
    if kind == "euler":
        out = S.SonarEuler.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, ns, None, kw)
    elif kind == "ancestral":
        out = S.SonarEulerAncestral.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, None, kw, 0.8, 1.1, ns)
    else:
        out = S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, None, kw, 0.9, 1.05, ns)
    want = g[f"{kind}_{name}"]
    assert len(trace) == want.shape[0]
    for i, t in enumerate(trace):
        close(t, want[i], rtol=1e-4, atol=1e-4)
    close(out, want[-1], rtol=1e-4, atol=1e-4)


def test_unfused_momentum_api_matches_fused(api):
    """get_momentum_denoised / get_momentum_d (reference signature) give the same step as the fused kernel."""
    S = api.sonar
    torch.manual_seed(2)
    x = torch.randn(2, 4, 8, 8, device="cuda") * 4
    for mode in ("NEW", "CLASSIC", "DENOISED"):
        cfg = S.SonarBase.get_config(None, {"momentum_mode": mode, "init": "SAMPLE_NORM"})
        a, b = S.SonarBase(cfg), S.SonarBase(cfg)
        xa = xb = x
        for step, (s, sd) in enumerate(((7.0, 4.0), (4.0, 2.0), (2.0, 1.0))):
            den = fake_model(xa, torch.tensor(s, device="cuda"))
            xa = a.momentum_step(step, xa, den, torch.tensor(s), torch.tensor(sd))
            den_b = fake_model(xb, torch.tensor(s, device="cuda"))
            dm = b.get_momentum_denoised(xb, den_b, s, step=step)
            md = b.get_momentum_d(xb, dm, s, step=step)
            xb = api.hl.axpby_(api.hl.mul_scalar(md, sd - s), 1.0, xb, 1.0)
            close(xa, xb, rtol=1e-5, atol=1e-5)
            close(a.history_d, b.history_d, rtol=1e-5, atol=1e-5)


def test_momentum_one_is_plain_euler(api):
    """README.md:50: momentum = 1 disables momentum -> plain Euler."""
    S = api.sonar
    torch.manual_seed(4)
    x = torch.randn(1, 4, 16, 16, device="cuda") * 10
    sigmas = torch.tensor([10.0, 6.0, 3.0, 1.0, 0.0])
    out = S.SonarEuler.sampler(fake_model, x.clone(), sigmas, {"seed": 0}, None, True, None, None, {"momentum": 1.0})
    ref = x.clone()
    for i in range(4):
        den = fake_model(ref, sigmas[i].cuda() * torch.ones(1, device="cuda"))
        ref = ref + (ref - den) / sigmas[i].item() * (sigmas[i + 1] - sigmas[i]).item()
    close(out, ref, rtol=1e-5, atol=1e-5)


def test_config_fixups(api):
    S = api.sonar
    cfg = S.SonarBase.get_config(None, {"momentum_mode": " classic ", "init": "rand", "noise_type": "perlin"})
    assert cfg.momentum_mode == S.MomentumMode.CLASSIC and cfg.init == S.HistoryType.RAND
    with pytest.raises(ValueError):
        S.SonarBase.get_config(None, {"momentum_mode": "nope"})
    with pytest.raises(TypeError):
        S.SonarBase.get_config(None, {"init": 3})


# ------------------------------------------------------------------------------------------------ generate mode
@pytest.mark.parametrize("name", ["gaussian", "uniform", "perlin", "pyramid"])
def test_device_mode_distribution_and_shard_invariance(api, name):
    shape = (8, 4, 64, 64)
    NG = api.noise_generation

    def gen(b0, b):
        torch.manual_seed(77)
        with NG.shard_offset(b0):
            x = torch.zeros((b, *shape[1:]), device="cuda")
            ns = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=False)
            return ns(*SIG)

    whole = gen(0, 8)
    assert torch.equal(torch.cat([gen(0, 3), gen(3, 5)]), whole)
    torch.manual_seed(77)
    x = torch.zeros(shape, device="cuda")
    normed = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=True)(*SIG)
    thr = 2.5 / math.sqrt(normed.numel())  # scale_noise leaves statistics inside the threshold untouched
    assert abs(normed.std().item() - 1.0) < thr + 1e-4 and abs(normed.mean().item()) < thr + 1e-4
    # successive calls differ (stream ids advance); reseeding reproduces
    ns = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=False)
    a, b = ns(*SIG), ns(*SIG)
    assert not torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ entry-point nodes
def test_noisy_latent_like_and_noise_adapter_nodes(pkg, api, golden):
    """NoisyLatentLike.go / CustomNOISE.generate_noise (py/nodes/misc.py:73-155,360-419) vs the reference, incl. repeat_batch,
    add_to_latent, the saved/restored global RNG state and per-batch-index seeding."""
    g = golden("entry_nodes")
    M = pkg.NODE_CLASS_MAPPINGS
    before = torch.random.get_rng_state()
    (out,) = M["NoisyLatentLike"].go(noise_type="perlin", seed=5, latent={"samples": torch.zeros(2, 4, 16, 16)}, multiplier=0.7,
                                     add_to_latent=False, repeat_batch=2, cpu_noise=True, normalize=True)
    assert torch.equal(torch.random.get_rng_state(), before)
    assert out["samples"].device.type == "cpu"  # returned on the latent's original device
    close(out["samples"], g["nll_perlin"])
    chain = chain_of(api, item(api, "gaussian", 0.6), item(api, "uniform", -0.3))
    (out,) = M["NoisyLatentLike"].go(noise_type="gaussian", seed=6, latent={"samples": g["nll_chain_latent"]}, multiplier=1.3, add_to_latent=True,
                                     repeat_batch=1, cpu_noise=True, normalize=True, custom_noise_opt=chain)
    close(out["samples"], g["nll_chain_out"])
    (nobj,) = M["SONAR_CUSTOM_NOISE to NOISE"].go(custom_noise=chain, seed=9, cpu_noise=True, normalize=True, multiplier=0.5)
    close(nobj.generate_noise({"samples": torch.zeros(3, 4, 8, 8)}), g["noise_plain"])
    close(nobj.generate_noise({"samples": torch.zeros(3, 4, 8, 8), "batch_index": [2, 0, 2]}), g["noise_batch_index"])
    zero = M["SONAR_CUSTOM_NOISE to NOISE"].go(custom_noise=chain, seed=9, multiplier=0.0)[0].generate_noise({"samples": torch.zeros(1, 4, 8, 8)})
    assert torch.count_nonzero(zero) == 0


def test_sampler_config_override_node(pkg, api, golden):
    """SamplerConfigOverride (py/nodes/misc.py:461-619): wraps a SAMPLER; eta / s_noise and the noise source are replaced, parameters
    the wrapped sampler function does not take are dropped."""
    g = golden("momentum")
    M = pkg.NODE_CLASS_MAPPINGS
    S = api.sonar
    x0, sigmas = g["x0"].cuda(), g["sigmas"]
    (inner,) = M["SamplerSonarEulerA"].get_sampler(momentum=0.95, momentum_hist=0.75, momentum_init="ZERO", direction=1.0, rand_init_noise_type="gaussian",
                                                   noise_type="gaussian", eta=1.0, s_noise=1.0)
    chain = chain_of(api, item(api, "uniform", 1.0))
    (wrapped,) = M["SamplerConfigOverride"]().get_sampler(sampler=inner, eta=0.5, s_noise=1.2, s_churn=0.3, r=0.25, sde_solver="heun", cpu_noise=True,
                                                          noise_type="DEFAULT", custom_noise_opt=chain, normalize=True, yaml_parameters="eta: 0.6")
    assert wrapped.extra_options == inner.extra_options and wrapped.extra_options is not inner.extra_options
    torch.manual_seed(11)
    out = wrapped.sampler_function(fake_model, x0.clone(), sigmas, extra_args={"seed": 3}, callback=None, disable=True, **wrapped.extra_options)
    torch.manual_seed(11)
    ns = chain.make_noise_sampler(x0, sigmas[sigmas > 0].min(), sigmas.max(), seed=3, cpu=True, normalized=True)
    want = S.SonarEulerAncestral.sampler(fake_model, x0.clone(), sigmas, {"seed": 3}, None, True, sonar_config=inner.extra_options["sonar_config"],
                                         eta=0.6, s_noise=1.2, noise_sampler=ns)
    close(out, want, rtol=0, atol=0)
    # a sampler function without a noise_sampler parameter only receives the kwargs it declares
    seen = {}

    def plain(model, x, sigmas, extra_args=None, s_churn=0.0):
        seen.update(s_churn=s_churn, extra_args=extra_args)
        return x

    import types
    carrier = types.SimpleNamespace(sampler_function=plain, extra_options={}, inpaint_options={})
    (w2,) = M["SamplerConfigOverride"]().get_sampler(sampler=carrier, eta=0.5, s_noise=1.2, s_churn=0.3, r=0.25, sde_solver="heun", noise_type="perlin")
    assert w2.sampler_function(None, x0, sigmas) is x0 and seen == {"s_churn": 0.3, "extra_args": {}}
    with pytest.raises(ValueError):
        M["SamplerConfigOverride"]().get_sampler(sampler=carrier, eta=0.5, s_noise=1.2, s_churn=0.3, r=0.25, sde_solver="heun", yaml_parameters="[1, 2]")


def test_global_normalisation_and_sharded_sampler_single_process(api):
    par = __import__("importlib").import_module("comfyui_sonar_amd.parallel")
    torch.manual_seed(3)
    x = torch.randn(4, 4, 16, 16) * 0.7 + 0.3
    from oracle import sonar_oracle as orc

    want = orc.scale_noise(x.clone(), 1.1, normalized=True)
    close(par.normalise_global_(x.cuda(), 1.1), want)
    sh = par.ShardedNoiseSampler(lambda xs, **kw: api.noise.get_noise_sampler("perlin", xs, 0.03, 14.6, **kw), (6, 4, 32, 32), "cuda",
                                 seed=None, cpu=False, normalized=False)
    out = sh(*SIG)
    assert out.shape == (6, 4, 32, 32) and torch.equal(sh.gather(out), out)


# ------------------------------------------------------------------------------------------------ cfg5 in miniature, end to end
def test_cfg5_scheduled_power_perlin_chain_with_dpmpp_momentum(api, golden):
    """BASELINE.json cfg5 at test size, driven exactly like ComfyUI drives the reference: a chain of ScheduledNoise(power-law
    noise, gaussian fallback below sigma 4) + Perlin, normalised, as the noise sampler of SonarDPMPPSDE with momentum on a
    Flux-shaped (16-channel) latent.  Golden: the REAL reference end to end (tests/golden/make_golden.py gen_cfg5).
    6 sampler steps with 2 noise calls each: tolerance rtol/atol 3e-4 on |x| ~ 10."""
    g = golden("cfg5")
    N, S, pn = api.noise, api.sonar, api.powernoise
    x0, sigmas = g["x0"].cuda(), g["sigmas"]
    inner = N.CustomNoiseChain()
    inner.add(pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                                mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1"))
    fallback = N.CustomNoiseChain()
    fallback.add(N.CustomNoiseItem(1.0, noise_type="gaussian"))
    chain = N.CustomNoiseChain()
    chain.add(N.ScheduledNoise(0.7, noise=inner, start_sigma=20.0, end_sigma=4.0, normalize=None, fallback_noise=fallback))
    chain.add(N.CustomNoiseItem(0.5, noise_type="perlin"))
    params = dict(momentum=0.9, momentum_hist=0.7, direction=1.0, momentum_mode="NEW", init="SAMPLE_NORM")
    torch.manual_seed(62)
    ns = chain.make_noise_sampler(x0, sigmas[sigmas > 0].min(), sigmas.max(), seed=5, cpu=True, normalized=True)
    trace = []
    out = S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas, {"seed": 5}, lambda d: trace.append(d["x"].clone()), True, None,
                                  dict(params), 0.9, 1.05, ns)
    for row, step in enumerate(g["trace_steps"].tolist()):
        close(trace[step], g["trace"][row], rtol=3e-4, atol=3e-4)
    close(out, g["out"], rtol=3e-4, atol=3e-4)


# ------------------------------------------------------------------------------------------------ guidance
GUIDANCE_CASES = {
    "linear": dict(guidance_type="LINEAR", factor=0.05, start_step=1, end_step=4),
    "euler": dict(guidance_type="EULER", factor=0.2, start_step=0, end_step=9999),
    "linear_inject": dict(guidance_type="LINEAR", factor=0.1, start_step=2, end_step=5, guidance_blend_mode="inject"),
}


def test_guidance_building_blocks(api, golden):
    g = golden("guidance")
    S = api.sonar
    x0, ref = g["x0"].cuda(), g["ref_latent"].cuda()
    prepared = S.SonarGuidanceMixin.prepare_ref_latent(ref.clone())
    close(prepared, g["prepared_ref"], rtol=2e-5, atol=2e-5)
    close(S.SonarGuidanceMixin.guidance_shift(x0, prepared), g["shift"], rtol=2e-5, atol=2e-4)
    close(S.SonarGuidanceMixin.guidance_euler(torch.tensor(7.0), torch.tensor(5.0), x0, x0 * 0.5, prepared, 0.3), g["euler_step"], rtol=2e-5, atol=2e-4)
    close(S.SonarGuidanceMixin.guidance_linear(x0, prepared, 0.3), g["linear_step"], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("kind", ["euler", "ancestral", "dpmpp"])
@pytest.mark.parametrize("name", list(GUIDANCE_CASES))
def test_sampler_traces_with_guidance(api, golden, kind, name):
    """Reference samplers with a guidance latent (py/sonar.py:343-411 inside every step); golden = per-step x of the real reference."""
    g = golden("guidance")
    S = api.sonar
    x0, sigmas, bank = g["x0"].cuda(), g["sigmas"], g["noise_bank"]
    it = iter(bank)
    ns = lambda _s, _sn: next(it).cuda()  # noqa: E731
    kw = dict(GUIDANCE_CASES[name])
    blend = kw.pop("guidance_blend_mode", None)
    gcfg = S.GuidanceConfig(guidance_type=S.GuidanceType[kw.pop("guidance_type")], latent=g["ref_latent"].clone(), **kw)
    params = {"guidance": gcfg, "momentum": 0.9}
    if blend:
        params["guidance_blend_mode"] = blend
    trace = []
    cb = lambda d: trace.append(d["x"].clone())  # noqa: E731
    if kind == "euler":
        S.SonarEuler.sampler(fake_model, x0.clone(), sigmas, {"seed": 0}, cb, True, ns, None, params)
    elif kind == "ancestral":
        S.SonarEulerAncestral.sampler(fake_model, x0.clone(), sigmas, {"seed": 0}, cb, True, None, params, 0.8, 1.1, ns)
    else:
        S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas, {"seed": 0}, cb, True, None, params, 0.9, 1.05, ns)
    want = g[f"{kind}_{name}"]
    assert len(trace) == want.shape[0]
    for i, t in enumerate(trace):
        close(t, want[i], rtol=2e-4, atol=2e-4)


# ------------------------------------------------------------------------------------------------ 5-D (video) latents
@pytest.mark.parametrize("name", ["gaussian", "perlin", "pyramid", "pyramid_area", "onef_pinkish", "green_test", "velvet"])
@pytest.mark.parametrize("normalized", [False, True])
def test_video_latents(api, golden, name, normalized):
    """[B, C, F, H, W]: golden from the real reference (frames folded into channels where the generator needs 4-D)."""
    g = golden("video")
    want = g[f"{name}_{int(normalized)}"]
    out = run_type(api, name, tuple(want.shape), 81, normalized)
    tol = 3e-5 * float(want.abs().max()) if name in ("onef_pinkish", "green_test") else 5e-6
    close(out, want, rtol=2e-5, atol=tol)


def test_cfg5_generate_mode_with_brownian_on_flux_shape(api):
    """cfg5 as written (power-law + Perlin + brownian, scheduled / blended, DPM++ SDE with momentum) in generate mode on a
    Flux-shaped latent: runs entirely on device, finite, and the chain's noise stays unit-variance."""
    N, S, pn = api.noise, api.sonar, api.powernoise
    x0 = torch.randn(2, 16, 128, 128, device="cuda") * 10.0
    sigmas = torch.cat((torch.linspace(10.0, 0.5, 6), torch.zeros(1)))

    def chain_of(item):
        c = N.CustomNoiseChain()
        c.add(item)
        return c

    power = chain_of(pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0,
                                       pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1"))
    brown = chain_of(N.CustomNoiseItem(1.0, noise_type="brownian"))
    chain = N.CustomNoiseChain()
    chain.add(N.ScheduledNoise(0.6, noise=power, start_sigma=20.0, end_sigma=4.0, normalize=None, fallback_noise=brown))
    chain.add(N.BlendedNoise(0.4, custom_noise_1=chain_of(N.CustomNoiseItem(1.0, noise_type="perlin")), custom_noise_2=brown,
                             blend_function=api.utils.BLENDING_MODES["lerp"], noise_2_percent=0.5, normalize=None))
    ns = chain.make_noise_sampler(x0, 0.5, 10.0, seed=3, cpu=False, normalized=True)
    sample = ns(torch.tensor(8.0), torch.tensor(6.0))
    api.utils.pop_stats(sample)
    assert abs(sample.std().item() - 1.0) < 5e-3 and abs(sample.mean().item()) < 5e-3
    out = S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas, {"seed": 3}, None, True, None,
                                  dict(momentum=0.9, momentum_hist=0.7, direction=1.0), 0.9, 1.05, ns)
    assert out.shape == x0.shape and bool(torch.isfinite(out).all())


def test_power_noise_item_on_video_latents(api):
    """5-D [B, C, T, H, W] latents: the reference's irfft2 / channel mixer act on every [H, W] slice (py/nodes/powernoise.py:366-377)."""
    from oracle import sonar_oracle as orc

    shape = (2, 4, 3, 32, 48)
    x = torch.zeros(shape, device="cuda")
    for kw in ({}, {"common_mode": 0.25}):
        item = power_item(api, **kw)
        torch.manual_seed(21)
        out = item.make_noise_sampler(x, None, None, seed=None, cpu=True, normalized=True)(None, None)
        torch.manual_seed(21)
        z = orc.draw_power(shape)
        filt = item.make_filter(shape)
        noise = torch.fft.irfft2(z * filt, s=shape[-2:], norm="ortho")
        mixer = orc.channel_mixer(4, kw.get("common_mode", 0.0), torch.ones(6))
        if mixer is not None and not torch.equal(mixer, torch.eye(4)):
            noise = (mixer @ noise.swapaxes(0, 1).reshape(4, -1)).reshape(4, 2, *shape[2:]).swapaxes(1, 0)
        close(out, orc.scale_noise(noise, 1.0, normalized=True), rtol=0, atol=4e-5)
    # generate mode: two shards of the batch reproduce the whole (plane offsets count C * T planes per latent)
    ng = api.noise_generation
    item = power_item(api)

    def gen(xs, offset):
        torch.manual_seed(33)
        with ng.shard_offset(offset):
            return item.make_noise_sampler(xs, None, None, seed=None, cpu=False, normalized=False)(None, None)

    whole = gen(x, 0)
    assert torch.equal(gen(x[:1], 0), whole[:1]) and torch.equal(gen(x[1:], 1), whole[1:])
    assert abs(whole.std().item() - 0.9) < 0.1


def test_statistics_tags_are_dropped_by_any_write(api):
    """scale_noise tags its result with the result's statistics (so a wrapper that normalises again needs no sweep); the tag must not
    survive a write: through this library (directly or through a view) or through torch in-place ops."""
    U, hl = api.utils, api.hl
    x = torch.randn(4, 4, 32, 32, device="cuda") * 3 + 1
    y = U.scale_noise(x, 1.0, normalized=True)
    assert U._STATS_ATTR in y.__dict__
    again = U.scale_noise(y.clone(), 1.0, normalized=True)
    tagged = U.scale_noise(y, 1.0, normalized=True)  # uses the tag: decision "leave as is", values untouched
    assert torch.equal(tagged, again)
    for write in (lambda t: hl.mul_scalar(t, 2.0, out=t), lambda t: hl.mul_scalar(t.view(-1), 2.0, out=t.view(-1)), lambda t: t.mul_(2.0)):
        t = U.scale_noise(torch.randn(4, 4, 32, 32, device="cuda"), 1.0, normalized=True)
        write(t)
        out = U.scale_noise(t, 1.0, normalized=True)
        assert abs(out.std().item() - 1.0) < 1e-4  # a stale tag would have left the std at 2


def test_single_item_chain_uses_the_fused_normalised_fill(api):
    """A chain of ONE factor-1 gaussian / uniform item, normalised, on-device draws: draw + normalise with one write (NoiseSampler.
    normalized_call -> generate_normalized); same values as the unfused path (generator, then the chain's scale_noise)."""
    ng = api.noise_generation
    x = torch.zeros(8, 4, 64, 64, device="cuda")
    for name in ("gaussian", "uniform"):
        outs = []
        for fused in (True, False):
            chain = chain_of(api, item(api, name, 1.0))
            torch.manual_seed(77)
            ns = chain.make_noise_sampler(x, 0.03, 14.6, seed=1, cpu=False, normalized=True)
            if not fused:
                gen_cls = ng.GaussianNoiseGenerator if name == "gaussian" else ng.UniformNoiseGenerator
                saved = gen_cls.generate_normalized
                gen_cls.generate_normalized = lambda self, factor, *a: None
            try:
                outs.append(ns(*SIG))
            finally:
                if not fused:
                    gen_cls.generate_normalized = saved
        close(outs[0], outs[1], rtol=1e-5, atol=1e-6)
        assert abs(outs[0].std().item() - 1.0) < 7e-3  # inside scale_noise's 2.5 / sqrt(n) band nothing is rescaled (uniform: 3.46 / sqrt 12)


@pytest.mark.parametrize("name", ["gaussian", "perlin", "pyramid"])
def test_reseeding_with_the_same_value_rewinds_device_noise(api, name):
    """ComfyUI seeds before every run, usually with the same number: two runs in one process must draw the same device noise."""
    x = torch.zeros(4, 4, 32, 32, device="cuda")
    runs = []
    for _ in range(2):
        torch.manual_seed(1234)
        ns = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=True)
        runs.append(torch.stack([ns(*SIG) for _ in range(3)]))
    assert torch.equal(runs[0], runs[1])
    assert not torch.equal(runs[0][0], runs[0][1])  # consecutive calls within a run differ
    state = torch.cuda.get_rng_state()
    a = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=True)(*SIG)
    torch.cuda.set_rng_state(state)
    b = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, factor=1.0, normalized=True)(*SIG)
    assert torch.equal(a, b)
