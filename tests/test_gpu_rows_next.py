"""-m gpu: the §8a rows either side of the generators — spatial power-law noise (white / grey / velvet / violet,
AdvancedPowerLawNoise), wavelet-filtered noise (WF) and the sigma-gated latent operations (L) — through the
reference's plugin API with latents on the MI355X.

Goldens: tests/golden/powerlaw.npz, latent_ops.npz (captured from the real reference).  WF here: against
oracle/dwt_oracle.py (pinned to the reference's own WF outputs by tests/test_wavelet_oracle_cpu.py); the reference-run WF
fixtures themselves are compared in tests/test_gpu_wavelet_golden.py.
Tolerances: elementwise rows rtol 2e-5 / atol 5e-6 (powf differs from ATen's pow by <= 2 ulp); WF rtol/atol 3e-5."""
import importlib
import math
import types

import numpy as np
import pytest
import torch

from oracle import dwt_oracle as dwo
from tests.test_oracle_golden import LATENT_OP_CASES, ONEF_ADV, POWERLAW_ADV, POWERLAW_TYPES, SPECTRAL_TYPES

pytestmark = pytest.mark.gpu
SIG = (torch.tensor(14.6), torch.tensor(10.0))


@pytest.fixture(scope="module")
def api(pkg):
    pkg.hip_lib.load()
    mods = {m: importlib.import_module(f"comfyui_sonar_amd.py.{m}") for m in ("utils", "noise_generation", "noise", "latent_ops")}
    mods["registry"] = importlib.import_module("comfyui_sonar_amd.py.nodes.registry")
    mods["powernoise"] = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
    return types.SimpleNamespace(**mods, hl=pkg.hip_lib)


def close(a, b, rtol=2e-5, atol=5e-6):
    torch.testing.assert_close(a.detach().cpu().float(), b.detach().cpu().float(), rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------------ spatial power law
@pytest.mark.parametrize("name", list(POWERLAW_TYPES))
@pytest.mark.parametrize("normalized", [False, True])
def test_powerlaw_registry_types(api, golden, name, normalized):
    g = golden("powerlaw")
    x = torch.zeros(tuple(g["draw"].shape), device="cuda")
    torch.manual_seed(33)
    ns = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=33, cpu=True, factor=1.0, normalized=normalized)
    close(ns(*SIG), g[f"{name}_{int(normalized)}"])


@pytest.mark.parametrize("name", list(POWERLAW_ADV))
def test_powerlaw_advanced_item(api, golden, name):
    g = golden("powerlaw")
    x = torch.zeros(tuple(g["draw"].shape), device="cuda")
    torch.manual_seed(33)
    item = api.noise.AdvancedPowerLawNoise(1.0, **POWERLAW_ADV[name])
    close(item.make_noise_sampler(x, 0.03, 14.6, seed=33, cpu=True, normalized=False)(*SIG), g["adv_" + name])


def test_powerlaw_node_and_device_mode(api):
    node = api.registry.NODE_CLASS_MAPPINGS["SonarAdvancedPowerLawNoise"]()
    (chain,) = node.go(factor=1.0, rescale=0.0, alpha=0.5, div_max_dims="non-batch", use_sign=True, use_div_max_abs=True)
    x = torch.zeros(8, 4, 64, 64, device="cuda")
    out = chain.make_noise_sampler(x, 0.03, 14.6, seed=5, cpu=False, normalized=False)(*SIG)
    api.utils.pop_stats(out)
    peak = out.abs().flatten(1).amax(dim=1)
    torch.testing.assert_close(peak.cpu(), torch.ones(8), rtol=1e-6, atol=0)  # each latent divided by its own max |.|


# ------------------------------------------------------------------------------------------------ latent operations
OPS = (lambda latent: latent * 1.5 + 0.25, lambda latent: latent.abs() - 0.5)


@pytest.mark.parametrize("name", list(LATENT_OP_CASES))
def test_latent_operation_advanced(api, golden, name):
    g = golden("latent_ops")
    lo = api.latent_ops
    t = g["latent"].cuda()
    adv = lo.SonarLatentOperationAdvanced(ops=(lo.SonarLatentOperation(op=OPS[0]), lo.SonarLatentOperation(op=OPS[1])), start_sigma=10.0,
                                          end_sigma=1.0, op_alt=lo.SonarLatentOperation(op=OPS[1]), **LATENT_OP_CASES[name])
    close(adv(t.clone(), sigma=torch.tensor([5.0])), g[f"adv_{name}"])
    close(adv(t.clone(), sigma=torch.tensor([12.0])), g[f"adv_{name}_disabled"])


@pytest.mark.parametrize("scale_to_sigma", [False, True])
def test_latent_operation_noise(api, golden, scale_to_sigma):
    g = golden("latent_ops")
    lo = api.latent_ops
    chain = api.noise.CustomNoiseChain()
    chain.add(api.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    op = lo.SonarLatentOperationNoise(custom_noise=chain, scale_to_sigma=scale_to_sigma, cpu_noise=True, normalize=True)
    torch.manual_seed(45)
    close(op(g["latent"].cuda(), sigma=torch.tensor([3.0])), g[f"noise_{int(scale_to_sigma)}"])


def test_latent_operation_setseed_and_nodes(api, golden):
    g = golden("latent_ops")
    reg = api.registry.NODE_CLASS_MAPPINGS
    chain = api.noise.CustomNoiseChain()
    chain.add(api.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    (noise_op,) = reg["SonarLatentOperationNoise"].go(custom_noise=chain, scale_to_sigma=False, cpu_noise=True, normalize=True, lazy_noise_sampler=False)
    (seeded,) = reg["SonarLatentOperationSetSeed"].go(operation=noise_op, seed=77, restore_rng_state=True)
    torch.manual_seed(1)
    before = torch.random.get_rng_state()
    close(seeded(latent=g["latent"].cuda(), sigma=torch.tensor([2.0])), g["setseed"])
    assert torch.equal(before, torch.random.get_rng_state())
    (adv,) = reg["SonarLatentOperationAdvanced"].go(operation=OPS[0], operation_2=OPS[1], start_sigma=10.0, end_sigma=1.0, operation_alt=OPS[1],
                                                   **LATENT_OP_CASES["inject_scaled"])
    close(adv(g["latent"].cuda(), sigma=torch.tensor([5.0])), g["adv_inject_scaled"])
    assert adv(g["latent"].cuda(), sigma=None) is not None  # sigma None = always enabled


# ------------------------------------------------------------------------------------------------ wavelet-filtered noise
WF_CASES = {
    "defaults": dict(),
    "scaled": dict(yl_scale=0.5, yh_scales=[1.5, [1.0, 0.5, 2.0], "fill"]),
    "db4_sym": dict(wave="db4", mode="symmetric", level=2, yl_scale=0.0, yh_scales=1.25),
    "two_step": dict(wave="sym5", mode="reflect", level=2, yl_scale=1.5, yh_scales=[0.5, 2.0], two_step_inverse=True),
}


def _gauss_chain(api):
    chain = api.noise.CustomNoiseChain()
    chain.add(api.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    return chain


@pytest.mark.parametrize("name", list(WF_CASES))
def test_wavelet_filtered_generator(api, name):
    kw = WF_CASES[name]
    shape = (2, 4, 40, 24)
    x = torch.zeros(shape, device="cuda")
    torch.manual_seed(9)
    gen = api.noise_generation.WaveletFilteredNoiseGenerator(x, sigma_min=0.03, sigma_max=14.6, seed=9, cpu=True, normalized=False, **kw)
    out = gen(*SIG)
    torch.manual_seed(9)
    base = torch.randn(shape)
    want = dwo.wavelet_filtered_noise(base.numpy().astype(np.float64), **kw)
    close(out, torch.from_numpy(want), rtol=3e-5, atol=3e-5)


def test_wavelet_filtered_generator_1d_mode(api):
    """use_1d_dwt (py/noise_generation.py:1982-1986, 2027-2028): the transform runs over the flattened plane."""
    shape = (2, 4, 16, 24)
    x = torch.zeros(shape, device="cuda")
    torch.manual_seed(9)
    gen = api.noise_generation.WaveletFilteredNoiseGenerator(x, sigma_min=0.03, sigma_max=14.6, seed=9, cpu=True, normalized=False,
                                                             use_1d_dwt=True, wave="db2", level=2, yl_scale=0.5, yh_scales=[2.0, 1.5])
    out = gen(*SIG)
    torch.manual_seed(9)
    base = torch.randn(shape).numpy().astype(np.float64).reshape(2, 4, -1)
    want = dwo.wavelet_filtered_noise(base.reshape(shape), use_1d_dwt=True, wave="db2", level=2, yl_scale=0.5, yh_scales=[2.0, 1.5])
    close(out, torch.from_numpy(want), rtol=3e-5, atol=3e-5)


def test_wavelet_filtered_item_with_high_noise(api):
    """Low bands from one chain, high bands from another (yl_blend_high = 0, yh_blend_high = 1), then scale_noise."""
    shape = (2, 4, 32, 32)
    x = torch.zeros(shape, device="cuda")
    item = api.noise.WaveletFilteredNoise(1.0, noise=_gauss_chain(api), noise_high=_gauss_chain(api), normalize=None, normalize_noise=False,
                                          yaml_parameters="wave: db2\nlevel: 2\nmode: periodization\nyh_scales: [1.0, 0.5]\npreblend_yl_scale_high: 2.0\n")
    torch.manual_seed(12)
    out = item.make_noise_sampler(x, 0.03, 14.6, seed=12, cpu=True, normalized=True)(*SIG)
    torch.manual_seed(12)
    low, high = torch.randn(shape), torch.randn(shape)
    raw = dwo.wavelet_filtered_noise(low.numpy().astype(np.float64), noise_high=high.numpy().astype(np.float64), wave="db2", level=2,
                                     mode="periodization", yh_scales=[1.0, 0.5], preblend_high=(2.0, None))
    from oracle import sonar_oracle as orc

    want = orc.scale_noise(torch.from_numpy(raw).float(), 1.0, normalized=True)
    close(out, want, rtol=5e-5, atol=5e-5)


def test_wavelet_filtered_node(api):
    """The node wires custom_noise as BOTH sources when no high chain is given (py/nodes/noise_filters.py:952-954): the
    low band comes from the first draw, the high bands from the second."""
    node = api.registry.NODE_CLASS_MAPPINGS["SonarWaveletFilteredNoise"]()
    (chain,) = node.go(factor=1.0, rescale=0.0, normalize="disabled", normalize_noise=False, custom_noise=_gauss_chain(api),
                       yaml_parameters="wave: bior2.2\nlevel: 3\n")
    x = torch.zeros(4, 4, 64, 64, device="cuda")
    torch.manual_seed(3)
    out = chain.make_noise_sampler(x, 0.03, 14.6, seed=3, cpu=True, normalized=False)(*SIG)
    torch.manual_seed(3)
    low, high = torch.randn(4, 4, 64, 64), torch.randn(4, 4, 64, 64)
    want = dwo.wavelet_filtered_noise(low.numpy().astype(np.float64), noise_high=high.numpy().astype(np.float64), wave="bior2.2", level=3)
    close(out, torch.from_numpy(want), rtol=3e-5, atol=3e-5)


# ------------------------------------------------------------------------------------------------ spectral-gain rows (F1, F2, PowerFilterNoiseItem)
def near(a, b, rel=2e-5):
    """FFT rows: absolute tolerance relative to the reference output's peak (different FFT factorisation, fp32)."""
    b = b.detach().cpu().float()
    torch.testing.assert_close(a.detach().cpu().float(), b, rtol=0, atol=rel * float(b.abs().max()))


@pytest.mark.parametrize("name", list(SPECTRAL_TYPES))
@pytest.mark.parametrize("normalized", [False, True])
def test_spectral_registry_types(api, golden, name, normalized):
    g = golden("spectral")
    x = torch.zeros(tuple(g["draw1"].shape), device="cuda")
    torch.manual_seed(51)
    ns = api.noise.get_noise_sampler(name, x, 0.03, 14.6, seed=51, cpu=True, factor=1.0, normalized=normalized)
    near(ns(*SIG), g[f"{name}_{int(normalized)}"])


@pytest.mark.parametrize("name", list(ONEF_ADV))
def test_advanced_1f_item_and_node(api, golden, name):
    g = golden("spectral")
    kw = ONEF_ADV[name]
    x = torch.zeros(tuple(g["draw1"].shape), device="cuda")
    if "base_power" in kw:  # the node has no socket for it: the item, as the golden generator builds it
        chain = api.noise.CustomNoiseChain()
        chain.add(api.noise.Advanced1fNoise(1.0, **kw))
    else:
        node = api.registry.NODE_CLASS_MAPPINGS["SonarAdvanced1fNoise"]()
        (chain,) = node.go(factor=1.0, rescale=0.0, alpha=kw["alpha"], k=kw["k"], vertical_factor=kw["hfac"], horizontal_factor=kw["wfac"],
                           use_sqrt=kw["use_sqrt"])
    torch.manual_seed(51)
    near(chain.make_noise_sampler(x, 0.03, 14.6, seed=51, cpu=True, normalized=False)(*SIG), g["adv_" + name])


PF_CASES = {"pf_a": (dict(alpha=1.0, max_freq=0.5), dict(mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")),
            "pf_b": (dict(alpha=-0.5, min_freq=0.1, max_freq=0.7071, rotate=20.0, stretch=1.5),
                     dict(mix=0.7, common_mode=0.25, channel_correlation="1,0.5,0.2,1,0.3,0.1"))}


@pytest.mark.parametrize("tag", list(PF_CASES))
@pytest.mark.parametrize("normalized", [False, True])
def test_power_filter_noise_item(api, golden, tag, normalized):
    g = golden("spectral")
    fkw, ikw = PF_CASES[tag]
    x = torch.zeros(tuple(g["pf_draw"].shape), device="cuda")
    item = api.powernoise.PowerFilterNoiseItem(1.0, noise=_gauss_chain(api), normalize_noise=None, normalize_result=None, time_brownian=True,
                                               power_filter=api.powernoise.PowerFilter(**fkw), filter_norm_factor=1.0, **ikw)
    torch.manual_seed(52)
    near(item.make_noise_sampler(x, 0.03, 14.6, seed=52, cpu=True, normalized=normalized)(*SIG), g[f"{tag}_{int(normalized)}"], rel=4e-5)


def test_power_filter_noise_node_device_mode(api):
    """Node wiring + generate mode at SDXL size: the filtered chain keeps unit variance after normalisation."""
    reg = api.registry.NODE_CLASS_MAPPINGS
    (filt,) = reg["SonarPowerFilter"].go(min_freq=0.0, max_freq=0.7071, stretch=1.0, rotate=0.0, pnorm=2.0, alpha=1.0, blur=0.0, scale=1.0)
    node = reg["SonarPowerFilterNoise"]()
    (chain,) = node.go(factor=1.0, rescale=0.0, sonar_custom_noise=_gauss_chain(api), sonar_power_filter=filt, filter_norm_factor=1.0,
                       normalize_noise="default", normalize_result="default", mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
    x = torch.zeros(16, 4, 128, 128, device="cuda")
    out = chain.make_noise_sampler(x, 0.03, 14.6, seed=1, cpu=False, normalized=True)(*SIG)
    api.utils.pop_stats(out)
    assert abs(out.double().std().item() - 1.0) < 5e-3 and abs(out.double().mean().item()) < 5e-3
    # pink: low radial frequencies carry more energy than high ones
    spec = torch.fft.rfft2(out).abs().square().mean(dim=(0, 1))
    assert spec[1:4, 1:4].mean() > 4 * spec[40:60, 40:60].mean()


# ------------------------------------------------------------------------------------------------ Brownian-interval noise
def test_brownian_noise_is_one_consistent_path(api):
    """NoiseType.BROWNIAN (own counter-based Brownian-interval sampler; the reference's torchsde tree is un-vendored, parity
    unpinned): every call is N(0,1); increments add up exactly (nested / abutting intervals are one path); disjoint
    intervals are independent; same seed -> same values, another seed -> independent; batch shards agree."""
    NG = api.noise_generation
    x = torch.zeros(4, 4, 64, 64, device="cuda")
    ns = api.noise.get_noise_sampler("brownian", x, 0.03, 14.6, seed=1234, cpu=False, normalized=False)
    s = lambda v: torch.tensor(v)  # noqa: E731
    a = ns(s(10.0), s(6.0))
    n = a.numel()
    part = api.utils.pop_stats(a)  # the increment's statistics come out of the generating pass
    assert part is not None
    sums = part.view(-1, 2).sum(0)
    assert abs(sums[0].item() - a.double().sum().item()) < 1e-6 * n and abs(sums[1].item() - (a.double() ** 2).sum().item()) < 1e-6 * n
    tol = 5.0 / math.sqrt(n)
    assert abs(a.mean().item()) < tol and abs(a.var().item() - 1.0) < 3 * tol
    assert abs(((a.double() ** 4).mean() / a.double().var() ** 2).item() - 3.0) < 30 * tol
    b = ns(s(6.0), s(2.5))
    whole = ns(s(10.0), s(2.5))
    recomposed = (a * math.sqrt(4.0) + b * math.sqrt(3.5)) / math.sqrt(7.5)
    torch.testing.assert_close(whole, recomposed, rtol=0, atol=2e-5)
    assert abs((a * b).mean().item()) < tol  # disjoint intervals
    torch.testing.assert_close(ns(s(6.0), s(10.0)), -a, rtol=0, atol=0)  # direction flips the sign (k-diffusion convention)
    assert torch.equal(ns(s(10.0), s(6.0)), a)
    again = api.noise.get_noise_sampler("brownian", x, 0.03, 14.6, seed=1234, cpu=False, normalized=False)(s(10.0), s(6.0))
    other = api.noise.get_noise_sampler("brownian", x, 0.03, 14.6, seed=1235, cpu=False, normalized=False)(s(10.0), s(6.0))
    assert torch.equal(again, a) and abs((other * a).mean().item()) < tol
    with NG.shard_offset(2):
        part = api.noise.get_noise_sampler("brownian", x[:2], 0.03, 14.6, seed=1234, cpu=False, normalized=False)(s(10.0), s(6.0))
    assert torch.equal(part, a[2:])
    with pytest.raises(ValueError):
        api.noise.get_noise_sampler("brownian", x, None, None, seed=1)


@pytest.mark.parametrize("seeds", [None, [5, 6, 7]])
def test_brownian_bridge_route_equals_expansion_route(api, seeds):
    """A new time is evaluated from the kept W tensors of the two times it was bridged between (one fresh normal per element); without kept
    tensors (CACHE_POINTS = 0) from its whole expansion over the node normals, in chunks of 96 terms.  Same linear combination, fp32
    rounding apart.  The walk is a DPM++ SDE run's: (t, s) and (t, t') per step, 60 steps -> expansions of up to ~120 terms; the last steps run below sigma_min,
    where the path continues as independent increments from the outermost known time."""
    NG = api.noise_generation
    x = torch.zeros(3, 4, 64, 64, device="cuda")
    seed = 77 if seeds is None else seeds
    kept = NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=seed, tree_depth=0)  # (the path of bridges: the tree mode always expands)
    bare = NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=seed, tree_depth=0)
    bare.CACHE_POINTS = 0
    sig = [14.6 * 0.9**k for k in range(61)]
    worst = 0.0
    for k in range(60):
        mid = math.sqrt(sig[k] * sig[k + 1])
        for pair in ((sig[k], mid), (sig[k], sig[k + 1])):
            a = kept(torch.tensor(pair[0]), torch.tensor(pair[1]))
            b = bare(torch.tensor(pair[0]), torch.tensor(pair[1]))
            worst = max(worst, float((a - b).abs().max()))
            if k in (0, 30, 59):
                n = a.numel()
                assert abs(a.mean().item()) < 5 / math.sqrt(n) and abs(a.var().item() - 1.0) < 15 / math.sqrt(n)
    assert len(kept._points) <= kept.CACHE_POINTS and len(bare.path.coefficients(float(torch.tensor(sig[55])))) > 100
    assert worst < 2e-5, worst
    # the whole run is one path: the first and the last step's increments, recomposed from the end points, agree with a direct query
    whole = kept(torch.tensor(sig[0]), torch.tensor(sig[60]))
    direct = bare(torch.tensor(sig[0]), torch.tensor(sig[60]))
    torch.testing.assert_close(whole, direct, rtol=0, atol=2e-5)


def test_brownian_tree_mode_does_not_depend_on_the_query_history(api, monkeypatch):
    """The virtual Brownian tree (BrownianPath tree mode, the default: what ComfyUI's BrownianTree is up to its tolerance): an increment
    is the same BITS after any history -- a run with other step counts, a sampler made half way through a run, the other order --
    where the path of bridges (depth 0) is only self-consistent per instance; still N(0, 1), additive over abutting intervals, and the
    module switch reaches the registry's generator."""
    NG = api.noise_generation
    x = torch.zeros(2, 4, 64, 64, device="cuda")
    s = lambda v: torch.tensor(v)  # noqa: E731
    mk = lambda depth=24: NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=99, tree_depth=depth)  # noqa: E731
    fine, coarse, late = mk(), mk(), mk()
    sig = [14.6 * 0.8**k for k in range(16)]
    steps_fine = [fine(s(sig[k]), s(sig[k + 1])) for k in range(15)]         # 15 steps down
    steps_coarse = [coarse(s(sig[k]), s(sig[k + 3])) for k in (0, 3, 6)]     # 3 steps over the same times
    assert torch.equal(late(s(sig[7]), s(sig[8])), steps_fine[7])            # no history at all
    assert torch.equal(coarse(s(sig[7]), s(sig[8])), steps_fine[7])          # another history
    assert torch.equal(fine(s(sig[3]), s(sig[6])), steps_coarse[1])
    assert torch.equal(fine(s(sig[8]), s(sig[7])), -steps_fine[7])
    dt = lambda i, j: fine.path.resolve(sig[i]) - fine.path.resolve(sig[j])  # noqa: E731
    recomposed = sum(steps_fine[k] * math.sqrt(dt(k, k + 1)) for k in (3, 4, 5)) / math.sqrt(dt(3, 6))
    torch.testing.assert_close(steps_coarse[1], recomposed, rtol=0, atol=2e-5)
    n = steps_fine[0].numel()
    for a in (steps_fine[0], steps_fine[9], steps_coarse[2]):
        assert abs(a.mean().item()) < 5 / math.sqrt(n) and abs(a.var().item() - 1.0) < 15 / math.sqrt(n)
    assert abs((steps_fine[2] * steps_fine[3]).mean().item()) < 5 / math.sqrt(n)
    # the path of bridges (depth 0) is a function of the history: the same query after another history gives other values
    d1, d2 = mk(0), mk(0)
    d1(s(10.0), s(6.0))
    assert not torch.equal(d1(s(8.0), s(7.0)), d2(s(8.0), s(7.0)))
    # the registry's generator follows the module switch: a tree of depth 24 unless SONAR_BROWNIAN_TREE says otherwise
    monkeypatch.setattr(NG, "BROWNIAN_TREE_DEPTH", 24)
    ns = api.noise.get_noise_sampler("brownian", x, 0.03, 14.6, seed=99, cpu=False, normalized=False)
    assert torch.equal(ns(s(sig[7]), s(sig[8])), steps_fine[7])
    assert NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=99).path.tree_depth == 24
    monkeypatch.setattr(NG, "BROWNIAN_TREE_DEPTH", 0)
    assert NG.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=99).path.tree_depth == 0
    # a chain folds a tree sampler's increments like any other item's
    acc = torch.ones_like(x)
    assert late.accumulate(acc, 0.5, 2.0, None, s(sig[7]), s(sig[8]))
    torch.testing.assert_close(acc, 0.5 + 2.0 * steps_fine[7], rtol=0, atol=1e-6)


def test_brownian_batched_seeds_and_time_brownian_power_noise(api):
    x = torch.zeros(3, 4, 32, 32, device="cuda")
    tree = api.noise_generation.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=[5, 6, 5])
    out = tree(torch.tensor(9.0), torch.tensor(4.0))
    assert torch.equal(out[0], out[2]) and not torch.equal(out[0], out[1])  # one path per seed, counters restart per latent
    item = api.powernoise.PowerNoiseItem(1.0, time_brownian=True, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                                         mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
    ns = item.make_noise_sampler(x, 0.03, 14.6, seed=11, cpu=False, normalized=True)
    a, b = ns(torch.tensor(9.0), torch.tensor(4.0)), ns(torch.tensor(9.0), torch.tensor(4.0))
    for t in (a, b):
        api.utils.pop_stats(t)
    assert torch.equal(a, b) and abs(a.std().item() - 1.0) < 5e-3  # time-correlated: the same interval gives the same field
    with pytest.raises(ValueError):
        item.make_noise_sampler(x, None, None, seed=11, cpu=False, normalized=True)


# ------------------------------------------------------------------------------------------------ wavelet (octave) noise
WAVELET_BASE = {"octave_scale_mode": "adaptive_avg_pool2d", "octave_rescale_mode": "bilinear", "post_octave_rescale_mode": "bilinear",
                "initial_amplitude": 1.0, "persistence": 0.5, "octaves": 4, "octave_height_factor": 0.5, "octave_width_factor": 0.5,
                "height_factor": 2.0, "width_factor": 2.0, "update_blend": 1.0}
WAVELET_ADV = {"deep": dict(octaves=6, persistence=0.7, initial_amplitude=2.0, height_factor=1.5, width_factor=1.5, min_height=2, min_width=2),
               "reverse": dict(octaves=-3, persistence=0.5, octave_height_factor=0.25, octave_width_factor=0.5, update_blend=0.6),
               "modes": dict(octaves=3, octave_scale_mode="area", octave_rescale_mode="nearest-exact", post_octave_rescale_mode="bilinear")}


@pytest.mark.parametrize("normalized", [False, True])
def test_wavelet_noise_preset(api, golden, normalized):
    g = golden("wavelet_noise")
    want = g[f"preset_{int(normalized)}"]
    x = torch.zeros(tuple(want.shape), device="cuda")
    torch.manual_seed(91)
    ns = api.noise.get_noise_sampler("wavelet", x, 0.03, 14.6, seed=91, cpu=True, factor=1.0, normalized=normalized)
    close(ns(*SIG), want)


@pytest.mark.parametrize("name", list(WAVELET_ADV))
def test_wavelet_noise_advanced(api, golden, name):
    g = golden("wavelet_noise")
    want = g["adv_" + name]
    x = torch.zeros(tuple(want.shape), device="cuda")
    item = api.noise.AdvancedWaveletNoise(1.0, custom_noise=None, normalize_noise=False, normalize=None,
                                          update_blend_function=api.utils.BLENDING_MODES["lerp"], **(WAVELET_BASE | WAVELET_ADV[name]))
    torch.manual_seed(91)
    close(item.make_noise_sampler(x, 0.03, 14.6, seed=91, cpu=True, normalized=False)(*SIG), want)


def test_wavelet_noise_node_with_custom_source(api, golden):
    g = golden("wavelet_noise")
    want = g["adv_custom"]
    x = torch.zeros(tuple(want.shape), device="cuda")
    chain = api.noise.CustomNoiseChain()
    chain.add(api.noise.CustomNoiseItem(1.0, noise_type="uniform"))
    node = api.registry.NODE_CLASS_MAPPINGS["SonarWaveletNoise"]()
    (out_chain,) = node.go(factor=1.0, rescale=0.0, normalize="default", normalize_noise=True, custom_noise=chain, update_blend_mode="lerp",
                           **(WAVELET_BASE | {"octaves": 3}))
    torch.manual_seed(92)
    close(out_chain.make_noise_sampler(x, 0.03, 14.6, seed=92, cpu=True, normalized=True)(*SIG), want)
    with pytest.raises(ValueError):
        node.go(factor=1.0, rescale=0.0, normalize="default", normalize_noise=True, update_blend_mode="lerp", **(WAVELET_BASE | {"octaves": 0}))


# ------------------------------------------------------------------------------------------------ guided noise (8f rank 1)
@pytest.mark.parametrize("method", ["linear", "euler"])
@pytest.mark.parametrize("with_noise", [True, False])
def test_guided_noise(api, golden, method, with_noise):
    g = golden("guided_noise")
    x = g["x"].cuda()
    chain = _gauss_chain(api) if with_noise else None
    item = api.noise.GuidedNoise(1.0, guidance_factor=0.4, ref_latent=g["ref_latent"].cuda(), method=method, normalize_noise=None,
                                 normalize_result=None, noise=chain)
    torch.manual_seed(96)
    ns = item.make_noise_sampler(x.clone(), 0.03, 14.6, seed=96, cpu=True, normalized=True)
    close(ns(torch.tensor(9.0), torch.tensor(6.0)), g[f"{method}_{int(with_noise)}"], rtol=2e-5, atol=2e-5)


def test_guided_noise_node_prepares_the_reference(api, golden):
    g = golden("guided_noise")
    node = api.registry.NODE_CLASS_MAPPINGS["SonarGuidedNoise"]()
    (chain,) = node.go(factor=1.0, latent={"samples": g["latent"].cuda()}, normalize_noise="default", normalize_result="default",
                       normalize_ref=True, method="euler", guidance_factor=0.4, sonar_custom_noise=_gauss_chain(api))
    close(chain.items[0].ref_latent, g["ref_latent"], rtol=2e-5, atol=2e-5)
    torch.manual_seed(96)
    out = chain.make_noise_sampler(g["x"].cuda(), 0.03, 14.6, seed=96, cpu=True, normalized=True)(torch.tensor(9.0), torch.tensor(6.0))
    close(out, g["euler_1"], rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------------------------ ModulatedNoise
@pytest.mark.parametrize("with_ref", [True, False])
@pytest.mark.parametrize("dims", [1, 2, 3])
@pytest.mark.parametrize("mtype", ["intensity", "frequency", "none"])
def test_modulated_noise_matches_reference(api, golden, mtype, dims, with_ref):
    """py/noise.py:762-1019 against vectors captured from the reference itself (tests/golden/make_golden.py gen_modulated)."""
    g = golden("modulated")
    x = g["x"].cuda()
    item = api.noise.ModulatedNoise(0.8, noise=_gauss_chain(api), normalize_result=None if with_ref else False, normalize_noise=None, normalize_ref=True,
                                    modulation_type=mtype, modulation_strength=1.5 if with_ref else -0.6, modulation_dims=dims,
                                    ref_latent_opt=g["latent"] if with_ref else None)
    torch.manual_seed(98)
    ns = item.clone().make_noise_sampler(x, 0.03, 14.6, seed=98, cpu=True, normalized=True)
    out = ns(torch.tensor(9.0), torch.tensor(6.0))
    assert out.shape == x.shape and out.is_contiguous()
    close(out, g[f"{mtype}_{dims}_{int(with_ref)}"], rtol=2e-5, atol=2e-5)
    if not with_ref and mtype != "none":
        close(x, g["x_after"], rtol=1e-5, atol=1e-5)  # the reference normalises the sampler's x in place


def test_modulated_noise_full_size_properties_and_node(api):
    """SDXL batch: intensity mode with strength 1 keeps the plain noise's L2 norm; frequency mode runs on 128 x 128 planes; the node maps
    normalize_ref like the reference (a boolean becomes False)."""
    shape = (16, 4, 128, 128)
    x = torch.randn(shape, device="cuda")
    ref = torch.randn(shape, device="cuda") * torch.linspace(0.2, 3.0, 128, device="cuda")[None, None, :, None]
    for mtype in ("intensity", "frequency"):
        item = api.noise.ModulatedNoise(1.0, noise=_gauss_chain(api), normalize_result=False, normalize_noise=True, normalize_ref=False,
                                        modulation_type=mtype, modulation_strength=1.0, modulation_dims=2, ref_latent_opt=ref)
        out = item.make_noise_sampler(x, 0.03, 14.6, seed=1, cpu=False, normalized=True)(torch.tensor(9.0), torch.tensor(6.0))
        sigma_up = min(6.0, (36.0 * (81.0 - 36.0) / 81.0) ** 0.5)
        want_norm = sigma_up * (x.numel() ** 0.5)  # ||unit-variance noise * sigma_up||
        assert abs(out.norm().item() / want_norm - 1.0) < 5e-3
        assert torch.isfinite(out).all()
    node = api.registry.NODE_CLASS_MAPPINGS["SonarModulatedNoise"]()
    (chain,) = node.go(factor=1.0, sonar_custom_noise=_gauss_chain(api), modulation_type="intensity", dims=3, strength=2.0, normalize_result="default",
                       normalize_noise="default", normalize_ref=True, ref_latent_opt={"samples": ref.cpu()})
    assert chain.items[0].normalize_ref is False and chain.items[0].modulation_dims == 3


@pytest.mark.parametrize("strength", [2.0, -0.7])
@pytest.mark.parametrize("dims", [1, 2, 3])
@pytest.mark.parametrize("tag,shape", [("b1", (1, 4, 32, 32)), ("b4", (4, 4, 16, 16)), ("g1", (1, 4, 26, 38)), ("o1", (1, 3, 9, 15))])
def test_modulated_noise_spectral_signum(api, golden, tag, shape, dims, strength):
    """py/noise.py:938-1015 against the reference's own outputs.  FFT rows: 2e-5 of the output peak; a bin whose log amplitude sits
    within rounding of a quantile threshold may land on the other side of it, but the clamp is continuous there."""
    g = golden("spectral_signum")
    item = api.noise.ModulatedNoise(0.9, noise=_gauss_chain(api), normalize_result=None, normalize_noise=None, normalize_ref=False,
                                    modulation_type="spectral_signum", modulation_strength=strength, modulation_dims=dims)
    torch.manual_seed(99)
    ns = item.make_noise_sampler(torch.zeros(shape, device="cuda"), 0.03, 14.6, seed=99, cpu=True, normalized=True)
    near(ns(torch.tensor(9.0), torch.tensor(6.0)), g[f"{tag}_{dims}_{strength}"], rel=3e-5)


def test_modulated_noise_spectral_signum_batch_quirk(api, golden):
    """The reference's quantile broadcast only lines up for B = 1 or B = C; any other batch fails there, and here."""
    item = api.noise.ModulatedNoise(1.0, noise=_gauss_chain(api), normalize_result=None, normalize_noise=None, normalize_ref=False,
                                    modulation_type="spectral_signum")
    ns = item.make_noise_sampler(torch.zeros(2, 4, 16, 16, device="cuda"), 0.03, 14.6, seed=99, cpu=True, normalized=True)
    with pytest.raises(RuntimeError) as exc:
        ns(torch.tensor(9.0), torch.tensor(6.0))
    assert str(exc.value).startswith(str(golden("spectral_signum")["b2_error"])[:40])


# ------------------------------------------------------------------------------------------------ item wrappers
def _chain(api, *specs):
    c = api.noise.CustomNoiseChain()
    for name, f in specs:
        c.add(api.noise.CustomNoiseItem(f, noise_type=name))
    return c


def _sequence(ns, n):
    return torch.stack([ns(torch.tensor(9.0), torch.tensor(6.0)).cpu() for _ in range(n)])


@pytest.mark.parametrize("mix", [1, 2])
def test_random_noise(api, golden, mix):
    g = golden("item_wrappers")
    item = api.noise.RandomNoise(0.7, noise=_chain(api, ("gaussian", 1.0), ("uniform", 0.5), ("perlin", 0.8)), mix_count=mix, normalize=None)
    torch.manual_seed(41)
    ns = item.clone().make_noise_sampler(torch.zeros(2, 4, 8, 8, device="cuda"), 0.03, 14.6, seed=41, cpu=True, normalized=True)
    close(_sequence(ns, 4), g[f"random_mix{mix}"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("permute", ["enabled", "always", "disabled"])
def test_repeated_noise(api, golden, permute):
    g = golden("item_wrappers")
    item = api.noise.RepeatedNoise(0.9, noise=_chain(api, ("gaussian", 1.0)), repeat_length=2, max_recycle=3, normalize=None, permute=permute)
    torch.manual_seed(42)
    ns = item.clone().make_noise_sampler(torch.zeros(2, 4, 8, 8, device="cuda"), 0.03, 14.6, seed=42, cpu=True, normalized=True)
    close(_sequence(ns, 9), g[f"repeated_{permute}"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("mode", ["wrap", "repeat", "zero"])
def test_channel_noise_and_nodes(api, golden, mode):
    g = golden("item_wrappers")
    item = api.noise.ChannelNoise(1.1, noise=_chain(api, ("gaussian", 1.0), ("uniform", 0.5)), insufficient_channels_mode=mode, normalize=None)
    torch.manual_seed(43)
    ns = item.clone().make_noise_sampler(torch.zeros(2, 4, 8, 8, device="cuda"), 0.03, 14.6, seed=43, cpu=True, normalized=True)
    close(_sequence(ns, 2), g[f"channel_{mode}"], rtol=2e-5, atol=2e-5)
    M = api.registry.NODE_CLASS_MAPPINGS
    (c1,) = M["SonarChannelNoise"]().go(1.0, sonar_custom_noise=_chain(api, ("gaussian", 1.0)), insufficient_channels_mode=mode, normalize="default")
    (c2,) = M["SonarRandomNoise"]().go(1.0, _chain(api, ("gaussian", 1.0)), 1, "forced")
    (c3,) = M["SonarRepeatedNoise"]().go(factor=1.0, sonar_custom_noise=_chain(api, ("gaussian", 1.0)), repeat_length=4, max_recycle=10,
                                          normalize="disabled", permute="enabled")
    x = torch.zeros(1, 4, 16, 16, device="cuda")
    for c in (c1, c2, c3):
        out = c.make_noise_sampler(x, 0.03, 14.6, seed=1, cpu=False, normalized=True)(torch.tensor(9.0), torch.tensor(6.0))
        assert out.shape == x.shape and bool(torch.isfinite(out).all())
    assert c2.items[0].normalize is True and c3.items[0].normalize is False


def test_latent_operation_filtered_noise(api):
    """py/noise.py:1665-1698: chain noise -> LATENT_OPERATIONs (sigma-gated SonarLatentOperation wrappers or plain callables) -> scale_noise."""
    L = api.latent_ops
    x = torch.zeros(2, 4, 16, 16, device="cuda")
    double = lambda latent: api.hl.mul_scalar(latent, 2.0)  # a plain ComfyUI-style latent operation
    gated = L.SonarLatentOperation(start_sigma=5.0, end_sigma=1.0, op=lambda latent: api.hl.mul_scalar(latent, 10.0))  # outside its window at sigma 9
    node = api.registry.NODE_CLASS_MAPPINGS["SonarLatentOperationFilteredNoise"]()
    (chain,) = node.go(factor=0.5, rescale=0.0, normalize="disabled", normalize_noise=False, custom_noise=_chain(api, ("gaussian", 1.0)),
                       operation_1=double, operation_2=gated)
    torch.manual_seed(3)
    out = chain.make_noise_sampler(x, 0.03, 14.6, seed=3, cpu=True, normalized=False)(torch.tensor(9.0), torch.tensor(6.0))
    torch.manual_seed(3)
    want = torch.randn(2, 4, 16, 16) * 2.0 * 0.5 * 0.5  # item factor, then the unnormalised chain multiplies by the sum of |factors| again
    close(out, want, rtol=1e-6, atol=1e-6)
    torch.manual_seed(3)
    out = chain.make_noise_sampler(x, 0.03, 14.6, seed=3, cpu=True, normalized=False)(torch.tensor(3.0), torch.tensor(2.0))
    close(out, want * 10.0, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("tag,kw", [("cos_dim-1", dict(mode="cos", dim=-1, flatten=False)), ("sin_copysign_dim1", dict(mode="sin_copysign", dim=1, flatten=False)),
                                    ("cos_flat2", dict(mode="cos", dim=2, flatten=True))])
def test_ripple_filtered_noise(api, golden, tag, kw):
    g = golden("item_wrappers")
    item = api.noise.RippleFilteredNoise(0.8, noise=_chain(api, ("gaussian", 1.0)), offset=0.3, roll=1.5, amplitude_high=0.25, amplitude_low=1.6,
                                         period=3.0, normalize_noise=False, normalize=None, **kw)
    torch.manual_seed(44)
    ns = item.clone().make_noise_sampler(torch.zeros(2, 4, 8, 8, device="cuda"), 0.03, 14.6, seed=44, cpu=True, normalized=True)
    close(_sequence(ns, 3), g[f"ripple_{tag}"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tag,kw", [("dim1_chunk1", dict(dim=1, shrink_dim=False, chunk_size=1)), ("dim2_chunk4", dict(dim=2, shrink_dim=False, chunk_size=4)),
                                    ("dim1_shrink", dict(dim=1, shrink_dim=True, chunk_size=1))])
def test_per_dim_noise(api, golden, tag, kw):
    g = golden("item_wrappers")
    item = api.noise.PerDimNoise(0.6, noise=_chain(api, ("gaussian", 1.0)), offset=0, normalize_noise=False, normalize=None, **kw)
    torch.manual_seed(45)
    ns = item.clone().make_noise_sampler(torch.zeros(2, 4, 8, 8, device="cuda"), 0.03, 14.6, seed=45, cpu=True, normalized=True)
    close(_sequence(ns, 2), g[f"perdim_{tag}"], rtol=2e-5, atol=2e-5)


def _power_item(api, **kw):
    args = dict(time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0,
                channel_correlation="1,1,1,1,1,1")
    args.update(kw)
    return api.powernoise.PowerNoiseItem(1.0, **args)


def test_wrappers_over_power_items(api, golden):
    """RandomNoise / ChannelNoise call their items with normalized=False and normalise the result themselves: a PowerNoiseItem with
    factor 1 and the identity mixer returns early from its own scale_noise and must not hand on a statistics tag it never filled."""
    g = golden("power_wrapped")
    x = torch.zeros(2, 4, 16, 16, device="cuda")
    chain = api.noise.CustomNoiseChain()
    chain.add(_power_item(api))
    chain.add(_power_item(api, alpha=2.0))
    item = api.noise.RandomNoise(1.0, noise=chain, mix_count=1, normalize=None)
    torch.manual_seed(46)
    ns = item.make_noise_sampler(x, 0.03, 14.6, seed=46, cpu=True, normalized=True)
    near(_sequence(ns, 4), g["random_power"])
    chain = api.noise.CustomNoiseChain()
    chain.add(_power_item(api))
    chain.add(api.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    item = api.noise.ChannelNoise(1.0, noise=chain, insufficient_channels_mode="wrap", normalize=None)
    torch.manual_seed(47)
    ns = item.make_noise_sampler(x, 0.03, 14.6, seed=47, cpu=True, normalized=True)
    near(_sequence(ns, 2), g["channel_power"])
