"""CPU: the oracle (oracle/sonar_oracle.py) against the golden vectors captured from the real reference
(tests/golden/make_golden.py).  Same torch op sequence -> bit-exact on the machine that produced the
fixtures; on other host CPUs vectorised transcendental/FMA paths may differ in the last bit, so the
comparison is assert_close at 1-ulp-class tolerances (rtol 2e-6) with exact equality where no such op is involved."""
import math

import numpy as np
import pytest
import torch

from oracle import sonar_oracle as orc

TIGHT = dict(rtol=2e-6, atol=1e-7)


def close(a, b, **kw):
    torch.testing.assert_close(a, b, **(TIGHT | kw))


@pytest.mark.parametrize("case", ["plain", "shifted", "scaled", "both", "tiny_shift"])
def test_scale_noise(golden, case):
    g = golden("scale_noise")
    factor, sub, div = (float(v) for v in g[f"{case}_meta"])
    dec = {}
    out = orc.scale_noise(g[f"{case}_in"].clone(), factor, normalized=True, decisions=dec)
    close(out, g[f"{case}_out"])
    assert (dec["sub"], dec["div"]) == (bool(sub), bool(div))


def test_scale_noise_dims_and_unnormalized(golden):
    g = golden("scale_noise")
    close(orc.scale_noise(g["dims_in"].clone(), 0.7, normalized=True, normalize_dims=(-2, -1)), g["dims_out"])
    assert torch.equal(orc.scale_noise(g["dims_in"].clone(), 1.9, normalized=False), g["unnorm_out"])
    assert orc.scale_noise(torch.zeros(0), 2.0).numel() == 0
    assert torch.isnan(orc.scale_noise(torch.zeros(4, 4), 1.0)).all()  # zeros -> NaN


def test_basic_types(golden):
    g = golden("basic_types")
    assert torch.equal(g["gaussian_0"], g["gaussian_draw"])
    assert torch.equal(orc.uniform_noise(g["uniform_draw"]), g["uniform_0"])
    close(orc.scale_noise(g["gaussian_draw"].clone(), 1.0, normalized=True), g["gaussian_1"])
    torch.manual_seed(21)
    assert torch.equal(orc.draw_gaussian((2, 4, 8, 8)), g["gaussian_draw"])  # RNG order / dtype


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_perlin(golden, tag):
    g = golden("perlin")
    draws = orc.PerlinDraws(g[f"{tag}_base"], tuple(g[f"{tag}_angles"]))
    raw = orc.perlin_noise(draws, 2.0, str(g[f"{tag}_blend"]))
    close(raw, g[f"{tag}_raw"])
    close(orc.scale_noise(raw.clone(), 1.0, normalized=True), g[f"{tag}_out"], rtol=1e-5, atol=1e-6)
    torch.manual_seed(int(g[f"{tag}_seed"]))
    again = orc.draw_perlin(tuple(draws.base.shape), 2)
    assert torch.equal(again.base, draws.base) and all(torch.equal(a, b) for a, b in zip(again.angles, draws.angles))
    # lattice term is shared by every latent of the batch
    if draws.base.shape[0] > 1:
        d = raw[0] - raw[1]
        assert torch.allclose(d, (draws.base[0] - draws.base[1]) / 2, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_pyramid(golden, tag):
    g = golden("pyramid")
    n = int(g[f"{tag}_nlevels"])
    draws = orc.PyramidDraws(g[f"{tag}_base"], tuple(g[f"{tag}_rs"].tolist()), tuple(g[f"{tag}_level{i}"] for i in range(n)))
    raw = orc.pyramid_noise(draws, float(g[f"{tag}_discount"]), str(g[f"{tag}_mode"]))
    close(raw, g[f"{tag}_raw"])
    h, w = draws.base.shape[-2:]
    assert orc.pyramid_sizes(h, w, draws.rs) == [tuple(l.shape[-2:]) for l in draws.levels]
    assert tuple(draws.levels[0].shape[-2:]) == (h, w)  # level 0 is a second full-resolution draw
    torch.manual_seed(int(g[f"{tag}_seed"]))
    again = orc.draw_pyramid(tuple(draws.base.shape), 10)
    assert torch.equal(again.base, draws.base) and again.rs == draws.rs


def test_power_filter_and_mixer(golden):
    g = golden("power_filter")
    cases = {"white": dict(alpha=0.0), "pink": dict(alpha=1.0), "half": dict(alpha=0.5), "brown": dict(alpha=2.0),
             "blue": dict(alpha=-0.5), "band": dict(alpha=1.0, min_freq=0.1, max_freq=0.4),
             "rot_stretch": dict(alpha=1.0, rotate=30.0, stretch=2.0), "squash": dict(alpha=0.5, stretch=0.5), "pnorm1": dict(alpha=1.0, pnorm=1.0)}
    for name, kw in cases.items():
        for hw in ((32, 32), (16, 24)):
            shape = (1, 4, *hw)
            raw = orc.power_filter_build(shape, **kw)
            close(raw, g[f"{name}_{hw[0]}x{hw[1]}_raw"], atol=1e-30)
            for mix, nf in ((1.0, 1.0), (0.6, 1.0), (1.0, 0.5)):
                close(orc.power_filter_normalize(raw.clone(), shape, mix, nf), g[f"{name}_{hw[0]}x{hw[1]}_mix{mix}_nf{nf}"], atol=1e-30)
    f = orc.power_filter_normalize(orc.power_filter_build((1, 4, 128, 128), alpha=1.0, max_freq=0.7071), (1, 4, 128, 128))
    close(f, g["cfg2_128x128"], atol=1e-30)
    assert f[0, 0, 0, 0] == 0 and abs(f.square().mean().sqrt().item() - 1.0) < 1e-6
    corr = torch.ones(6)
    assert torch.equal(orc.channel_mixer(4, 0.0, corr), torch.eye(4))
    for cm in (0.25, -0.2):
        close(orc.channel_mixer(4, cm, corr), g[f"mixer_{cm}"])
    close(orc.channel_mixer(4, 0.4, torch.tensor([0.5, -0.3, 0.8])), g["mixer_partial"])


@pytest.mark.parametrize("tag", ["cfg2", "b", "c", "d", "e", "np2", "np2_rot", "odd", "sdxl_portrait"])
def test_power_noise(golden, tag):
    g = golden("power_noise")
    z = torch.view_as_complex(g[f"{tag}_z"].contiguous())
    shape = tuple(g[f"{tag}_out"].shape)
    mixer = g[f"{tag}_mixer"]
    mixer = None if torch.equal(mixer, torch.eye(mixer.shape[0])) else mixer
    dec, pre = {}, []
    out = orc.power_noise(z, g[f"{tag}_filter"], shape, mixer, 1.0, bool(g[f"{tag}_normalized"]), decisions=dec, pre_norm=pre)
    close(pre[0], g[f"{tag}_pre"], rtol=1e-5, atol=1e-6)
    close(out, g[f"{tag}_out"], rtol=1e-5, atol=2e-6)
    if bool(g[f"{tag}_normalized"]):
        assert [dec["sub"], dec["div"]] == [bool(v) for v in g[f"{tag}_branches"]]
    torch.manual_seed(int(g[f"{tag}_seed"]))
    assert torch.equal(torch.view_as_real(orc.draw_power(shape)), g[f"{tag}_z"])


def test_complex_randn_identity():
    """complex64 randn == view_as_complex(randn(..., 2)) * sqrt(1/2) — defines the replay input of PW."""
    torch.manual_seed(5)
    a = torch.randn(3, 7, dtype=torch.complex64)
    torch.manual_seed(5)
    b = torch.view_as_complex(torch.randn(3, 7, 2)) * math.sqrt(0.5)
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))


def test_composition(golden):
    g = golden("composition")
    p = orc.PerlinDraws(g["chain_perlin_base"], tuple(g["chain_perlin_angles"]))
    for tag in ("chain", "chain_rescaled"):
        out = orc.chain_noise([g["chain_gauss"], orc.uniform_noise(g["chain_uniform_u"]), orc.perlin_noise(p)], g[f"{tag}_factors"].tolist(), True)
        close(out, g[f"{tag}_out"], rtol=1e-5, atol=1e-6)
    dst = orc.chain_noise([g["comp_gauss"]], [1.0], True)
    src = orc.chain_noise([orc.uniform_noise(g["comp_uniform_u"])], [1.0], True)
    close(orc.composite_noise(dst, src, g["comp_mask_resized"], 0.8, True), g["comp_out"], rtol=1e-5, atol=1e-6)
    n1, n2 = g["blend_gauss"].clone(), orc.uniform_noise(g["blend_uniform_u"])
    close(orc.blended_noise(n1, n2, torch.full((1,), 0.3), "lerp", 1.2, True), g["blend_out"], rtol=1e-5, atol=1e-6)
    w = orc.blend_mask_weight(g["blendmask_maskdraw"].clone(), 0.1)
    close(w, g["blendmask_weight"])
    close(orc.blended_noise(g["blendmask_gauss"].clone(), orc.uniform_noise(g["blendmask_uniform_u"]), w, "inject", 1.0, True),
          g["blendmask_out"], rtol=1e-5, atol=1e-6)
    sp = orc.PerlinDraws(g["sched_perlin_base"], tuple(g["sched_perlin_angles"]))
    close(orc.scale_noise(orc.perlin_noise(sp), 1.0, normalized=True), g["sched_in"], rtol=1e-5, atol=1e-6)
    close(orc.scale_noise(g["sched_gauss"].clone(), 1.0, normalized=True), g["sched_out"], rtol=1e-5, atol=1e-6)


def fake_model(x, sigma, **_kw):
    s = sigma.reshape(-1, *([1] * (x.ndim - 1)))
    return x * 0.5 + torch.tanh(x) * (0.1 * s / (1.0 + s))


from tests.test_gpu_host_api import MOMENTUM_CASES  # noqa: E402  (same case table as the GPU parity test)


def to_cfg(kw):
    kw = dict(kw)
    if "momentum_mode" in kw:
        kw["mode"] = kw.pop("momentum_mode")
    return orc.MomentumCfg(**kw)


@pytest.mark.parametrize("kind", ["euler", "ancestral", "dpmpp"])
@pytest.mark.parametrize("name", list(MOMENTUM_CASES))
def test_momentum_traces(golden, kind, name):
    g = golden("momentum")
    it = iter(g["noise_bank"])
    ns = lambda s, sn: next(it).clone()  # noqa: E731
    trace = []
    cfg = to_cfg(MOMENTUM_CASES[name])
    if kind == "euler":
        orc.sonar_euler(fake_model, g["x0"].clone(), g["sigmas"], cfg, trace=trace)
    elif kind == "ancestral":
        orc.sonar_euler(fake_model, g["x0"].clone(), g["sigmas"], cfg, ancestral=True, eta=0.8, s_noise=1.1, noise_fn=ns, trace=trace)
    else:
        orc.sonar_dpmpp_sde(fake_model, g["x0"].clone(), g["sigmas"], cfg, eta=0.9, s_noise=1.05, noise_fn=ns, trace=trace)
    want = g[f"{kind}_{name}"]
    assert len(trace) == want.shape[0]
    for i, (x, _h) in enumerate(trace):
        close(x, want[i], rtol=2e-5, atol=2e-5)


def test_ancestral_step_formula():
    down, up = orc.ancestral_step(torch.tensor(10.0), torch.tensor(6.0), 1.0)
    assert abs(up.item() - min(6.0, math.sqrt(36 * (100 - 36) / 100))) < 1e-6
    assert abs(down.item() ** 2 + up.item() ** 2 - 36.0) < 1e-4
    assert orc.ancestral_step(3.0, 2.0, 0.0) == (2.0, 0.0)


# ------------------------------------------------------------------------------------------------ spatial power law, latent ops
POWERLAW_TYPES = {"white": dict(alpha=0.0, use_sign=True), "grey": dict(alpha=0.0),
                  "velvet": dict(alpha=1.0, use_sign=True, div_max_dims=(-3, -2, -1)),
                  "violet": dict(alpha=0.5, use_sign=True, div_max_dims=(-3, -2, -1))}
POWERLAW_ADV = {"a15_spatial": dict(alpha=1.5, div_max_dims=(-2, -1), use_sign=False, use_div_max_abs=True),
                "a07_all_noabs": dict(alpha=0.7, div_max_dims=(), use_sign=True, use_div_max_abs=False),
                "a2_batch": dict(alpha=2.0, div_max_dims=0, use_sign=False, use_div_max_abs=True),
                "a12_channel": dict(alpha=1.2, div_max_dims=1, use_sign=True, use_div_max_abs=True),
                "a03_height": dict(alpha=0.3, div_max_dims=2, use_sign=False, use_div_max_abs=True),
                "a25_width": dict(alpha=2.5, div_max_dims=3, use_sign=True, use_div_max_abs=True),
                "a1_none": dict(alpha=1.0, div_max_dims=None, use_sign=False, use_div_max_abs=True)}
LATENT_OP_CASES = {
    "lerp_half": dict(blend_mode="lerp", blend_strength=0.5, input_multiplier=1.0, output_multiplier=1.0, difference_multiplier=1.0),
    "inject_scaled": dict(blend_mode="inject", blend_strength=0.8, input_multiplier=0.5, output_multiplier=2.0, difference_multiplier=0.7),
    "lerp_big": dict(blend_mode="lerp", blend_strength=0.9, input_multiplier=1.5, output_multiplier=1.0, difference_multiplier=1.3),
    "subtract_b": dict(blend_mode="subtract_b", blend_strength=0.25, input_multiplier=1.0, output_multiplier=0.5, difference_multiplier=1.0),
}


def test_powerlaw(golden):
    g = golden("powerlaw")
    for name, kw in POWERLAW_TYPES.items():
        for normalized in (False, True):
            close(orc.scale_noise(orc.powerlaw_noise(g["draw"], **kw), 1.0, normalized=normalized), g[f"{name}_{int(normalized)}"])
    for name, kw in POWERLAW_ADV.items():
        close(orc.powerlaw_noise(g["draw"], **kw), g["adv_" + name])


def test_latent_ops(golden):
    g = golden("latent_ops")
    t = g["latent"]
    ops = (lambda latent: latent * 1.5 + 0.25, lambda latent: latent.abs() - 0.5)
    for name, kw in LATENT_OP_CASES.items():
        close(orc.latent_op_advanced(t, ops, **kw), g[f"adv_{name}"])
        close(ops[1](t), g[f"adv_{name}_disabled"])
    sigma = torch.tensor([3.0])
    close(orc.latent_op_noise(t, g["noise_raw"], sigma, False), g["noise_0"])
    close(orc.latent_op_noise(t, g["noise_raw"], sigma, True), g["noise_1"])


SPECTRAL_TYPES = ("onef_pinkish", "onef_greenish", "onef_pinkishgreenish", "onef_pinkish_mix", "onef_greenish_mix", "green_test",
                  "rainbow_mild", "rainbow_intense", "pink_old")
ONEF_ADV = {"sqrt": dict(alpha=0.25, k=2.0, hfac=2.0, wfac=0.5, use_sqrt=True), "nosqrt": dict(alpha=1.0, k=0.5, hfac=1.0, wfac=1.0, use_sqrt=False),
            "k0": dict(alpha=-1.0, k=0.0, hfac=1.0, wfac=1.0, use_sqrt=True),
            "neg_k": dict(alpha=1.0, k=-1.5, hfac=1.0, wfac=1.0, use_sqrt=True),
            "neg_base": dict(alpha=0.5, k=1.0, hfac=1.0, wfac=1.0, base_power=-2.0, use_sqrt=True),
            "neg_k_nosqrt": dict(alpha=1.0, k=-0.75, hfac=1.0, wfac=1.0, base_power=-1.5, use_sqrt=False)}


def test_spectral_gain_generators(golden):
    """FFT-based rows: pocketfft on another host CPU may round differently -> relative-to-peak tolerance 2e-6."""
    g = golden("spectral")
    d1, d2 = g["draw1"], g["draw2"]

    def near(a, b):
        close(a, b, rtol=0, atol=4e-6 * float(b.abs().max()))

    sn = lambda t: orc.scale_noise(t, 1.0, normalized=True)  # noqa: E731
    near(orc.onef_noise(d1, alpha=-0.5), g["onef_pinkish_0"])
    near(sn(orc.onef_noise(d1, alpha=0.5)), g["onef_greenish_1"])
    near((sn(orc.onef_noise(d1, alpha=0.5)) + sn(orc.onef_noise(d2, alpha=-0.5))).mul_(0.5), g["onef_pinkishgreenish_0"])
    near((sn(orc.onef_noise(d1, alpha=0.5)).mul_(-1.0) + sn(orc.onef_noise(d2, alpha=0.5))).mul_(0.5), g["onef_greenish_mix_0"])
    near(orc.green_test_noise(d1), g["green_test_0"])
    near(sn(orc.green_test_noise(d1)), g["green_test_1"])
    near((sn(orc.green_test_noise(d1)).mul_(0.75) + sn(orc.green_test_noise(d2)).mul_(0.5)).mul_(1.15), g["rainbow_intense_0"])
    close(g["pink_old_0"], d1)
    for name, kw in ONEF_ADV.items():
        near(orc.onef_noise(d1, **kw), g["adv_" + name])
    near(orc.spectral_filter(g["pf_draw"], g["pf_a_filter"]), g["pf_a_0"])
    near(sn(orc.spectral_filter(g["pf_draw"], g["pf_a_filter"])), g["pf_a_1"])


def test_onef_is_a_per_plane_filter():
    """The identity the HIP path relies on: fftn over all dims with an (h, w)-only gain == per-plane rfft2 filter."""
    torch.manual_seed(8)
    x = torch.randn(3, 4, 16, 32)
    want = orc.onef_noise(x, alpha=0.5, k=1.5)
    fx, fy = torch.meshgrid(torch.fft.fftfreq(16), torch.fft.fftfreq(32), indexing="ij")
    power = 1.5 / (fx**2 + fy**2) ** (-0.25)
    power[0, 0] = 1.0
    gain = (1.0 / torch.sqrt(power))[:, :17]
    got = torch.fft.irfft2(torch.fft.rfft2(x) * gain, s=(16, 32))
    close(got, want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("with_ref", [True, False])
@pytest.mark.parametrize("dims", [1, 2, 3])
@pytest.mark.parametrize("mtype", ["intensity", "frequency"])
def test_modulated_noise(golden, mtype, dims, with_ref):
    """ModulatedNoise (py/noise.py:762-1019): intensity is the same op sequence (tight); frequency multiplies the spectrum by the real
    boost directly instead of rebuilding it from magnitude and angle (1e-5)."""
    g = golden("modulated")
    torch.manual_seed(98)
    noise = orc.scale_noise(torch.randn(g["x"].shape), 1.0, normalized=True)  # the gaussian chain item, normalised
    ref_in = (g["latent"] if with_ref else g["x"]).clone()
    out = orc.modulated_noise(ref_in, noise, torch.tensor(9.0), torch.tensor(6.0), modulation_type=mtype, strength=1.5 if with_ref else -0.6,
                              modulation_dims=dims, factor=0.8, normalize_ref=True, normalize_result=with_ref)
    close(out, g[f"{mtype}_{dims}_{int(with_ref)}"], **({} if mtype == "intensity" else dict(rtol=1e-5, atol=1e-5)))
    if not with_ref:
        close(ref_in, g["x_after"])


def test_laplacian_and_power_old(golden):
    g = golden("basic_types")
    torch.manual_seed(22)
    n = torch.randn(3, 4, 8, 8)
    u = torch.empty(3, 4, 8, 8).uniform_(torch.finfo(torch.float32).eps - 1, 1)
    close(orc.laplacian_noise(n, u), g["laplacian_0"])
    torch.manual_seed(22)
    torch.randn(3, 4, 8, 8)
    raw = orc.power_old_noise(torch.rand(3, 4, 8, 8))
    close(raw, g["power_old_0"], rtol=1e-5, atol=1e-6)
    close(orc.scale_noise(raw.clone(), 1.0, normalized=True), g["power_old_1"], rtol=1e-5, atol=1e-6)


def test_studentt(golden):
    g = golden("basic_types")
    close(orc.studentt_noise(g["studentt_normal_draw"], g["studentt_gamma_draw"]), g["studentt_0"])
    torch.manual_seed(23)
    xn = torch.empty(3, 4, 8, 8).normal_()
    gm = torch._standard_gamma(torch.full((3, 4, 8, 8), 0.5))
    assert torch.equal(xn, g["studentt_normal_draw"]) and torch.equal(gm, g["studentt_gamma_draw"])
