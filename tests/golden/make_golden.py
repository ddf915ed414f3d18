"""Generates tests/golden/*.npz by running the REAL reference (imported from /root/reference through
oracle/ref_import.py) in the build container.  Container-only tool: the reference's Python never
travels; only these vectors (inputs incl. the captured base random draws, and expected outputs) do.

    python tests/golden/make_golden.py

Each case is also run through oracle/sonar_oracle.py here, and generation aborts if the oracle does
not reproduce the reference (bit-exact where the op sequence is identical).
"""
from __future__ import annotations

import importlib
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import sonar_oracle as orc  # noqa: E402
from oracle.ref_import import load_reference  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
ref = load_reference()
NT = ref.noise_generation.NoiseType


def save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def must_equal(a, b, what):
    if not torch.equal(a, b):
        raise SystemExit(f"oracle != reference for {what}: max |diff| = {(a - b).abs().max().item():.3e}")


# ------------------------------------------------------------------------------------------------ scale_noise
def gen_scale_noise():
    g = torch.Generator().manual_seed(11)
    cases = {}
    shape = (2, 4, 16, 16)
    n = math.prod(shape)
    thr = 2.5 / math.sqrt(n)
    specs = {
        "plain": (0.0, 1.0, 1.0),          # neither branch
        "shifted": (0.3, 1.0, 1.0),        # mean branch only (|mean| > thr)
        "scaled": (0.0, 0.8, 1.0),         # std branch only
        "both": (-0.4, 1.7, 0.6),          # both + factor
        "tiny_shift": (thr * 0.5, 1.0, 2.0),   # just inside the mean threshold
    }
    for name, (mu, sd, factor) in specs.items():
        x = torch.randn(shape, generator=g)
        x = (x - x.mean()) / x.std() * sd + mu
        dec = {}
        want = ref.utils.scale_noise(x.clone(), factor, normalized=True)
        got = orc.scale_noise(x.clone(), factor, normalized=True, decisions=dec)
        must_equal(want, got, f"scale_noise[{name}]")
        cases[f"{name}_in"] = x
        cases[f"{name}_out"] = want
        cases[f"{name}_meta"] = np.array([factor, float(dec["sub"]), float(dec["div"])], dtype=np.float64)
    x = torch.randn(shape, generator=g) * 1.3 + 0.2
    want = ref.utils.scale_noise(x.clone(), 0.7, normalized=True, normalize_dims=(-2, -1))
    must_equal(want, orc.scale_noise(x.clone(), 0.7, normalized=True, normalize_dims=(-2, -1)), "scale_noise dims")
    cases["dims_in"], cases["dims_out"] = x, want
    want = ref.utils.scale_noise(x.clone(), 1.9, normalized=False)
    cases["unnorm_out"] = want
    save("scale_noise", **cases)


# ------------------------------------------------------------------------------------------------ noise types
def ref_noise(noise_type, shape, seed, normalized, **kw):
    x = torch.zeros(shape)
    torch.manual_seed(seed)
    ns = ref.noise.get_noise_sampler(noise_type, x, 0.03, 14.6, seed=seed, cpu=True, factor=1.0, normalized=normalized, **kw)
    return ns(torch.tensor(14.6), torch.tensor(10.0))


def gen_perlin():
    cases = {}
    for tag, shape, seed, blend_mode in (("a", (2, 4, 16, 16), 0, "lerp"), ("b", (3, 4, 32, 24), 5, "lerp"),
                                         ("c", (1, 2, 8, 8), 7, "inject")):
        kw = {} if blend_mode == "lerp" else {"blend_mode": blend_mode}
        raw = ref_noise(NT.PERLIN, shape, seed, False, **kw)
        normed = ref_noise(NT.PERLIN, shape, seed, True, **kw)
        torch.manual_seed(seed)
        draws = orc.draw_perlin(shape, 2)
        mine = orc.perlin_noise(draws, 2.0, blend_mode)
        must_equal(raw, mine, f"perlin[{tag}] raw")
        must_equal(normed, orc.scale_noise(mine.clone(), 1.0, normalized=True), f"perlin[{tag}] normalised")
        cases[f"{tag}_base"] = draws.base
        cases[f"{tag}_angles"] = torch.stack(draws.angles)
        cases[f"{tag}_raw"] = raw
        cases[f"{tag}_out"] = normed
        cases[f"{tag}_seed"] = np.array(seed)
        cases[f"{tag}_blend"] = np.array(blend_mode)
    save("perlin", **cases)


def gen_pyramid():
    cases = {}
    for tag, shape, seed, nt, discount, mode in (("a", (2, 4, 32, 32), 0, NT.PYRAMID, 0.7, "bilinear"),
                                                ("b", (1, 4, 64, 48), 3, NT.PYRAMID, 0.7, "bilinear"),
                                                ("c", (2, 3, 32, 32), 9, NT.PYRAMID_DISCOUNT5, 0.5, "bilinear"),
                                                ("d", (1, 4, 32, 32), 4, NT.PYRAMID_AREA, 0.7, "area")):
        raw = ref_noise(nt, shape, seed, False)
        normed = ref_noise(nt, shape, seed, True)
        torch.manual_seed(seed)
        draws = orc.draw_pyramid(shape, 10)
        mine = orc.pyramid_noise(draws, discount, mode)
        must_equal(raw, mine, f"pyramid[{tag}] raw")
        must_equal(normed, orc.scale_noise(mine.clone(), 1.0, normalized=True), f"pyramid[{tag}] normalised")
        assert orc.pyramid_sizes(shape[-2], shape[-1], draws.rs) == [tuple(l.shape[-2:]) for l in draws.levels]
        cases[f"{tag}_base"] = draws.base
        cases[f"{tag}_rs"] = np.array(draws.rs, dtype=np.float64)
        for i, lvl in enumerate(draws.levels):
            cases[f"{tag}_level{i}"] = lvl
        cases[f"{tag}_nlevels"] = np.array(len(draws.levels))
        cases[f"{tag}_raw"] = raw
        cases[f"{tag}_out"] = normed
        cases[f"{tag}_seed"] = np.array(seed)
        cases[f"{tag}_discount"] = np.array(discount)
        cases[f"{tag}_mode"] = np.array(mode)
    save("pyramid", **cases)


def gen_basic_types():
    """gaussian / uniform through the NoiseType registry (plumbing: RNG order, factor, normalisation)."""
    cases = {}
    for name, nt in (("gaussian", NT.GAUSSIAN), ("uniform", NT.UNIFORM)):
        for normalized in (False, True):
            shape = (2, 4, 8, 8)
            out = ref_noise(nt, shape, 21, normalized)
            cases[f"{name}_{int(normalized)}"] = out
    torch.manual_seed(21)
    cases["gaussian_draw"] = torch.randn(2, 4, 8, 8)
    torch.manual_seed(21)
    cases["uniform_draw"] = torch.rand(2, 4, 8, 8)
    must_equal(cases["gaussian_0"], cases["gaussian_draw"], "gaussian")
    must_equal(cases["uniform_0"], orc.uniform_noise(cases["uniform_draw"]), "uniform")
    # laplacian (randn, then Laplace.rsample's uniform) and power_old (randn dropped, then rand), both on the global generator
    for name, nt in (("laplacian", NT.LAPLACIAN), ("power_old", NT.POWER_OLD)):
        for normalized in (False, True):
            cases[f"{name}_{int(normalized)}"] = ref_noise(nt, (3, 4, 8, 8), 22, normalized)
    torch.manual_seed(22)
    n = torch.randn(3, 4, 8, 8)
    u = torch.empty(3, 4, 8, 8).uniform_(torch.finfo(torch.float32).eps - 1, 1)
    must_equal(cases["laplacian_0"], orc.laplacian_noise(n, u), "laplacian")
    torch.manual_seed(22)
    torch.randn(3, 4, 8, 8)
    must_equal(cases["power_old_0"], orc.power_old_noise(torch.rand(3, 4, 8, 8)), "power_old")
    # studentt: X = empty.normal_(), then torch._standard_gamma(df / 2), both on the global generator
    for normalized in (False, True):
        cases[f"studentt_{int(normalized)}"] = ref_noise(NT.STUDENTT, (3, 4, 8, 8), 23, normalized)
    cases["studentt_df3"] = ref_noise(NT.STUDENTT, (3, 4, 8, 8), 23, False, df=3, quantile_fac=0.9, pow_fac=0.75, scale=0.5, loc=0.1, nq_fac=0.8)
    torch.manual_seed(23)
    xn = torch.empty(3, 4, 8, 8).normal_()
    gm = torch._standard_gamma(torch.full((3, 4, 8, 8), 0.5))
    cases["studentt_normal_draw"], cases["studentt_gamma_draw"] = xn, gm
    must_equal(cases["studentt_0"], orc.studentt_noise(xn, gm), "studentt")
    torch.manual_seed(23)
    xn3 = torch.empty(3, 4, 8, 8).normal_()
    gm3 = torch._standard_gamma(torch.full((3, 4, 8, 8), 1.5))
    must_equal(cases["studentt_df3"], orc.studentt_noise(xn3, gm3, loc=0.1, scale=0.5, df=3, quantile_fac=0.9, pow_fac=0.75, nq_fac=0.8), "studentt df3")
    save("basic_types", **cases)


# ------------------------------------------------------------------------------------------------ power noise
FILTER_CASES = {
    "white": dict(alpha=0.0),
    "pink": dict(alpha=1.0),
    "half": dict(alpha=0.5),
    "brown": dict(alpha=2.0),
    "blue": dict(alpha=-0.5),
    "band": dict(alpha=1.0, min_freq=0.1, max_freq=0.4),
    "rot_stretch": dict(alpha=1.0, rotate=30.0, stretch=2.0),
    "squash": dict(alpha=0.5, stretch=0.5),
    "pnorm1": dict(alpha=1.0, pnorm=1.0),
}


def gen_power_filter():
    cases = {}
    PF = ref.powernoise.PowerFilter
    for name, kw in FILTER_CASES.items():
        for hw in ((32, 32), (16, 24)):
            shape = (1, 4, *hw)
            raw = PF(**kw).build(shape)
            mine = orc.power_filter_build(shape, **kw)
            must_equal(raw, mine, f"filter[{name}] build")
            for mix, nf in ((1.0, 1.0), (0.6, 1.0), (1.0, 0.5)):
                want = PF.normalize(raw.clone(), shape, mix=mix, normalization_factor=nf)
                must_equal(want, orc.power_filter_normalize(mine.clone(), shape, mix, nf), f"filter[{name}] normalise")
                cases[f"{name}_{hw[0]}x{hw[1]}_mix{mix}_nf{nf}"] = want
            cases[f"{name}_{hw[0]}x{hw[1]}_raw"] = raw
    # the benchmark filter (cfg2): alpha = 1, 128 x 128
    shape = (1, 4, 128, 128)
    want = PF.normalize(PF(alpha=1.0, max_freq=0.7071).build(shape), shape)
    must_equal(want, orc.power_filter_normalize(orc.power_filter_build(shape, alpha=1.0, max_freq=0.7071), shape), "cfg2 filter")
    cases["cfg2_128x128"] = want
    a = PF(alpha=1.0).build((1, 4, 32, 32))
    b = PF(alpha=0.0, min_freq=0.2, max_freq=0.3).build((1, 4, 32, 32))
    for mode in ("max", "min", "add", "sub", "mul"):
        want = PF.compose(a.clone(), b.clone(), mode)
        must_equal(want, orc.power_filter_compose(a.clone(), b.clone(), mode), f"compose {mode}")
        cases[f"compose_{mode}"] = want
    corr = torch.tensor([1.0, 1.0, 1.0, 1.0, 1.0, 1.0])
    for cm in (0.0, 0.25, -0.2):
        want = ref.powernoise.ChannelMixer(4, cm, corr).mixer
        must_equal(want, orc.channel_mixer(4, cm, corr), f"mixer {cm}")
        cases[f"mixer_{cm}"] = want
    corr2 = torch.tensor([0.5, -0.3, 0.8])
    want = ref.powernoise.ChannelMixer(4, 0.4, corr2).mixer
    must_equal(want, orc.channel_mixer(4, 0.4, corr2), "mixer partial correlation")
    cases["mixer_partial"] = want
    save("power_filter", **cases)


def ref_power_item(**kw):
    args = dict(time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                common_mode=0.0, channel_correlation="1,1,1,1,1,1")
    args.update(kw)
    return ref.powernoise.PowerNoiseItem(1.0, **args)


def gen_power_noise():
    cases = {}
    for tag, shape, seed, kw, normalized in (
        ("cfg2", (1, 4, 128, 128), 0, {}, True),
        ("b", (2, 4, 32, 32), 1, {"alpha": 0.5}, True),
        ("c", (2, 4, 64, 64), 2, {"alpha": 2.0, "common_mode": 0.25}, True),
        ("d", (3, 4, 32, 64), 3, {"alpha": 1.0, "min_freq": 0.1, "max_freq": 0.4}, False),
        ("e", (1, 2, 16, 16), 4, {"alpha": 0.0, "mix": 0.5}, True),
        ("np2", (2, 4, 40, 56), 5, {"alpha": 1.0}, True),                       # not powers of two (general-size kernels)
        ("np2_rot", (1, 4, 52, 76), 6, {"alpha": 1.5, "rotate": 20.0, "stretch": 1.5, "common_mode": 0.1}, True),  # quarter-size 832 x 1216 px
        ("odd", (2, 4, 27, 35), 7, {"alpha": 1.0, "common_mode": 0.1}, True),   # odd height and width (1080 lines -> 135 rows): direct DFT passes
        ("sdxl_portrait", (1, 4, 104, 152), 8, {"alpha": 1.0}, True),           # 832 x 1216 px at full size: codelets 13 x 8 and 19 x 4
    ):
        item = ref_power_item(**kw)
        x = torch.zeros(shape)
        torch.manual_seed(seed)
        want = item.make_noise_sampler(x, None, None, seed=None, cpu=True, normalized=normalized)(None, None)
        filt = item.make_filter(shape)
        torch.manual_seed(seed)
        z = orc.draw_power(shape)
        fkw = {k: v for k, v in kw.items() if k in ("alpha", "min_freq", "max_freq", "stretch", "rotate", "pnorm")}
        fkw.setdefault("max_freq", 0.7071)
        fkw.setdefault("alpha", 1.0)
        mine_f = orc.power_filter_normalize(orc.power_filter_build(shape, **fkw), shape, mix=kw.get("mix", 1.0))
        must_equal(filt, mine_f, f"power[{tag}] filter")
        cm = kw.get("common_mode", 0.0)
        mixer = orc.channel_mixer(shape[1], cm, torch.ones(6))
        dec, pre = {}, []
        mine = orc.power_noise(z, mine_f, shape, mixer, 1.0, normalized, decisions=dec, pre_norm=pre)
        must_equal(want, mine, f"power[{tag}]")
        cases[f"{tag}_z"] = torch.view_as_real(z)
        cases[f"{tag}_filter"] = filt
        cases[f"{tag}_mixer"] = mixer
        cases[f"{tag}_pre"] = pre[0]
        cases[f"{tag}_out"] = want
        cases[f"{tag}_seed"] = np.array(seed)
        cases[f"{tag}_normalized"] = np.array(normalized)
        cases[f"{tag}_branches"] = np.array([dec.get("sub", False), dec.get("div", False)])
    save("power_noise", **cases)


# ------------------------------------------------------------------------------------------------ composition
def gen_composition():
    cases = {}
    N = ref.noise
    shape = (2, 4, 16, 16)
    x = torch.zeros(shape)

    def chain_of(*items):
        c = N.CustomNoiseChain()
        for it in items:
            c.add(it)
        return c

    def item(nt, f):
        return N.CustomNoiseItem(f, noise_type=nt)

    # chain: 0.6 gaussian - 0.3 uniform + 0.5 perlin, normalised; and rescaled(1.0)
    chain = chain_of(item(NT.GAUSSIAN, 0.6), item(NT.UNIFORM, -0.3), item(NT.PERLIN, 0.5))
    for tag, ch in (("chain", chain), ("chain_rescaled", chain.rescaled(1.0))):
        torch.manual_seed(31)
        want = ch.make_noise_sampler(x, 0.03, 14.6, seed=31, cpu=True, normalized=True)(torch.tensor(5.0), torch.tensor(4.0))
        torch.manual_seed(31)
        g = orc.draw_gaussian(shape)
        u = torch.rand(shape)
        p = orc.draw_perlin(shape, 2)
        factors = [i.factor for i in ch.items]
        mine = orc.chain_noise([g, orc.uniform_noise(u), orc.perlin_noise(p)], factors, True)
        must_equal(want, mine, tag)
        cases[f"{tag}_out"] = want
        cases[f"{tag}_factors"] = np.array(factors)
    cases["chain_gauss"], cases["chain_uniform_u"], cases["chain_perlin_base"] = g, u, p.base
    cases["chain_perlin_angles"] = torch.stack(p.angles)

    # composite: dst gaussian, src uniform, smooth mask, all normalised
    mask = torch.linspace(0, 1, 8 * 8).reshape(1, 8, 8)
    comp = N.CompositeNoise(0.8, dst_noise=chain_of(item(NT.GAUSSIAN, 1.0)), src_noise=chain_of(item(NT.UNIFORM, 1.0)),
                            normalize_dst=True, normalize_src=True, normalize_result=True, mask=mask)
    torch.manual_seed(32)
    want = comp.make_noise_sampler(x, 0.03, 14.6, seed=32, cpu=True, normalized=True)(torch.tensor(5.0), torch.tensor(4.0))
    torch.manual_seed(32)
    g = orc.draw_gaussian(shape)
    u = torch.rand(shape)
    m = torch.nn.functional.interpolate(mask.reshape(-1, 1, 8, 8), size=shape[-2:], mode="bilinear").repeat(shape[0], 1, 1, 1)
    dst = orc.chain_noise([g], [1.0], True)
    src = orc.chain_noise([orc.uniform_noise(u)], [1.0], True)
    must_equal(want, orc.composite_noise(dst, src, m, 0.8, True), "composite")
    cases.update(comp_out=want, comp_gauss=g, comp_uniform_u=u, comp_mask=mask, comp_mask_resized=m)

    # blended: lerp(gaussian, uniform, 0.3) normalised; and mask-driven weights
    bl = N.BlendedNoise(1.2, normalize=True, blend_function=ref.utils.BLENDING_MODES["lerp"],
                        custom_noise_1=chain_of(item(NT.GAUSSIAN, 1.0)), custom_noise_2=chain_of(item(NT.UNIFORM, 1.0)),
                        noise_2_percent=0.3)
    torch.manual_seed(33)
    want = bl.make_noise_sampler(x, 0.03, 14.6, seed=33, cpu=True, normalized=True)(torch.tensor(5.0), torch.tensor(4.0))
    torch.manual_seed(33)
    g = orc.draw_gaussian(shape)
    u = torch.rand(shape)
    n1 = orc.chain_noise([g], [1.0], False)
    n2 = orc.chain_noise([orc.uniform_noise(u)], [1.0], False)
    must_equal(want, orc.blended_noise(n1, n2, torch.full((1,), 0.3), "lerp", 1.2, True), "blended")
    cases.update(blend_out=want, blend_gauss=g, blend_uniform_u=u)
    blm = N.BlendedNoise(1.0, normalize=True, blend_function=ref.utils.BLENDING_MODES["inject"],
                         custom_noise_1=chain_of(item(NT.GAUSSIAN, 1.0)), custom_noise_2=chain_of(item(NT.UNIFORM, 1.0)),
                         custom_noise_mask=chain_of(item(NT.GAUSSIAN, 1.0)), noise_2_percent=0.1)
    torch.manual_seed(34)
    want = blm.make_noise_sampler(x, 0.03, 14.6, seed=34, cpu=True, normalized=True)(torch.tensor(5.0), torch.tensor(4.0))
    torch.manual_seed(34)
    g = orc.draw_gaussian(shape)
    u = torch.rand(shape)
    gm = orc.draw_gaussian(shape)
    wgt = orc.blend_mask_weight(gm.clone(), 0.1)
    must_equal(want, orc.blended_noise(g.clone(), orc.uniform_noise(u), wgt, "inject", 1.0, True), "blended mask")
    cases.update(blendmask_out=want, blendmask_gauss=g, blendmask_uniform_u=u, blendmask_maskdraw=gm, blendmask_weight=wgt)

    # scheduled: perlin inside [2, 10], gaussian fallback outside, normalised
    sch = N.ScheduledNoise(1.0, noise=chain_of(item(NT.PERLIN, 1.0)), start_sigma=10.0, end_sigma=2.0, normalize=True,
                           fallback_noise=chain_of(item(NT.GAUSSIAN, 1.0)))
    ns = sch.make_noise_sampler(x, 0.03, 14.6, seed=35, cpu=True, normalized=True)
    torch.manual_seed(35)
    inside = ns(torch.tensor(5.0), torch.tensor(4.0))
    outside = ns(torch.tensor(12.0), torch.tensor(11.0))
    torch.manual_seed(35)
    p = orc.draw_perlin(shape, 2)
    g = orc.draw_gaussian(shape)
    must_equal(inside, orc.scale_noise(orc.chain_noise([orc.perlin_noise(p)], [1.0], False), 1.0, normalized=True), "scheduled in")
    must_equal(outside, orc.scale_noise(orc.chain_noise([g], [1.0], False), 1.0, normalized=True), "scheduled out")
    cases.update(sched_in=inside, sched_out=outside, sched_perlin_base=p.base, sched_perlin_angles=torch.stack(p.angles), sched_gauss=g)
    save("composition", **cases)


# ------------------------------------------------------------------------------------------------ momentum samplers
def fake_model(x, sigma, **_kw):
    # smooth, non-linear in x, sigma-dependent: exercises every elementwise path deterministically
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


def to_oracle_cfg(kw):
    kw = dict(kw)
    if "momentum_mode" in kw:
        kw["mode"] = kw.pop("momentum_mode")
    return orc.MomentumCfg(**kw)


def gen_momentum():
    S = ref.sonar
    cases = {}
    shape = (2, 4, 8, 8)
    torch.manual_seed(3)
    x0 = torch.randn(shape) * 14.6
    sigmas = torch.cat((torch.linspace(14.6, 0.03, 7), torch.zeros(1)))
    cases["x0"], cases["sigmas"] = x0, sigmas
    noise_bank = torch.randn(16, *shape)
    cases["noise_bank"] = noise_bank

    def bank_sampler():
        it = iter(noise_bank)
        return lambda s, sn: next(it).clone()

    for name, kw in MOMENTUM_CASES.items():
        for kind in ("euler", "ancestral", "dpmpp"):
            trace = []
            cb = lambda d: trace.append(d["x"].clone())  # noqa: E731
            extra = {"seed": 0}
            if kind == "euler":
                want = S.SonarEuler.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, bank_sampler(), None, dict(kw))
                mine_trace = []
                mine = orc.sonar_euler(fake_model, x0.clone(), sigmas, to_oracle_cfg(kw), trace=mine_trace)
            elif kind == "ancestral":
                want = S.SonarEulerAncestral.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, None, dict(kw), 0.8, 1.1, bank_sampler())
                mine_trace = []
                mine = orc.sonar_euler(fake_model, x0.clone(), sigmas, to_oracle_cfg(kw), ancestral=True, eta=0.8, s_noise=1.1,
                                       noise_fn=bank_sampler(), trace=mine_trace)
            else:
                want = S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, None, dict(kw), 0.9, 1.05, bank_sampler())
                mine_trace = []
                mine = orc.sonar_dpmpp_sde(fake_model, x0.clone(), sigmas, to_oracle_cfg(kw), eta=0.9, s_noise=1.05,
                                           noise_fn=bank_sampler(), trace=mine_trace)
            must_equal(want, mine, f"{kind}[{name}] final")
            for i, (a, b) in enumerate(zip(trace, mine_trace)):
                must_equal(a, b[0], f"{kind}[{name}] step {i}")
            cases[f"{kind}_{name}"] = torch.stack(trace)
    save("momentum", **cases)


# ------------------------------------------------------------------------------------------------ node ABI (boundary facts)
def gen_node_abi():
    """Names / socket types / defaults of every registered node: data only, no code."""
    import json

    out = {}
    for key, cls in ref.nodes.NODE_CLASS_MAPPINGS.items():
        try:
            spec = cls.INPUT_TYPES()
        except Exception:  # noqa: BLE001  (first call probes optional third-party packs and may fail once)
            spec = cls.INPUT_TYPES()
        inputs = {}
        for section in ("required", "optional"):
            for name, item in spec.get(section, {}).items():
                typ = item[0]
                opts = item[1] if len(item) > 1 else {}
                entry = {"section": section, "type": list(typ) if isinstance(typ, (tuple, list)) else str(typ)}
                for k in ("default", "min", "max"):
                    if k in opts and isinstance(opts[k], (int, float, str, bool)):
                        entry[k] = opts[k]
                inputs[name] = entry
        out[key] = {"returns": list(cls.RETURN_TYPES), "function": cls.FUNCTION, "category": cls.CATEGORY,
                    "output_node": bool(getattr(cls, "OUTPUT_NODE", False)), "inputs": inputs}
    with open(os.path.join(OUT, "node_abi.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("node_abi.json", len(out), "nodes")




# ------------------------------------------------------------------------------------------------ entry-point nodes
def gen_entry_nodes():
    M = ref.nodes.NODE_CLASS_MAPPINGS
    N = ref.noise
    cases = {}
    lat0 = torch.zeros(2, 4, 16, 16)
    (out,) = M["NoisyLatentLike"].go(noise_type="perlin", seed=5, latent={"samples": lat0}, multiplier=0.7, add_to_latent=False,
                                     repeat_batch=2, cpu_noise=True, normalize=True)
    cases["nll_perlin"] = out["samples"]
    chain = N.CustomNoiseChain()
    chain.add(N.CustomNoiseItem(0.6, noise_type=NT.GAUSSIAN))
    chain.add(N.CustomNoiseItem(-0.3, noise_type=NT.UNIFORM))
    g = torch.Generator().manual_seed(77)
    lat1 = torch.randn(2, 4, 16, 16, generator=g)
    (out,) = M["NoisyLatentLike"].go(noise_type="gaussian", seed=6, latent={"samples": lat1}, multiplier=1.3, add_to_latent=True,
                                     repeat_batch=1, cpu_noise=True, normalize=True, custom_noise_opt=chain)
    cases["nll_chain_latent"], cases["nll_chain_out"] = lat1, out["samples"]
    (nobj,) = M["SONAR_CUSTOM_NOISE to NOISE"].go(custom_noise=chain, seed=9, cpu_noise=True, normalize=True, multiplier=0.5)
    cases["noise_plain"] = nobj.generate_noise({"samples": torch.zeros(3, 4, 8, 8)})
    cases["noise_batch_index"] = nobj.generate_noise({"samples": torch.zeros(3, 4, 8, 8), "batch_index": [2, 0, 2]})
    save("entry_nodes", **cases)


# ------------------------------------------------------------------------------------------------ spatial power law + latent ops
POWERLAW_TYPES = {"white": dict(alpha=0.0, use_sign=True), "grey": dict(alpha=0.0), "velvet": dict(alpha=1.0, use_sign=True, div_max_dims=(-3, -2, -1)),
                  "violet": dict(alpha=0.5, use_sign=True, div_max_dims=(-3, -2, -1))}
POWERLAW_ADV = {"a15_spatial": dict(alpha=1.5, div_max_dims=(-2, -1), use_sign=False, use_div_max_abs=True),
                "a07_all_noabs": dict(alpha=0.7, div_max_dims=(), use_sign=True, use_div_max_abs=False),
                "a2_batch": dict(alpha=2.0, div_max_dims=0, use_sign=False, use_div_max_abs=True),
                "a12_channel": dict(alpha=1.2, div_max_dims=1, use_sign=True, use_div_max_abs=True),
                "a03_height": dict(alpha=0.3, div_max_dims=2, use_sign=False, use_div_max_abs=True),
                "a25_width": dict(alpha=2.5, div_max_dims=3, use_sign=True, use_div_max_abs=True),
                "a1_none": dict(alpha=1.0, div_max_dims=None, use_sign=False, use_div_max_abs=True)}


def gen_powerlaw():
    cases = {}
    shape = (3, 4, 16, 12)
    torch.manual_seed(33)
    draw = torch.randn(shape)
    cases["draw"] = draw
    for name, kw in POWERLAW_TYPES.items():
        for normalized in (False, True):
            out = ref_noise(getattr(NT, name.upper()), shape, 33, normalized)
            mine = orc.scale_noise(orc.powerlaw_noise(draw, **kw), 1.0, normalized=normalized)
            must_equal(out, mine, f"powerlaw type {name}")
            cases[f"{name}_{int(normalized)}"] = out
    for name, kw in POWERLAW_ADV.items():
        item = ref.noise.AdvancedPowerLawNoise(1.0, **kw)
        torch.manual_seed(33)
        out = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=33, cpu=True, normalized=False)(torch.tensor(14.6), torch.tensor(10.0))
        must_equal(out, orc.powerlaw_noise(draw, **kw), f"powerlaw adv {name}")
        cases["adv_" + name] = out
    save("powerlaw", **cases)


LATENT_OP_CASES = {
    "lerp_half": dict(blend_mode="lerp", blend_strength=0.5, input_multiplier=1.0, output_multiplier=1.0, difference_multiplier=1.0),
    "inject_scaled": dict(blend_mode="inject", blend_strength=0.8, input_multiplier=0.5, output_multiplier=2.0, difference_multiplier=0.7),
    "lerp_big": dict(blend_mode="lerp", blend_strength=0.9, input_multiplier=1.5, output_multiplier=1.0, difference_multiplier=1.3),
    "subtract_b": dict(blend_mode="subtract_b", blend_strength=0.25, input_multiplier=1.0, output_multiplier=0.5, difference_multiplier=1.0),
}


def gen_latent_ops():
    lo = ref.latent_ops
    cases = {}
    torch.manual_seed(44)
    t = torch.randn(2, 4, 8, 8)
    cases["latent"] = t
    op1 = lambda latent: latent * 1.5 + 0.25  # noqa: E731  (plain LATENT_OPERATION callables)
    op2 = lambda latent: latent.abs() - 0.5  # noqa: E731
    for name, kw in LATENT_OP_CASES.items():
        adv = lo.SonarLatentOperationAdvanced(ops=(lo.SonarLatentOperation(op=op1), lo.SonarLatentOperation(op=op2)), start_sigma=10.0,
                                              end_sigma=1.0, op_alt=lo.SonarLatentOperation(op=op2), **kw)
        out = adv(t.clone(), sigma=torch.tensor([5.0]))
        must_equal(out, orc.latent_op_advanced(t, (op1, op2), **kw), f"latent op advanced {name}")
        cases[f"adv_{name}"] = out
        cases[f"adv_{name}_disabled"] = adv(t.clone(), sigma=torch.tensor([12.0]))  # outside [1, 10] -> op_alt
    chain = ref.noise.CustomNoiseChain()
    chain.add(ref.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    for scale_to_sigma in (False, True):
        op = lo.SonarLatentOperationNoise(custom_noise=chain, scale_to_sigma=scale_to_sigma, cpu_noise=True, normalize=True)
        torch.manual_seed(45)
        out = op(t.clone(), sigma=torch.tensor([3.0]))
        cases[f"noise_{int(scale_to_sigma)}"] = out
    # the noise the op drew: same RNG sequence (randint for the seed, then the sampler)
    torch.manual_seed(45)
    seed = torch.randint(1, 1 << 31, (), device="cpu").item()
    raw = chain.make_noise_sampler(t, sigma_min=None, sigma_max=None, normalized=True, seed=seed, cpu=True)(torch.tensor([3.0]), torch.tensor([3.0]))
    cases["noise_raw"] = raw
    must_equal(cases["noise_1"], orc.latent_op_noise(t, raw, torch.tensor([3.0]), True), "latent op noise")
    setseed = lo.SonarLatentOperationSetSeed(op=lo.SonarLatentOperationNoise(custom_noise=chain, cpu_noise=True), seed=77, restore_rng_state=True)
    torch.manual_seed(1)
    before = torch.random.get_rng_state()
    cases["setseed"] = setseed(latent=t.clone(), sigma=torch.tensor([2.0]))
    assert torch.equal(before, torch.random.get_rng_state())
    save("latent_ops", **cases)


# ------------------------------------------------------------------------------------------------ spectral-gain generators
SPECTRAL_TYPES = ("onef_pinkish", "onef_greenish", "onef_pinkishgreenish", "onef_pinkish_mix", "onef_greenish_mix", "green_test",
                  "rainbow_mild", "rainbow_intense", "pink_old")
ONEF_ADV = {"sqrt": dict(alpha=0.25, k=2.0, hfac=2.0, wfac=0.5, use_sqrt=True), "nosqrt": dict(alpha=1.0, k=0.5, hfac=1.0, wfac=1.0, use_sqrt=False),
            "k0": dict(alpha=-1.0, k=0.0, hfac=1.0, wfac=1.0, use_sqrt=True),
            # negative spectral power: its complex square root is imaginary, those frequencies drop out of the real part (use_sqrt), or
            # the gain is simply negative (no square root)
            "neg_k": dict(alpha=1.0, k=-1.5, hfac=1.0, wfac=1.0, use_sqrt=True),
            "neg_base": dict(alpha=0.5, k=1.0, hfac=1.0, wfac=1.0, base_power=-2.0, use_sqrt=True),
            "neg_k_nosqrt": dict(alpha=1.0, k=-0.75, hfac=1.0, wfac=1.0, base_power=-1.5, use_sqrt=False)}


def gen_spectral():
    cases = {}
    shape = (2, 4, 32, 32)
    torch.manual_seed(51)
    d1, d2 = torch.randn(shape), torch.randn(shape)
    cases["draw1"], cases["draw2"] = d1, d2
    for name in SPECTRAL_TYPES:
        for normalized in (False, True):
            cases[f"{name}_{int(normalized)}"] = ref_noise(getattr(NT, name.upper()), shape, 51, normalized)
    must_equal(cases["onef_pinkish_0"], orc.onef_noise(d1, alpha=-0.5), "onef_pinkish")
    must_equal(cases["onef_greenish_1"], orc.scale_noise(orc.onef_noise(d1, alpha=0.5), 1.0, normalized=True), "onef_greenish")
    sn = lambda t: orc.scale_noise(t, 1.0, normalized=True)  # noqa: E731  (sub-generators of a mix normalise themselves)
    must_equal(cases["onef_pinkish_mix_0"], (sn(orc.onef_noise(d1, alpha=-0.5)).mul_(-1.0) + sn(orc.onef_noise(d2, alpha=-0.5))).mul_(0.5), "onef_pinkish_mix")
    must_equal(cases["green_test_0"], orc.green_test_noise(d1), "green_test")
    must_equal(cases["rainbow_mild_0"], (sn(orc.green_test_noise(d1)).mul_(0.55) + sn(orc.green_test_noise(d2)).mul_(0.7)).mul_(1.15), "rainbow_mild")
    for name, kw in ONEF_ADV.items():
        item = ref.noise.Advanced1fNoise(1.0, **kw)
        torch.manual_seed(51)
        out = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=51, cpu=True, normalized=False)(torch.tensor(14.6), torch.tensor(10.0))
        must_equal(out, orc.onef_noise(d1, **kw), f"onef adv {name}")
        cases["adv_" + name] = out
    # PowerFilterNoiseItem: a power filter over a gaussian chain (time_brownian=True is what the node passes)
    pn = ref.powernoise
    chain = ref.noise.CustomNoiseChain()
    chain.add(ref.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    for tag, fkw, ikw in (("pf_a", dict(alpha=1.0, max_freq=0.5), dict(mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")),
                          ("pf_b", dict(alpha=-0.5, min_freq=0.1, max_freq=0.7071, rotate=20.0, stretch=1.5), dict(mix=0.7, common_mode=0.25, channel_correlation="1,0.5,0.2,1,0.3,0.1"))):
        filt = pn.PowerFilter(**fkw)
        item = pn.PowerFilterNoiseItem(1.0, noise=chain, normalize_noise=None, normalize_result=None, time_brownian=True, power_filter=filt,
                                       filter_norm_factor=1.0, **ikw)
        for normalized in (False, True):
            torch.manual_seed(52)
            out = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=52, cpu=True, normalized=normalized)(torch.tensor(14.6), torch.tensor(10.0))
            cases[f"{tag}_{int(normalized)}"] = out
        cases[f"{tag}_filter"] = item.make_filter(shape)
    torch.manual_seed(52)
    cases["pf_draw"] = torch.randn(shape)
    filt_a = cases["pf_a_filter"]
    must_equal(cases["pf_a_0"], orc.spectral_filter(cases["pf_draw"], filt_a), "power filter noise item")
    save("spectral", **cases)


# ------------------------------------------------------------------------------------------------ cfg5 in miniature (end to end)
def gen_cfg5():
    """BASELINE.json cfg5 at test size, through the REAL reference end to end: scheduled (power-law | gaussian fallback) + Perlin
    chain, normalised, feeding SonarDPMPPSDE with momentum on a Flux-shaped latent (16 channels).  Inputs, per-step x and the
    final latent are stored; the GPU test drives the same calls with the latent on the MI355X (replay mode)."""
    S, N, pn = ref.sonar, ref.noise, ref.powernoise
    shape = (2, 16, 32, 32)
    torch.manual_seed(61)
    x0 = torch.randn(shape) * 10.0
    sigmas = torch.cat((torch.linspace(10.0, 0.5, 6), torch.zeros(1)))

    def build_chain(mod_noise, mod_pn):
        inner = mod_noise.CustomNoiseChain()
        inner.add(mod_pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                                        mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1"))
        fallback = mod_noise.CustomNoiseChain()
        fallback.add(mod_noise.CustomNoiseItem(1.0, noise_type="gaussian"))
        chain = mod_noise.CustomNoiseChain()
        chain.add(mod_noise.ScheduledNoise(0.7, noise=inner, start_sigma=20.0, end_sigma=4.0, normalize=None, fallback_noise=fallback))
        chain.add(mod_noise.CustomNoiseItem(0.5, noise_type="perlin"))
        return chain

    chain = build_chain(N, pn)
    params = dict(momentum=0.9, momentum_hist=0.7, direction=1.0, momentum_mode="NEW", init="SAMPLE_NORM")
    torch.manual_seed(62)
    ns = chain.make_noise_sampler(x0, sigmas[sigmas > 0].min(), sigmas.max(), seed=5, cpu=True, normalized=True)
    trace = []
    out = S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas, {"seed": 5}, lambda d: trace.append(d["x"].clone()), True, None,
                                  dict(params), 0.9, 1.05, ns)
    save("cfg5", x0=x0, sigmas=sigmas, trace_steps=torch.tensor([1, 3]), trace=torch.stack([trace[1], trace[3]]), out=out)


def cfg5_full_chain(mod_noise, mod_pn):
    """BASELINE.json cfg5's noise at the Flux latent's real size: ScheduledNoise over a power-law + Perlin + third-source chain with a
    Gaussian fallback, normalised.  The third source is Gaussian here (the configured Brownian one delegates to torchsde, which the
    image lacks): the chain arithmetic, the scheduling and the normalisation around it are the configured ones."""
    inner = mod_noise.CustomNoiseChain()
    inner.add(mod_pn.PowerNoiseItem(0.5, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                                    mix=1.0, common_mode=0.0, channel_correlation="1"))
    inner.add(mod_noise.CustomNoiseItem(0.3, noise_type="perlin"))
    inner.add(mod_noise.CustomNoiseItem(0.2, noise_type="gaussian"))
    fallback = mod_noise.CustomNoiseChain()
    fallback.add(mod_noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    chain = mod_noise.CustomNoiseChain()
    chain.add(mod_noise.ScheduledNoise(1.0, noise=inner, start_sigma=20.0, end_sigma=0.0, normalize=None, fallback_noise=fallback))
    return chain


def gen_cfg5_full():
    """cfg5 at 2 x 16 x 128 x 128 through the REAL reference: two SonarDPMPPSDE steps with momentum 0.95 (four noise calls).  The latent
    is re-created from its seed by the test; stored: every fourth pixel of the result and of one noise call, and fp64 plane sums of
    both (whole-tensor evidence in 300 KB)."""
    S, N, pn = ref.sonar, ref.noise, ref.powernoise
    shape = (2, 16, 128, 128)
    torch.manual_seed(71)
    x0 = torch.randn(shape) * 10.0
    sigmas = torch.tensor([10.0, 7.0, 4.0, 0.0])
    torch.manual_seed(72)
    ns = cfg5_full_chain(N, pn).make_noise_sampler(x0, sigmas[sigmas > 0].min(), sigmas.max(), seed=5, cpu=True, normalized=True)
    calls = []

    def spy(s, sn):
        out = ns(s, sn)
        calls.append(out.clone())
        return out

    out = S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas[:3], {"seed": 5}, None, True, None, dict(momentum=0.95), 1.0, 1.0, spy)
    assert len(calls) == 4
    save("cfg5_full", sigmas=sigmas, out_sub=out[..., ::4, ::4].contiguous(), out_plane_sums=out.double().sum(dim=(-2, -1)),
         out_plane_sq=(out.double() ** 2).sum(dim=(-2, -1)), noise_sub=calls[1][..., ::4, ::4].contiguous(),
         noise_plane_sums=calls[1].double().sum(dim=(-2, -1)), noise_plane_sq=(calls[1].double() ** 2).sum(dim=(-2, -1)))


# ------------------------------------------------------------------------------------------------ guidance
GUIDANCE_CASES = {
    "linear": dict(guidance_type="LINEAR", factor=0.05, start_step=1, end_step=4),
    "euler": dict(guidance_type="EULER", factor=0.2, start_step=0, end_step=9999),
    "linear_inject": dict(guidance_type="LINEAR", factor=0.1, start_step=2, end_step=5, guidance_blend_mode="inject"),
}


def gen_guidance():
    S = ref.sonar
    cases = {}
    shape = (2, 4, 8, 8)
    torch.manual_seed(71)
    x0 = torch.randn(shape) * 14.6
    ref_latent = torch.randn(shape) * 0.7 + 0.3
    sigmas = torch.cat((torch.linspace(14.6, 0.03, 7), torch.zeros(1)))
    noise_bank = torch.randn(16, *shape)
    cases.update(x0=x0, ref_latent=ref_latent, sigmas=sigmas, noise_bank=noise_bank)

    def bank_sampler():
        it = iter(noise_bank)
        return lambda s, sn: next(it).clone()

    for name, kw in GUIDANCE_CASES.items():
        kw = dict(kw)
        blend = kw.pop("guidance_blend_mode", None)
        gcfg = S.GuidanceConfig(guidance_type=S.GuidanceType[kw.pop("guidance_type")], latent=ref_latent.clone(), **kw)
        params = {"guidance": gcfg, "momentum": 0.9}
        if blend:
            params["guidance_blend_mode"] = blend
        for kind in ("euler", "ancestral", "dpmpp"):
            trace = []
            cb = lambda d: trace.append(d["x"].clone())  # noqa: E731
            if kind == "euler":
                S.SonarEuler.sampler(fake_model, x0.clone(), sigmas, {"seed": 0}, cb, True, bank_sampler(), None, dict(params))
            elif kind == "ancestral":
                S.SonarEulerAncestral.sampler(fake_model, x0.clone(), sigmas, {"seed": 0}, cb, True, None, dict(params), 0.8, 1.1, bank_sampler())
            else:
                S.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas, {"seed": 0}, cb, True, None, dict(params), 0.9, 1.05, bank_sampler())
            cases[f"{kind}_{name}"] = torch.stack(trace)
    # the building blocks on their own
    cases["prepared_ref"] = S.SonarGuidanceMixin.prepare_ref_latent(ref_latent.clone())
    cases["shift"] = S.SonarGuidanceMixin.guidance_shift(x0, cases["prepared_ref"])
    cases["euler_step"] = S.SonarGuidanceMixin.guidance_euler(torch.tensor(7.0), torch.tensor(5.0), x0, x0 * 0.5, cases["prepared_ref"], 0.3)
    cases["linear_step"] = S.SonarGuidanceMixin.guidance_linear(x0, cases["prepared_ref"], 0.3)
    save("guidance", **cases)


# ------------------------------------------------------------------------------------------------ 5-D (video) latents
VIDEO_TYPES = ("gaussian", "perlin", "pyramid", "pyramid_area", "onef_pinkish", "green_test", "velvet")


def gen_video():
    """[B, C, F, H, W] latents: FramesToChannels generators fold frames into channels (py/noise_generation.py:182-209)."""
    cases = {}
    shape = (2, 4, 3, 16, 16)
    for name in VIDEO_TYPES:
        for normalized in (False, True):
            cases[f"{name}_{int(normalized)}"] = ref_noise(getattr(NT, name.upper()), shape, 81, normalized)
    save("video", **cases)


# ------------------------------------------------------------------------------------------------ wavelet (octave) noise
WAVELET_ADV = {"deep": dict(octaves=6, persistence=0.7, initial_amplitude=2.0, height_factor=1.5, width_factor=1.5, min_height=2, min_width=2),
               "reverse": dict(octaves=-3, persistence=0.5, octave_height_factor=0.25, octave_width_factor=0.5, update_blend=0.6),
               "modes": dict(octaves=3, octave_scale_mode="area", octave_rescale_mode="nearest-exact", post_octave_rescale_mode="bilinear")}


def gen_wavelet_noise():
    cases = {}
    shape = (2, 4, 48, 32)
    for normalized in (False, True):
        cases[f"preset_{int(normalized)}"] = ref_noise(NT.WAVELET, shape, 91, normalized)
    for name, kw in WAVELET_ADV.items():
        item = ref.noise.AdvancedWaveletNoise(1.0, custom_noise=None, normalize_noise=False, normalize=None, update_blend_function=torch.lerp,
                                              **({"octave_scale_mode": "adaptive_avg_pool2d", "octave_rescale_mode": "bilinear",
                                                  "post_octave_rescale_mode": "bilinear", "initial_amplitude": 1.0, "persistence": 0.5,
                                                  "octaves": 4, "octave_height_factor": 0.5, "octave_width_factor": 0.5, "height_factor": 2.0,
                                                  "width_factor": 2.0, "update_blend": 1.0} | kw))
        torch.manual_seed(91)
        cases["adv_" + name] = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=91, cpu=True, normalized=False)(
            torch.tensor(14.6), torch.tensor(10.0))
    # octaves drawn by another chain (larger than the latent when an octave is): custom_noise
    chain = ref.noise.CustomNoiseChain()
    chain.add(ref.noise.CustomNoiseItem(1.0, noise_type="uniform"))
    item = ref.noise.AdvancedWaveletNoise(1.0, custom_noise=chain, normalize_noise=True, normalize=None, update_blend_function=torch.lerp,
                                          octave_scale_mode="adaptive_avg_pool2d", octave_rescale_mode="bilinear", post_octave_rescale_mode="bilinear",
                                          initial_amplitude=1.0, persistence=0.5, octaves=3, octave_height_factor=0.5, octave_width_factor=0.5,
                                          height_factor=2.0, width_factor=2.0, update_blend=1.0)
    torch.manual_seed(92)
    cases["adv_custom"] = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=92, cpu=True, normalized=True)(torch.tensor(14.6), torch.tensor(10.0))
    save("wavelet_noise", **cases)


# ------------------------------------------------------------------------------------------------ guided noise (8f rank 1)
def gen_guided_noise():
    cases = {}
    shape = (2, 4, 16, 16)
    torch.manual_seed(95)
    x = torch.randn(shape) * 3.0
    latent = torch.randn(shape) * 0.8 + 0.2
    cases["x"], cases["latent"] = x, latent
    S = ref.sonar
    ref_lat = ref.utils.scale_noise(S.SonarGuidanceMixin.prepare_ref_latent(latent.clone()), normalized=True)
    cases["ref_latent"] = ref_lat
    chain = ref.noise.CustomNoiseChain()
    chain.add(ref.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    for method in ("linear", "euler"):
        for with_noise in (True, False):
            item = ref.noise.GuidedNoise(1.0, guidance_factor=0.4, ref_latent=ref_lat, method=method, normalize_noise=None,
                                         normalize_result=None, noise=chain if with_noise else None)
            torch.manual_seed(96)
            ns = item.make_noise_sampler(x.clone(), 0.03, 14.6, seed=96, cpu=True, normalized=True)
            cases[f"{method}_{int(with_noise)}"] = ns(torch.tensor(9.0), torch.tensor(6.0))
    save("guided_noise", **cases)


def gen_modulated():
    """ModulatedNoise (py/noise.py:762-1019): intensity and frequency modes x the three modulation_dims, reference latent given or
    the sampler's own x; (s, sn) = (9, 6) -> sigma_up via get_ancestral_step(eta=1)."""
    cases = {}
    shape = (2, 4, 16, 16)
    torch.manual_seed(97)
    x = torch.randn(shape) * 2.0 + 0.3
    latent = torch.randn(shape) * torch.linspace(0.5, 2.0, 16)[None, None, :, None] + 0.1
    cases["x"], cases["latent"] = x, latent
    chain = ref.noise.CustomNoiseChain()
    chain.add(ref.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    for mtype in ("intensity", "frequency", "none"):
        for dims in (1, 2, 3):
            for with_ref in (True, False):
                item = ref.noise.ModulatedNoise(0.8, noise=chain, normalize_result=None if with_ref else False, normalize_noise=None, normalize_ref=True,
                                                modulation_type=mtype, modulation_strength=1.5 if with_ref else -0.6, modulation_dims=dims,
                                                ref_latent_opt=latent if with_ref else None)
                xin = x.clone()
                torch.manual_seed(98)
                ns = item.make_noise_sampler(xin, 0.03, 14.6, seed=98, cpu=True, normalized=True)
                cases[f"{mtype}_{dims}_{int(with_ref)}"] = ns(torch.tensor(9.0), torch.tensor(6.0))
                if not with_ref and mtype == "intensity" and dims == 3:
                    cases["x_after"] = xin  # normalize_ref acts on the sampler's x IN PLACE when no reference latent is given
    save("modulated", **cases)


# b1 / b4: LDS-resident power-of-two planes; g1: a general-size plane (SDXL 832 x 1216 px in miniature); o1: an odd plane (direct passes)
SIGNUM_SHAPES = (("b1", (1, 4, 32, 32)), ("b4", (4, 4, 16, 16)), ("g1", (1, 4, 26, 38)), ("o1", (1, 3, 9, 15)))


def gen_spectral_signum():
    """ModulatedNoise spectral_signum (py/noise.py:938-1015): fftn over the modulation dims, per-sample quantiles of |log amplitude|,
    soft clamp of the bins outside the 5 % / 95 % quantiles, inverse.  The reference expands the per-sample quantile vector [B] as
    [B, 1, 1] against [B, C, H, W], which only works for B = 1 (broadcast) or B = C (the vector then runs along the CHANNEL axis)."""
    cases = {}
    chain = ref.noise.CustomNoiseChain()
    chain.add(ref.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    for tag, shape in SIGNUM_SHAPES:
        for dims in (1, 2, 3):
            for strength in (2.0, -0.7):
                item = ref.noise.ModulatedNoise(0.9, noise=chain, normalize_result=None, normalize_noise=None, normalize_ref=False,
                                                modulation_type="spectral_signum", modulation_strength=strength, modulation_dims=dims)
                torch.manual_seed(99)
                ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=99, cpu=True, normalized=True)
                cases[f"{tag}_{dims}_{strength}"] = ns(torch.tensor(9.0), torch.tensor(6.0))
    item = ref.noise.ModulatedNoise(1.0, noise=chain, normalize_result=None, normalize_noise=None, normalize_ref=False, modulation_type="spectral_signum")
    try:
        item.make_noise_sampler(torch.zeros(2, 4, 16, 16), 0.03, 14.6, seed=99, cpu=True, normalized=True)(torch.tensor(9.0), torch.tensor(6.0))
        raise SystemExit("expected an error")
    except RuntimeError as exc:
        cases["b2_error"] = np.array(str(exc)[:60])
    save("spectral_signum", **cases)


def gen_item_wrappers():
    """RandomNoise / RepeatedNoise / ChannelNoise (py/noise.py:681-760, 1022-1131): sequences of calls on gaussian / uniform items."""
    cases = {}
    N = ref.noise
    shape = (2, 4, 8, 8)

    def chain(*specs):
        c = N.CustomNoiseChain()
        for name, f in specs:
            c.add(N.CustomNoiseItem(f, noise_type=name))
        return c

    sig = (torch.tensor(9.0), torch.tensor(6.0))
    for mix in (1, 2):
        item = N.RandomNoise(0.7, noise=chain(("gaussian", 1.0), ("uniform", 0.5), ("perlin", 0.8)), mix_count=mix, normalize=None)
        torch.manual_seed(41)
        ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=41, cpu=True, normalized=True)
        cases[f"random_mix{mix}"] = torch.stack([ns(*sig) for _ in range(4)])
    for permute in ("enabled", "always", "disabled"):
        item = N.RepeatedNoise(0.9, noise=chain(("gaussian", 1.0)), repeat_length=2, max_recycle=3, normalize=None, permute=permute)
        torch.manual_seed(42)
        ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=42, cpu=True, normalized=True)
        cases[f"repeated_{permute}"] = torch.stack([ns(*sig) for _ in range(9)])
    for mode in ("wrap", "repeat", "zero"):
        item = N.ChannelNoise(1.1, noise=chain(("gaussian", 1.0), ("uniform", 0.5)), insufficient_channels_mode=mode, normalize=None)
        torch.manual_seed(43)
        ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=43, cpu=True, normalized=True)
        cases[f"channel_{mode}"] = torch.stack([ns(*sig) for _ in range(2)])
    for tag, kw in (("cos_dim-1", dict(mode="cos", dim=-1, flatten=False)), ("sin_copysign_dim1", dict(mode="sin_copysign", dim=1, flatten=False)),
                    ("cos_flat2", dict(mode="cos", dim=2, flatten=True))):
        item = N.RippleFilteredNoise(0.8, noise=chain(("gaussian", 1.0)), offset=0.3, roll=1.5, amplitude_high=0.25, amplitude_low=1.6, period=3.0,
                                     normalize_noise=False, normalize=None, **kw)
        torch.manual_seed(44)
        ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=44, cpu=True, normalized=True)
        cases[f"ripple_{tag}"] = torch.stack([ns(*sig) for _ in range(3)])
    for tag, kw in (("dim1_chunk1", dict(dim=1, shrink_dim=False, chunk_size=1)), ("dim2_chunk4", dict(dim=2, shrink_dim=False, chunk_size=4)),
                    ("dim1_shrink", dict(dim=1, shrink_dim=True, chunk_size=1))):
        item = N.PerDimNoise(0.6, noise=chain(("gaussian", 1.0)), offset=0, normalize_noise=False, normalize=None, **kw)
        torch.manual_seed(45)
        ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=45, cpu=True, normalized=True)
        cases[f"perdim_{tag}"] = torch.stack([ns(*sig) for _ in range(2)])
    save("item_wrappers", **cases)


def gen_power_wrapped():
    """RandomNoise / ChannelNoise over PowerNoiseItems with factor 1 and the identity channel mixer (py/noise.py:1022-1131 calling
    py/nodes/powernoise.py:338-408 with normalized=False, then scale_noise(normalized=True) on the result)."""
    cases = {}
    N = ref.noise
    shape = (2, 4, 16, 16)
    sig = (torch.tensor(9.0), torch.tensor(6.0))

    def chain(*items):
        c = N.CustomNoiseChain()
        for it in items:
            c.add(it)
        return c

    item = N.RandomNoise(1.0, noise=chain(ref_power_item(), ref_power_item(alpha=2.0)), mix_count=1, normalize=None)
    torch.manual_seed(46)
    ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=46, cpu=True, normalized=True)
    cases["random_power"] = torch.stack([ns(*sig) for _ in range(4)])
    item = N.ChannelNoise(1.0, noise=chain(ref_power_item(), N.CustomNoiseItem(1.0, noise_type="gaussian")), insufficient_channels_mode="wrap",
                          normalize=None)
    torch.manual_seed(47)
    ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=47, cpu=True, normalized=True)
    cases["channel_power"] = torch.stack([ns(*sig) for _ in range(2)])
    save("power_wrapped", **cases)


# ------------------------------------------------------------------------------------------------ round 2: remaining section-8 rows
PYRAMID_VARIANTS = ("highres_pyramid", "highres_pyramid_area", "pyramid_old", "pyramid_old_area", "pyramid_mix", "pyramid_mix_area")


def gen_pyramid_variants():
    """HighresPyramid / PyramidOld / pyramid_mix presets (py/noise_generation.py:517-606, py/noise.py:2345-2418), replay mode."""
    cases = {}
    shape = (2, 4, 16, 16)
    for name in PYRAMID_VARIANTS:
        for normalized in (False, True):
            cases[f"{name}_{int(normalized)}"] = ref_noise(getattr(NT, name.upper()), shape, 61, normalized)
    cases["video_highres"] = ref_noise(NT.HIGHRES_PYRAMID, (1, 4, 2, 8, 8), 62, True)
    save("pyramid_variants", **cases)


def gen_ffilter():
    """FreeU-Extreme's ffilter (py/nodes/freeu_extreme.py:10-29): irfft2(rfft2(x) * normalised power filter), with its filter cache."""
    fx = importlib.import_module("sonar_ref.nodes.freeu_extreme")
    PF = ref.powernoise.PowerFilter
    cases = {}
    g = torch.Generator().manual_seed(71)
    specs = {"a": ((2, 8, 32, 32), dict(alpha=1.0, max_freq=0.7071), 1.0), "b": ((1, 6, 16, 64), dict(alpha=-0.5, min_freq=0.05, max_freq=0.5), 0.7),
             "c": ((1, 3, 40, 56), dict(alpha=2.0, max_freq=0.7071, stretch=1.5, rotate=20.0), 1.0)}
    cache = {}
    for tag, (shape, fkw, nf) in specs.items():
        x = torch.randn(shape, generator=g)
        cases[f"{tag}_x"] = x
        cases[f"{tag}_out"] = fx.ffilter(x, PF(**fkw), normalization_factor=nf, cfg_idx=tag, filter_cache=cache)
    # same cache key, another filter: the cached one wins (:13-16)
    cases["a_cached_out"] = fx.ffilter(cases["a_x"], PF(alpha=3.0), cfg_idx="a", filter_cache=cache)
    assert torch.equal(cases["a_cached_out"], cases["a_out"])
    # half precision in, half precision out (:22,29)
    cases["a_half_out"] = fx.ffilter(cases["a_x"].half(), PF(alpha=1.0, max_freq=0.7071), cfg_idx=0, filter_cache={}).float()
    for kw in (dict(), dict(cfg_idx=1), dict(filter_cache={})):  # no cache key: the reference never binds filter_rfft
        try:
            fx.ffilter(cases["a_x"], PF(alpha=1.0), **kw)
            raise SystemExit("expected an error")
        except UnboundLocalError:
            pass
    save("ffilter", **cases)


def gen_cfg_exact():
    """BASELINE cfg1 exactly as SURVEY.md 8d states it, cfg3 in replay mode at SDXL size, and the RAND history initialisation."""
    S = ref.sonar
    cases = {}
    torch.manual_seed(3)
    x0 = torch.randn(1, 4, 64, 64) * 14.6
    sigmas = torch.cat((torch.linspace(14.6, 0.03, 20), torch.zeros(1)))
    trace = []
    out = S.SonarEuler.sampler(lambda x, sigma, **_k: x * 0.5, x0.clone(), sigmas, {"seed": 0}, lambda d: trace.append(d["x"].clone()), True, None, None,
                               dict(momentum=0.95))
    mine = orc.sonar_euler(lambda x, sigma, **_k: x * 0.5, x0.clone(), sigmas, orc.MomentumCfg(momentum=0.95))
    must_equal(out, mine, "cfg1")
    cases["cfg1_x0"], cases["cfg1_sigmas"], cases["cfg1_out"], cases["cfg1_trace"] = x0, sigmas, out, torch.stack(trace)
    # gaussian noise on the same latent, as the config names it (not consumed by the non-ancestral sampler)
    cases["cfg1_noise"] = ref_noise(NT.GAUSSIAN, (1, 4, 64, 64), 3, True)
    shape = (2, 4, 128, 128)
    for name in ("perlin", "pyramid"):
        cases[f"cfg3_{name}"] = ref_noise(getattr(NT, name.upper()), shape, 0, True)
    # init = RAND draws its history from the global generator inside the first momentum step, BEFORE that step's noise (py/sonar.py:169-206)
    shape = (2, 4, 8, 8)
    torch.manual_seed(5)
    x1 = torch.randn(shape) * 14.6
    sig7 = torch.cat((torch.linspace(14.6, 0.03, 7), torch.zeros(1)))
    cases["rand_x0"], cases["rand_sigmas"] = x1, sig7
    for kind in ("euler", "ancestral", "dpmpp"):
        trace = []
        cb = lambda d: trace.append(d["x"].clone())  # noqa: E731
        torch.manual_seed(17)
        x = torch.zeros(shape)
        ns = ref.noise.get_noise_sampler(NT.GAUSSIAN, x, 0.03, 14.6, seed=17, cpu=True, factor=1.0, normalized=True)
        kw = dict(init="RAND", momentum=0.9)
        if kind == "euler":
            S.SonarEuler.sampler(fake_model, x1.clone(), sig7, {"seed": 0}, cb, True, None, None, dict(kw))
        elif kind == "ancestral":
            S.SonarEulerAncestral.sampler(fake_model, x1.clone(), sig7, {"seed": 0}, cb, True, None, dict(kw), 0.8, 1.1, ns)
        else:
            S.SonarDPMPPSDE.sampler(fake_model, x1.clone(), sig7, {"seed": 0}, cb, True, None, dict(kw), 0.9, 1.05, ns)
        cases[f"rand_{kind}"] = torch.stack(trace)
    save("cfg_exact", **cases)


def gen_resample_modes():
    """scale_samples (py/utils.py:58-67 -> F.interpolate) in every mode the reference lists except bislerp, enlarging, shrinking and
    at non-integer ratios; the pyramid generators with bicubic / nearest levels (py/noise_generation.py:608-680,517-606); GuidedNoise
    with a reference latent of another size (bicubic, align_corners=True, py/noise.py:581-588)."""
    cases = {}
    g = torch.Generator().manual_seed(131)
    src = {"a": torch.randn(2, 3, 8, 12, generator=g), "b": torch.randn(1, 4, 33, 17, generator=g), "c": torch.randn(1, 2, 5, 5, generator=g)}
    sizes = {"a": [(16, 24), (13, 31), (4, 6), (8, 12)], "b": [(64, 64), (16, 9), (40, 17)], "c": [(1, 1), (5, 11)]}
    for tag, x in src.items():
        cases[f"src_{tag}"] = x
        for (h, w) in sizes[tag]:
            for mode in ("bilinear", "nearest-exact", "nearest", "area", "bicubic", "adaptive_avg_pool2d"):
                cases[f"scale_{tag}_{h}x{w}_{mode}"] = ref.utils.scale_samples(x, w, h, mode=mode)
    shape = (2, 4, 16, 16)
    for mode in ("bicubic", "nearest"):
        cases[f"pyramid_{mode}"] = ref_noise(NT.PYRAMID, shape, 63, True, upscale_mode=mode)
        cases[f"pyramid_old_{mode}"] = ref_noise(NT.PYRAMID_OLD, shape, 64, False, upscale_mode=mode)
        cases[f"highres_pyramid_{mode}"] = ref_noise(NT.HIGHRES_PYRAMID, shape, 65, True, upscale_mode=mode)
    torch.manual_seed(132)
    x = torch.randn(shape) * 3.0
    small = ref.utils.scale_noise(torch.randn(2, 4, 6, 10) * 0.8 + 0.2, normalized=True)
    cases["guided_x"], cases["guided_ref"] = x, small
    for method in ("linear", "euler"):
        item = ref.noise.GuidedNoise(1.0, guidance_factor=0.4, ref_latent=small, method=method, normalize_noise=None, normalize_result=None, noise=None)
        ns = item.make_noise_sampler(x.clone(), 0.03, 14.6, seed=96, cpu=True, normalized=True)
        cases[f"guided_{method}"] = ns(torch.tensor(9.0), torch.tensor(6.0))
    # normalize_to_scale (py/utils.py:450-469): default dims, the last two, everything; a constant group (hi == lo -> eps only)
    t = torch.randn(3, 4, 9, 7, generator=g) * 2.0 + 0.5
    t[1, 2] = 0.75
    cases["nts_in"] = t
    cases["nts_default"] = ref.utils.normalize_to_scale(t.clone(), 0.0, 1.0)
    cases["nts_hw"] = ref.utils.normalize_to_scale(t.clone(), -1.5, 2.0, dim=(-2, -1))
    cases["nts_all"] = ref.utils.normalize_to_scale(t.clone(), 0.25, 0.5, dim=(-4, -3, -2, -1), eps=1e-3)
    cases["nts_inexact"] = ref.utils.normalize_to_scale(t.clone(), 0.1, 0.3)  # targets that are not fp32 numbers: span = fp32(0.3 - 0.1)
    save("resample_modes", **cases)
    # reductions over dimensions that are not the trailing ones (py/utils.py:97-99,452-470, py/sonar.py:372-377)
    nt = {"t": t}
    nt["nts_c"] = ref.utils.normalize_to_scale(t.clone(), -1.0, 1.0, dim=(1,))
    nt["nts_bh"] = ref.utils.normalize_to_scale(t.clone(), 0.0, 2.0, dim=(0, 2))
    tn = torch.randn(3, 4, 9, 7, generator=g) * 1.7 + 0.3
    nt["sn_in"] = tn
    nt["sn_c"] = ref.utils.scale_noise(tn.clone(), 0.8, normalize_dims=(1,))
    nt["sn_bw"] = ref.utils.scale_noise(tn.clone(), 1.25, normalize_dims=(0, 3))
    rl = torch.randn(3, 4, 9, 7, generator=g)
    nt["gs_ref"] = rl
    nt["gs_c"] = ref.sonar.SonarGuidanceMixin.guidance_shift(tn.clone(), rl.clone(), dim=(1,))
    nt["gs_bhw"] = ref.sonar.SonarGuidanceMixin.guidance_shift(tn.clone(), rl.clone(), dim=(0, 2, 3))
    save("nontrailing", **nt)


# ------------------------------------------------------------------------------------------------ every registry type on odd shapes
SWEEP_SHAPES = [(1, 3, 10, 14), (2, 4, 12, 20), (1, 16, 8, 8), (3, 1, 18, 6), (2, 4, 9, 7), (1, 4, 2, 12, 8), (2, 2, 16, 16), (1, 5, 24, 4)]
SWEEP_SKIP = {"BROWNIAN",  # torchsde, un-vendored
              "COLLATZ", "DISTRO", "VORONOI_FUZZ", "VORONOI_MIX",  # outside the path
              "PYRAMID_BISLERP", "HIGHRES_PYRAMID_BISLERP", "PYRAMID_MIX_BISLERP", "PYRAMID_OLD_BISLERP"}  # comfy.utils.bislerp: not in the image


def gen_shape_sweep():
    """Every NoiseType of the registry (py/noise.py:2244-2457) that the path covers, in replay mode on shapes the other fixtures do not
    have: channel counts 1 / 3 / 5 / 16, H != W, odd sizes, a 5-D video latent.  What the reference RAISES on a shape is a result too
    (its type is recorded and the product must raise the same)."""
    import json
    import random

    rnd = random.Random(20260)
    cases, meta = {}, {}
    for nt in NT:
        if nt.name in SWEEP_SKIP:
            continue
        for k, shape in enumerate(rnd.sample(SWEEP_SHAPES, 3)):
            seed, normalized = 300 + 7 * k + len(nt.name), bool(k % 2)
            key = f"{nt.name.lower()}__{k}"
            try:
                out = ref_noise(nt, shape, seed, normalized)
                cases[key] = out
                meta[key] = dict(type=nt.name.lower(), shape=list(shape), seed=seed, normalized=normalized, error=None)
            except Exception as exc:  # noqa: BLE001 -- the reference's refusal is the expected behaviour
                meta[key] = dict(type=nt.name.lower(), shape=list(shape), seed=seed, normalized=normalized, error=type(exc).__name__,
                                 message=str(exc)[:200])
    ok = sum(1 for m in meta.values() if m["error"] is None)
    print(f"shape sweep: {ok} outputs, {len(meta) - ok} refusals")
    save("shape_sweep", meta_json=json.dumps(meta), **cases)


TINY_SHAPES = [(1, 1, 1, 1), (1, 4, 2, 2), (2, 3, 1, 5), (2, 4, 3, 3), (1, 4, 1, 2, 2)]


def gen_tiny_sweep():
    """The same registry types on degenerate latents: single pixels, 1-pixel rows, 2 x 2 and 3 x 3 planes.  Outputs where the reference
    produces one, its error type where it refuses.  (Empty batches are not part of the path: ComfyUI never samples one.)"""
    import json

    cases, meta = {}, {}
    for nt in NT:
        if nt.name in SWEEP_SKIP:
            continue
        for k, shape in enumerate(TINY_SHAPES):
            key = f"{nt.name.lower()}__{k}"
            seed, normalized = 500 + k, bool(k % 2)
            try:
                out = ref_noise(nt, shape, seed, normalized)
                cases[key] = out
                meta[key] = dict(type=nt.name.lower(), shape=list(shape), seed=seed, normalized=normalized, error=None)
            except Exception as exc:  # noqa: BLE001
                meta[key] = dict(type=nt.name.lower(), shape=list(shape), seed=seed, normalized=normalized, error=type(exc).__name__,
                                 message=str(exc)[:160])
    ok = sum(1 for m in meta.values() if m["error"] is None)
    print(f"tiny sweep: {ok} outputs, {len(meta) - ok} refusals")
    save("tiny_sweep", meta_json=json.dumps(meta), **cases)


def gen_wrapper_sweep():
    """Noise-item wrappers (py/noise.py: Random / Repeated / Channel / RippleFiltered / PerDim / Scheduled / Blended) over inner chains,
    on odd shapes and a 5-D video latent, several calls each: tests/golden/sweep_cases.py holds the specs both sides build from."""
    from tests.golden import sweep_cases as sc

    cases = {}
    for name in sc.WRAPPERS:
        item, shape, seed, calls = sc.build(ref.noise, ref.utils, name)
        torch.manual_seed(seed)
        ns = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=seed, cpu=True, normalized=True)
        cases[name] = torch.stack([ns(torch.tensor(sc.SIGMAS[k % len(sc.SIGMAS)][0]), torch.tensor(sc.SIGMAS[k % len(sc.SIGMAS)][1])) for k in range(calls)])
    save("wrapper_sweep", **cases)


def gen_sampler_sweep():
    """The three Sonar samplers end to end (py/sonar.py:452-820) with registry noise in replay mode on odd shapes, 16 / 5 / 1 channels
    and a 5-D video latent: per-step x of the reference (tests/golden/sweep_cases.py SAMPLERS)."""
    from tests.golden import sweep_cases as sc

    cases = {}
    for name in sc.SAMPLERS:
        cases[name] = torch.stack(sc.run_sampler(ref.sonar, ref.noise, name, "cpu"))
    save("sampler_sweep", **cases)


def gen_advanced_sweep():
    """ModulatedNoise, GuidedNoise, AdvancedWaveletNoise, PowerFilterNoiseItem and PowerNoiseItem (py/noise.py, py/nodes/powernoise.py) on odd
    planes and non-integer resize ratios: tests/golden/sweep_cases.py ADVANCED."""
    from tests.golden import sweep_cases as sc

    pn = importlib.import_module("sonar_ref.nodes.powernoise")
    cases = {}
    for name in sc.ADVANCED:
        cases[name] = torch.stack(sc.run_advanced(ref.noise, pn, ref.utils, name, "cpu"))
    save("advanced_sweep", **cases)


def gen_node_sweep():
    """Every node of the registry that returns a noise chain, run with the defaults of its own sockets (node_abi.json) over a gaussian base
    chain and sampled on an odd latent: tests/golden/sweep_cases.py run_node.  Nodes whose construction or sampling the reference refuses
    with these inputs are recorded with the exception type."""
    import json
    from tests.golden import sweep_cases as sc

    abi = json.load(open(os.path.join(OUT, "node_abi.json")))
    M = ref.nodes.NODE_CLASS_MAPPINGS
    cases, meta = {}, {}
    for key, entry in abi.items():
        if not entry["returns"] or entry["returns"][0] != "SONAR_CUSTOM_NOISE" or key in sc.NODE_SKIP:
            continue
        try:
            outs = sc.run_node(M, abi, key, "cpu")
        except Exception as exc:  # noqa: BLE001
            meta[key] = dict(error=type(exc).__name__, message=str(exc)[:160])
            continue
        if outs is None:
            continue
        cases[key.replace(" ", "_")] = torch.stack(outs)
        meta[key] = dict(error=None)
    print("node sweep:", sum(1 for m in meta.values() if m["error"] is None), "chains,", sum(1 for m in meta.values() if m["error"]), "refusals")
    save("node_sweep", meta_json=json.dumps(meta), **cases)


if __name__ == "__main__" and "--only" in sys.argv:
    globals()["gen_" + sys.argv[sys.argv.index("--only") + 1]]()
    sys.exit(0)

if __name__ == "__main__" and "--nodes-only" not in sys.argv:
    gen_scale_noise()
    gen_basic_types()
    gen_perlin()
    gen_pyramid()
    gen_power_filter()
    gen_power_noise()
    gen_composition()
    gen_momentum()
    gen_powerlaw()
    gen_latent_ops()
    gen_spectral()
    gen_cfg5()
    gen_cfg5_full()
    gen_guidance()
    gen_video()
    gen_wavelet_noise()
    gen_guided_noise()
    gen_modulated()
    gen_spectral_signum()
    gen_item_wrappers()
    gen_power_wrapped()
    gen_pyramid_variants()
    gen_ffilter()
    gen_cfg_exact()
    gen_resample_modes()
    gen_shape_sweep()
    gen_tiny_sweep()
    gen_wrapper_sweep()
    gen_sampler_sweep()
    gen_advanced_sweep()
    gen_node_sweep()
    globals()["gen_node_abi"]()
    gen_entry_nodes()
    print("golden vectors written to", OUT)

if __name__ == "__main__" and "--nodes-only" in sys.argv:
    gen_node_abi()
    gen_entry_nodes()
