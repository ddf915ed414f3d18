"""Case tables shared by tests/golden/make_golden.py (which runs them through the REAL reference) and the tests
(which run them through the oracle on CPU and the HIP path on the GPU).  Data only: inputs of the reference's
wavelet rows (W / WC / WF), no expected values -- those live in the .npz / .json fixtures.
"""
from __future__ import annotations

import math

import numpy as np
import torch

# ------------------------------------------------------------------------------------------------ model-sampling stand-in
class DiscreteSampling:
    """What ``WCFGPercentages.build`` asks of ``model.model_sampling`` (py/wavelet_cfg.py:139-153): ``sigma_min``,
    ``sigma_max`` and ``timestep(sigma)``.  Restates the published k-diffusion / ComfyUI discrete schedule (scaled-linear
    betas 0.00085..0.012 over 1000 steps, timestep = index of the nearest log-sigma).  Host-side interface object: the
    same instance type is handed to the reference (fixture generation) and to the product (tests)."""

    def __init__(self):
        betas = torch.linspace(0.00085**0.5, 0.012**0.5, 1000, dtype=torch.float64) ** 2
        alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.sigmas = (((1 - alphas_cumprod) / alphas_cumprod) ** 0.5).float()
        self.log_sigmas = self.sigmas.log()

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma):
        dists = sigma.log().reshape(-1)[None, :] - self.log_sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)


class FakeModel:
    def __init__(self):
        self.model_sampling = DiscreteSampling()


def karras_sigmas(n=12, sigma_min=0.0292, sigma_max=14.6146, rho=7.0):
    ramp = torch.linspace(0, 1, n)
    s = (sigma_max ** (1 / rho) + ramp * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([s, s.new_zeros(1)]).float()


# ------------------------------------------------------------------------------------------------ (a) scaling / blend
# (source tag in dwt.npz, yl_scale, yh_scales)
SCALING_CASES = {
    "scalar": ("db4_sym_l3_odd", 2.0, 3.0),
    "list": ("db4_sym_l3_odd", 0.5, [1.5, 0.5, 0.25]),
    "nested": ("db4_sym_l3_odd", 1.0, [[1.0, 2.0, 3.0], [0.5], 0.25]),
    "nested_long_short": ("db4_sym_l3_odd", 1.0, [[1.0, 2.0, 3.0, 4.0], [0.5, 2.0]]),
    "fill_pad": ("db4_sym_l5_64", 0.0, [1.5, [2.0, 0.5], "fill", 0.25]),
    "fill_drop": ("db4_sym_l3_odd", 1.0, [1.5, 0.5, "fill", 0.25]),
    "fill_last": ("db4_sym_l5_64", 1.0, [2.0, "fill"]),
    "too_long": ("db4_sym_l3_odd", 1.0, [1, 2, 3, 4, 5]),
    "none": ("db4_sym_l3_odd", 3.0, None),
    "ints": ("db4_sym_l3_odd", 2, 2),
    "oned_scalar": ("d1_db4_sym_l5_1024", 2.0, 3.0),
    "oned_list": ("d1_db4_sym_l5_1024", 0.5, [1.5, [2.0, 0.5], "fill"]),
    "oned_short": ("d1_sym5_reflect_l2", 1.0, [0.25]),
}
SCALING_ERRORS = {
    "fill_first": ("db4_sym_l3_odd", ["fill", 1.0]),
    "fill_twice": ("db4_sym_l3_odd", [1.0, "fill", "fill"]),
    "fill_only": ("db4_sym_l3_odd", ["fill"]),
}
# (tag, yl blend name, yh blend name or None, yl_factor, yh_factor or None)
BLEND_CASES = {
    "lerp_shared": ("db4_sym_l3_odd", "lerp", None, 0.3, None),
    "lerp_split": ("db4_sym_l3_odd", "lerp", None, 0.0, 1.0),
    "lerp_far": ("db4_sym_l3_odd", "lerp", None, 0.75, 0.5),
    "inject": ("db4_sym_l3_odd", "inject", None, 0.5, 2.0),
    "mixed": ("db4_sym_l3_odd", "lerp", "inject", 0.25, 0.8),
    "subtract": ("d1_sym5_reflect_l2", "subtract_b", None, 1.0, None),
}


def dwt_case(G, tag):
    """(meta, x, yl, [yh...]) of a PyWavelets case in dwt.npz as fp64 torch tensors."""
    wave, mode, level = (str(v) for v in G[f"{tag}__meta"])
    level = int(level)
    return (wave, mode, level), torch.from_numpy(G[f"{tag}__x"]), torch.from_numpy(G[f"{tag}__yl"]), [
        torch.from_numpy(G[f"{tag}__yh{j}"]) for j in range(level)]


def second_coeffs(yl, yh, seed=77):
    """A second coefficient tuple of the same shapes (for wavelet_blend): seeded Gaussian values."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(yl.shape, generator=g, dtype=yl.dtype), [torch.randn(b.shape, generator=g, dtype=b.dtype) for b in yh]


# ------------------------------------------------------------------------------------------------ (b) host schedule logic
SCHEDULES = ("linear", "logarithmic", "log", "exponential", "exp", "half_cosine", "sine", "sin")
INTERP_GRID = [-0.5, 0.0, 1e-9, 0.1, 0.25, 1 / 3, 0.5, 0.75, 0.9, 1.0, 1.5]

# WCFGPercentages.build inputs: (start_sigma, end_sigma, sigma, sample_sigmas key)
SAMPLE_SIGMAS = {
    "none": None,
    "karras12": karras_sigmas(12),
    "karras12_2d": torch.stack([karras_sigmas(12), karras_sigmas(12) * 0.5]),
    "two": torch.tensor([14.6146, 0.0292, 0.0]),
    "one": torch.tensor([7.0, 0.0]),
}
PCT_CASES = [
    (math.inf, 0.0, 7.0, "none"), (math.inf, 0.0, 14.6146, "none"), (math.inf, 0.0, 0.0292, "none"), (math.inf, 0.0, 100.0, "none"),
    (10.0, 1.0, 5.0, "none"), (10.0, 1.0, 10.0, "none"), (10.0, 1.0, 1.0, "none"), (10.0, 1.0, 0.5, "none"), (10.0, 1.0, 12.0, "none"),
    (math.inf, 0.0, 7.0, "karras12"), (math.inf, 0.0, 14.6146, "karras12"), (10.0, 1.0, 5.0, "karras12"), (10.0, 1.0, 3.3, "karras12"),
    (10.0, 1.0, 1.0, "karras12"), (10.0, 1.0, 9.9, "karras12"), (6.0, 6.0, 6.0, "karras12"), (3.0, 0.0, 0.05, "karras12"),
    (math.inf, 0.0, 2.0, "karras12_2d"), (math.inf, 0.0, 3.0, "two"), (10.0, 1.0, 4.0, "two"), (math.inf, 0.0, 7.0, "one"),
    (20.0, 0.01, 14.0, "karras12"), (8.0, 2.0, float(karras_sigmas(12)[4]), "karras12"), (8.0, 2.0, float(karras_sigmas(12)[3]) - 0.2, "karras12"),
]
PCT_ERRORS = [
    (1.0, 5.0, 3.0, "none"),           # start < end
    (math.inf, 0.0, 3.0, "ascending"),  # non-descending sample sigmas
    (math.inf, 0.0, 3.0, "bad_ndim"),
]

# WCFGScheduledScale.get_b_scale: kwargs handed to WCFGScheduledScale.build
SCHED_SCALE_CASES = [
    {}, {"schedule": "sine"}, {"schedule": "log", "schedule_mode": "sampling"}, {"schedule": "exp", "schedule_mode": "sigmas"},
    {"schedule": "half_cosine", "schedule_mode": "enabled_sigmas", "reverse_schedule": True},
    {"schedule_mode": "step", "schedule_offset": 0.1, "schedule_multiplier": 1.5},
    {"schedule_mode": "enabled_sigma_range", "schedule_offset_after": -0.2, "schedule_multiplier_after": 2.0, "schedule_min": 0.1, "schedule_max": 0.9},
    {"schedule": "linear", "schedule_mode": "model_sampling", "reverse_schedule_after": True},
    {"schedule": "sin", "schedule_mode": "enabled_model_sampling", "schedule_min": -1.0, "schedule_max": 2.0, "schedule_multiplier": 0.5},
]

# WCFGScalesRange.build(...).get_scales(pcts, yh) -- "yh" only supplies the band / orientation counts
SCALES_RANGE_CASES = {
    "plain": dict(yl_scale=2.0, yh_scales=[1.0, 2.0]),
    "same_end": dict(yl_scale=2.0, yh_scales=3.0, scales_end=dict(yl_scale=2.0, yh_scales=3.0)),
    "lerp_range": dict(yl_scale=1.0, yh_scales=[1.0, [2.0, 3.0], "fill"], scales_end=dict(yl_scale=0.0, yh_scales=4.0), schedule="linear",
                       schedule_mode="sigmas"),
    "inject_range": dict(scales_start=dict(yl_scale=1.5, yh_scales=2.0), scales_end=dict(yl_scale=0.5, yh_scales=[1.0, 0.5, 0.25]),
                         blend_mode="inject", schedule="sine", schedule_mode="enabled_sigmas"),
    "subtract_range": dict(yl_scale=1.0, yh_scales=1.0, scales_end=dict(yh_scales=[3.0, "fill"]), blend_mode="subtract_b",
                           schedule_mode="step"),
    "end_only_yl": dict(yl_scale=1.0, yh_scales=2.0, scales_end=dict(yl_scale=3.0), schedule_mode="enabled_model_sampling"),
}
SCALES_RANGE_LEVELS = (5, 3)   # (bands, orientations)

# WCFGScheduledFloat(...).get_value and WCFGScheduledFloat.build
SCHED_FLOAT_CASES = [0.7, {"value_start": 0.2, "value_end": 1.2, "schedule": "sine", "schedule_mode": "sigmas"},
                     {"value_start": 2, "value_end": 0, "schedule_mode": "step"}, {"value_start": 0.5}]

# WCFGRules.build(**params).get_rule(sigma): params + probe sigmas
RULES_CASES = {
    "single_default": dict(params=dict(difference=dict(yl_scale=5.0, yh_scales=3.0)), sigmas=[0.0, 0.5, 14.6, 1e9]),
    "window": dict(params=dict(start_sigma=10.0, end_sigma=2.0, diff=dict(yl_scale=2.0)), sigmas=[10.0, 10.01, 2.0, 1.99, 5.0]),
    "negative_start": dict(params=dict(start_sigma=-1.0, end_sigma=3.0), sigmas=[2.0, 3.0, 1e6]),
    "many": dict(params=dict(start_sigma=14.0, end_sigma=8.0, target_mode="noise", wave="haar", level=2, padding_mode="periodization",
                             rules=[dict(start_sigma=8.0, end_sigma=4.0, blend_strength=0.5, difference_blend_mode="lerp", high_precision_mode=False),
                                    dict(start_sigma=6.0, end_sigma=0.0, target_mode="NOISE_NORM", cond=dict(yl_scale=1.1), uncond=dict(yh_scales=0.9),
                                         final=dict(yl_scale=1.0, yh_scales=1.0, scales_end=dict(yh_scales=2.0)), fallback_existing=False,
                                         inv_wave="db2", inv_padding_mode="zero")]),
                 sigmas=[14.0, 8.0, 7.99, 5.0, 4.0, 3.9, 0.0, 15.0]),
}

# ------------------------------------------------------------------------------------------------ (c) WaveletCFG end to end
PLACEHOLDER = dict(difference=dict(yl_scale=5.0, yh_scales=3.0))  # the node's placeholder YAML (py/nodes/misc.py:664ff)
# name -> dict(shape, sigma, rule params, [sample_sigmas key], [cond_scale], [existing: use an existing cfg function], [seed])
WCFG_CASES = {
    "placeholder": dict(shape=(2, 4, 64, 64), sigma=7.0, params=PLACEHOLDER),
    "placeholder_f32": dict(shape=(2, 4, 64, 64), sigma=7.0, params=dict(PLACEHOLDER, high_precision_mode=False)),
    "placeholder_128": dict(shape=(1, 4, 128, 128), sigma=3.0, params=PLACEHOLDER),
    "identity_scales": dict(shape=(1, 4, 32, 32), sigma=7.0, params=dict(difference=dict(yl_scale=1.0, yh_scales=1.0), difference_blend_strength=5.0)),
    "haar_per": dict(shape=(2, 4, 48, 40), sigma=7.0, params=dict(PLACEHOLDER, wave="haar", level=3, padding_mode="periodization")),
    "odd_sizes": dict(shape=(1, 3, 37, 50), sigma=7.0, params=dict(difference=dict(yl_scale=2.0, yh_scales=[1.5, [2.0, 0.5], "fill", 0.25]), level=3)),
    "lerp_diff": dict(shape=(2, 4, 32, 32), sigma=7.0, params=dict(difference=dict(yl_scale=2.0, yh_scales=[1.5, [2.0, 0.5], "fill", 0.25]),
                                                                   difference_blend_mode="lerp", difference_blend_strength=0.8, level=4)),
    "lerp_diff_small_t": dict(shape=(2, 4, 32, 32), sigma=7.0, params=dict(PLACEHOLDER, difference_blend_mode="lerp", difference_blend_strength=0.3,
                                                                           level=2)),
    "subtract_diff": dict(shape=(1, 4, 32, 32), sigma=7.0, params=dict(PLACEHOLDER, difference_blend_mode="subtract_b", level=2)),
    "all_scales": dict(shape=(2, 4, 40, 56), sigma=4.0, params=dict(cond=dict(yl_scale=1.1, yh_scales=[1.0, 1.2]), uncond=dict(yl_scale=0.9, yh_scales=0.95),
                                                                    diff=dict(yl_scale=3.0, yh_scales=[[2.0, 2.5, 3.0], 1.5, "fill"]),
                                                                    final=dict(yl_scale=1.05, yh_scales=[0.9, 1.1, "fill"]), level=3, wave="sym5",
                                                                    padding_mode="reflect")),
    "all_scales_f32": dict(shape=(2, 4, 40, 56), sigma=4.0, params=dict(cond=dict(yl_scale=1.1, yh_scales=[1.0, 1.2]), uncond=dict(yl_scale=0.9, yh_scales=0.95),
                                                                        diff=dict(yl_scale=3.0, yh_scales=[[2.0, 2.5, 3.0], 1.5, "fill"]),
                                                                        final=dict(yl_scale=1.05, yh_scales=[0.9, 1.1, "fill"]), level=3, wave="sym5",
                                                                        padding_mode="reflect", high_precision_mode=False)),
    "target_noise": dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, target_mode="noise", level=3)),
    "target_noise_norm": dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, target_mode="noise_norm", level=3, wave="db2")),
    "target_noise_norm_blend": dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, target_mode="noise_norm", level=2, blend_strength=0.6)),
    "blend_half": dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, level=3, blend_strength=0.5)),
    "blend_inject": dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, level=3, blend_strength=0.25, blend_mode="inject")),
    "blend_zero": dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, level=3, blend_strength=0.0)),
    "blend_noise_target": dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, level=3, blend_strength=0.4, target_mode="noise")),
    "outside_window": dict(shape=(2, 4, 32, 32), sigma=3.0, params=dict(PLACEHOLDER, start_sigma=14.0, end_sigma=5.0)),
    "outside_window_existing": dict(shape=(2, 4, 32, 32), sigma=3.0, params=dict(PLACEHOLDER, start_sigma=14.0, end_sigma=5.0), existing=True),
    "existing_ignored": dict(shape=(2, 4, 32, 32), sigma=3.0, params=dict(PLACEHOLDER, start_sigma=14.0, end_sigma=5.0, fallback_existing=False),
                             existing=True),
    "second_rule": dict(shape=(2, 4, 32, 32), sigma=3.0, params=dict(PLACEHOLDER, start_sigma=14.0, end_sigma=5.0,
                                                                      rules=[dict(start_sigma=5.0, end_sigma=0.0, difference=dict(yl_scale=0.5, yh_scales=2.0),
                                                                                  wave="haar", level=2, padding_mode="periodization")])),
    "scheduled_scales": dict(shape=(2, 4, 32, 32), sigma=3.3, sample_sigmas="karras12",
                             params=dict(start_sigma=10.0, end_sigma=1.0, level=3,
                                         difference=dict(yl_scale=1.0, yh_scales=[1.0, [2.0, 3.0], "fill"], scales_end=dict(yl_scale=4.0, yh_scales=0.5),
                                                         schedule="sine", schedule_mode="enabled_sigmas"),
                                         final=dict(yl_scale=1.0, yh_scales=1.0, scales_end=dict(yh_scales=[1.5, "fill"]), blend_mode="inject",
                                                    schedule_mode="step"))),
    "per_batch_sigma": dict(shape=(2, 4, 32, 32), sigma=[6.0, 2.0], params=dict(PLACEHOLDER, target_mode="noise_norm", level=2)),
    "two_levels_inv_wave": dict(shape=(1, 4, 32, 32), sigma=7.0, params=dict(PLACEHOLDER, wave="bior2.2", inv_wave="bior2.2", level=2,
                                                                             padding_mode="periodic", inv_padding_mode="periodic")),
    "video_5d": dict(shape=(1, 4, 3, 24, 32), sigma=7.0, params=dict(PLACEHOLDER, level=2)),
    "oned": dict(shape=(2, 4, 24, 20), sigma=7.0, params=dict(difference=dict(yl_scale=2.0, yh_scales=[3.0, 0.5, "fill"]), use_1d_dwt=True, level=3)),
    "oned_f32": dict(shape=(2, 4, 24, 20), sigma=7.0, params=dict(difference=dict(yl_scale=2.0, yh_scales=[3.0, 0.5, "fill"]), use_1d_dwt=True, level=3,
                                                                  high_precision_mode=False)),
}
# a periodised inverse after a symmetric forward transform (and the other way round) does NOT reconstruct: the reference's result
# carries the shift of the mismatched pair, so these rules must take the band-by-band path (no linearity shortcut)
WCFG_CASES["mixed_modes"] = dict(shape=(2, 4, 32, 32), sigma=7.0, params=dict(PLACEHOLDER, wave="db2", level=1, padding_mode="symmetric",
                                                                              inv_padding_mode="periodization"))
WCFG_CASES["mixed_modes_zero_f32"] = dict(shape=(1, 4, 32, 32), sigma=7.0, params=dict(PLACEHOLDER, wave="db4", level=1, padding_mode="zero",
                                                                                       inv_padding_mode="periodization", high_precision_mode=False))
WCFG_CASES["with_ops"] = dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, level=2, blend_strength=0.5), ops=True)
WCFG_CASES["with_ops_plain"] = dict(shape=(2, 4, 32, 32), sigma=5.0, params=dict(PLACEHOLDER, level=2), ops=True)
WCFG_ERRORS = {
    "three_d_needs_1d": dict(shape=(2, 4, 480), sigma=7.0, params=PLACEHOLDER),
    "two_d": dict(shape=(4, 480), sigma=7.0, params=PLACEHOLDER),
    "no_sample_sigmas": dict(shape=(1, 4, 16, 16), sigma=7.0, params=PLACEHOLDER, sample_sigmas=None),
    # odd latent sizes reconstruct one sample larger (17 x 23 -> 18 x 24); the plain path crops in process_output, but a blend with the
    # fallback CFG (py/wavelet_cfg.py:825-836) meets the uncropped reconstruction first and torch refuses to broadcast
    "odd_size_blend": dict(shape=(2, 4, 17, 23), sigma=7.0, params=dict(PLACEHOLDER, blend_strength=0.5)),
}


def wcfg_inputs(case: dict, name: str):
    """Seeded inputs of a WaveletCFG call (ComfyUI ``sampler_cfg_function`` args minus the model): cond / uncond are the
    two denoised predictions, ``args["cond"]`` / ``args["uncond"]`` the matching noise predictions ``x - denoised``."""
    seed = sum(name.encode()) + 1000
    g = torch.Generator().manual_seed(seed)
    shape = tuple(case["shape"])
    cond, uncond, x = (torch.randn(shape, generator=g) for _ in range(3))
    sig = case["sigma"]
    sigma = torch.tensor(sig if isinstance(sig, list) else [sig] * shape[0], dtype=torch.float32)
    return dict(input=x, cond_scale=case.get("cond_scale", 7.0), cond=x - cond, uncond=x - uncond, cond_denoised=cond, uncond_denoised=uncond, sigma=sigma)


class _ExtendedOp:
    """A latent operation that asks for the extended keyword set (py/wavelet_cfg.py:663-675)."""

    EXTENDED_LATENT_OPERATION = True

    def __call__(self, latent, sigma=None, cond=None, uncond=None, cond_scale=None, raw_args=None, **_kw):
        return latent + 0.01 * cond_scale * (cond - uncond) / sigma.reshape(-1, *((1,) * (latent.ndim - 1)))


def wcfg_ops() -> dict:
    """User-supplied hooks of the five ``operation_*`` sockets (plain torch callables: the hooks are not product code)."""
    return dict(operation_cond=lambda latent: latent * 1.25, operation_uncond=lambda latent: latent - 0.1, operation_fallback_cfg=_ExtendedOp(),
                operation_wavelet_cfg=lambda latent: latent * 0.9, operation_result=lambda latent: latent + 0.5)


def existing_cfg(args):
    """A stand-in for a previously installed ``sampler_cfg_function`` (rescaled CFG-like: distinguishable from plain CFG)."""
    x, scale = args["input"], args["cond_scale"]
    uncond, cond = args["uncond_denoised"], args["cond_denoised"]
    return x - (uncond + (cond - uncond) * (scale * 0.5))


# ------------------------------------------------------------------------------------------------ (d) wavelet-filtered noise
# WaveletFilteredNoiseGenerator keyword arguments (py/noise_generation.py:1927-1950)
WF_GEN_CASES = {
    "defaults": dict(shape=(2, 4, 40, 24), kw=dict()),
    "scaled": dict(shape=(2, 4, 40, 24), kw=dict(yl_scale=0.5, yh_scales=[1.5, [1.0, 0.5, 2.0], "fill"])),
    "db4_sym": dict(shape=(2, 4, 40, 24), kw=dict(wave="db4", mode="symmetric", level=2, yl_scale=0.0, yh_scales=1.25)),
    "two_step": dict(shape=(2, 4, 40, 24), kw=dict(wave="sym5", mode="reflect", level=2, yl_scale=1.5, yh_scales=[0.5, 2.0], two_step_inverse=True)),
    "odd": dict(shape=(1, 3, 37, 50), kw=dict(wave="db2", mode="symmetric", level=3, yh_scales=[2.0, "fill"])),
    "inv_wave": dict(shape=(1, 4, 32, 32), kw=dict(wave="bior2.2", level=2, mode="periodization", inv_wave="bior2.2", inv_mode="periodization", yl_scale=2.0)),
    "video": dict(shape=(1, 4, 3, 16, 24), kw=dict(level=2, yh_scales=[0.5, 1.5])),
    "oned": dict(shape=(2, 4, 16, 24), kw=dict(use_1d_dwt=True, wave="db2", level=2, yl_scale=0.5, yh_scales=[2.0, 1.5])),
    "oned_haar": dict(shape=(1, 4, 16, 16), kw=dict(use_1d_dwt=True, level=3, yh_scales=0.25)),
}
# WaveletFilteredNoise item: (yaml parameters as a dict, with a separate high chain?, normalize_noise, normalized, shape)
WF_ITEM_CASES = {
    "high_db2": dict(shape=(2, 4, 32, 32), high=True, normalize_noise=False, normalized=True,
                     yaml=dict(wave="db2", level=2, mode="periodization", yh_scales=[1.0, 0.5], preblend_yl_scale_high=2.0)),
    "high_blend": dict(shape=(2, 4, 32, 32), high=True, normalize_noise=True, normalized=False,
                       yaml=dict(wave="haar", level=3, yl_blend_high=0.25, yh_blend_high=0.6, preblend_yh_scales_low=[2.0, "fill"],
                                 preblend_yl_scale_low=0.5, preblend_yh_scales_high=[[1.0, 2.0, 3.0]], yl_scale=1.5)),
    "high_inject": dict(shape=(1, 4, 24, 40), high=True, normalize_noise=False, normalized=True,
                        yaml=dict(wave="db4", mode="symmetric", level=2, yl_blend_function="inject", yh_blend_function="lerp", yl_blend_high=0.5,
                                  yh_blend_high=0.9, two_step_inverse=True)),
    "single": dict(shape=(2, 4, 32, 32), high=False, normalize_noise=False, normalized=True, yaml=dict(wave="sym5", level=2, mode="reflect", yl_scale=0.25)),
}
WF_NODE_CASE = dict(shape=(4, 4, 64, 64), yaml="wave: bior2.2\nlevel: 3\n", seed=3)


def np_float(a):
    return np.asarray(a, dtype=np.float64)
