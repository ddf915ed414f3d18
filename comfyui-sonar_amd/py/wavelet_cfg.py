"""Wavelet-domain CFG on MI355X (API of the reference's ``py/wavelet_cfg.py``).

Device work per call (py/wavelet_cfg.py:750-791): DWT(cond), DWT(uncond) -> per-band
``result = blend(uncond*s_u, (cond*s_c - uncond*s_u)*s_d, strength)*s_f`` (ONE fused kernel per band,
``sonar_wcfg_band_*``) -> IDWT -> ``x - result`` with crop and cast (``sonar_wcfg_output_f32``).
fp64 internally when ``high_precision_mode`` (the reference default).  Rule parsing and the schedule
arithmetic are host scalars (YAML keys and defaults are API and are kept).
"""
from __future__ import annotations

import math
from enum import Enum, auto
from typing import Callable, NamedTuple, Optional, Sequence

import numpy as _np
import torch

from .. import hip_lib
from . import utils
from .wavelet_functions import Wavelet, expand_yh_scales, wavelet_scaling

try:
    from tqdm import tqdm

    _say = tqdm.write
except ImportError:  # pragma: no cover
    _say = print


def clamp_float(val: float, minval=0.0, maxval=1.0) -> float:
    return max(minval, min(val, maxval))


def filter_dict(d: dict, keep) -> dict:
    return {k: v for k, v in d.items() if k in keep}


def pretty_non_default(obj, *, defaults=None) -> str:
    parts = []
    for name in obj._fields:
        val = getattr(obj, name)
        if defaults is not None and val == getattr(defaults, name):
            continue
        parts.append(f"{name}={val.pretty_non_default()}" if hasattr(val, "pretty_non_default") else f"{name}={val!r}")
    return f"{obj.__class__.__name__}({', '.join(parts)})"


class WCFGSchedule(Enum):
    LINEAR = auto()
    LOGARITHMIC = auto()
    LOG = LOGARITHMIC
    EXPONENTIAL = auto()
    EXP = EXPONENTIAL
    HALF_COSINE = auto()
    SINE = auto()
    SIN = SINE

    def interp(self, val: float) -> float:
        """py/wavelet_cfg.py:43-57."""
        val = clamp_float(val)
        if self == WCFGSchedule.LINEAR:
            return val
        if self == WCFGSchedule.LOGARITHMIC:
            out = 0.0 if val == 0 else math.log(val) + 1.0
        elif self == WCFGSchedule.EXPONENTIAL:
            out = math.exp(val) - 1.0
        elif self == WCFGSchedule.HALF_COSINE:
            out = 1.0 - ((1.0 + math.cos(val * math.pi)) / 2)
        elif self == WCFGSchedule.SINE:
            out = math.sin(val * math.pi)
        else:
            raise ValueError("Bad interpolation schedule!?")
        return clamp_float(out)


class WCFGSchedMode(Enum):
    SAMPLING = auto()
    ENABLED_SAMPLING = auto()
    SIGMAS = auto()
    ENABLED_SIGMAS = auto()
    STEP = auto()
    ENABLED_STEPS = auto()
    MODEL_SAMPLING = SAMPLING
    ENABLED_MODEL_SAMPLING = ENABLED_SAMPLING
    SIGMA_RANGE = SIGMAS
    ENABLED_SIGMA_RANGE = ENABLED_SIGMAS


class WCFGTarget(Enum):
    DENOISED = auto()
    NOISE = auto()
    NOISE_NORM = auto()


def step_from_sigmas(sigma, sigmas: torch.Tensor, *, decimals: Optional[int] = 4, output_decimals: int = 2):
    """py/utils.py:682-726: fractional step index of ``sigma`` within a descending schedule."""
    sigma = utils.tensor_item(sigma)
    sigmas = sigmas.detach().cpu()
    if sigmas.ndim == 2:
        sigmas = sigmas.max(dim=0).values
    elif sigmas.ndim != 1:
        raise ValueError(f"Unexpected number of dimensions in sigmas, should be 1 or 2 but got shape {sigmas.shape}")
    return _StepTable(sigmas, decimals).step(sigma, output_decimals)


class _StepTable:
    """Everything ``step_from_sigmas`` derives from the schedule alone (the rounded sigmas, their range), kept as host numbers: a
    sampling run asks once per model evaluation with the same schedule.  The fp32 semantics of the reference's tensor expressions are
    kept (a Python scalar meets an fp32 tensor as an fp32 value)."""

    def __init__(self, sigmas_1d_cpu: torch.Tensor, decimals: Optional[int] = 4):
        import numpy as np

        self.np = np
        sig = sigmas_1d_cpu[:-1]
        self.decimals = decimals
        self.valid = bool(len(sig)) and not bool(torch.any(sig <= 0))
        if not self.valid:
            return
        if decimals is not None:
            sig = sig.round(decimals=decimals)
        self.sig = sig.to(torch.float32).numpy().copy() if sig.dtype == torch.float32 else None
        self.sig_t = sig
        self.vals = sig.tolist()
        lo, hi = sig.aminmax()
        self.lo, self.hi = lo.item(), hi.item()
        self.last = len(sig) - 1

    def step(self, sigma: float, output_decimals: int = 2):
        if not self.valid:
            return None
        if self.decimals is not None:
            sigma = round(sigma, self.decimals)
        if self.sig is not None:
            s32 = self.np.float32(sigma)
            if not self.lo <= float(s32) <= self.hi:
                return None
            idx = int(self.np.abs(self.sig - s32).argmin())
        else:  # unusual dtypes: the tensor expressions themselves
            if not self.lo <= sigma <= self.hi:
                return None
            idx = int((self.sig_t - sigma).abs().argmin())
        at = self.vals[idx]
        if self.decimals is not None:
            at = round(at, self.decimals)
        if sigma == at:
            return float(idx)
        below, above = (idx, idx - 1) if sigma > at else (idx + 1, idx)
        if min(below, above) < 0 or max(below, above) > self.last:
            return None
        s_lo, s_hi = self.vals[below], self.vals[above]
        if s_hi == s_lo:
            return float(idx)
        return round(above + (1.0 - ((sigma - s_lo) / (s_hi - s_lo))), output_decimals)


class _Schedule:
    """What ``WCFGPercentages.build`` derives from (sample_sigmas, rule window) alone -- including which of the reference's failures
    the pair leads to -- computed once per schedule tensor and window."""

    def __init__(self, sigmas: torch.Tensor, start_sigma: float, end_sigma: float):
        self.keep = sigmas  # the key holds id(sigmas): keep the tensor alive while the entry lives
        self.error = None
        if sigmas.ndim == 2:
            sigmas = sigmas.max(dim=0).values
        elif sigmas.ndim != 1:
            self.error = ValueError("Unexpected number of dimensions for sample_sigmas")
            return
        sigmas = sigmas.detach().cpu()
        self.sigma_first, self.sigma_last = sigmas[0].item(), sigmas[-2].item()
        if self.sigma_first <= self.sigma_last:
            self.error = ValueError("Cannot handle non-descending sigmas (possibly Restart or unsampling)")
            return
        self.start, self.end = min(start_sigma, self.sigma_first), max(end_sigma, self.sigma_last)
        self.steps = len(sigmas) - 1  # >= 2 here: a two-entry schedule has sigma_first == sigmas[-2] and was rejected above
        self.table = _StepTable(sigmas)
        enabled = torch.arange(len(sigmas), dtype=torch.int32)[(sigmas <= self.start) & (sigmas >= self.end)]
        self.enabled = None if len(enabled) <= 1 else (enabled[0].item(), enabled[-1].item())


_PCT_CACHE: dict = {}
_SCHED_CACHE: dict = {}


class WCFGPercentages(NamedTuple):
    """py/wavelet_cfg.py:82-212."""

    sigma: float
    sigma_min: float
    sigma_max: float
    sigma_first: Optional[float]
    sigma_last: Optional[float]
    steps: Optional[int]
    step: Optional[float]
    step_first: Optional[int]
    step_last: Optional[int]
    pct_sampling: float
    pct_enabled_sampling: float
    pct_sigmas: Optional[float]
    pct_enabled_sigmas: Optional[float]
    pct_steps: Optional[float]
    pct_enabled_steps: Optional[float]

    def invert(self) -> "WCFGPercentages":
        flip = lambda v: None if v is None else 1.0 - v  # noqa: E731
        return self._replace(pct_sampling=1.0 - self.pct_sampling, pct_enabled_sampling=1.0 - self.pct_enabled_sampling,
                             pct_sigmas=flip(self.pct_sigmas), pct_enabled_sigmas=flip(self.pct_enabled_sigmas),
                             pct_steps=flip(self.pct_steps), pct_enabled_steps=flip(self.pct_enabled_steps))

    def pct_from_schedmode(self, mode: WCFGSchedMode):
        if mode == WCFGSchedMode.MODEL_SAMPLING:
            return self.pct_sampling
        if mode == WCFGSchedMode.SIGMA_RANGE:
            return self.pct_sigmas
        if mode == WCFGSchedMode.ENABLED_MODEL_SAMPLING:
            return self.pct_enabled_sampling
        if mode == WCFGSchedMode.ENABLED_SIGMA_RANGE:
            return self.pct_enabled_sigmas
        if mode == WCFGSchedMode.STEP:
            if self.pct_steps is None:
                raise RuntimeError("Step percentage not available")
            return self.pct_steps
        raise ValueError("Unknown mode")

    @classmethod
    def build(cls, *, ms, start_sigma: float, end_sigma: float, sigma: float, sigmas: Optional[torch.Tensor], **_kw) -> "WCFGPercentages":
        if start_sigma < end_sigma:
            raise ValueError("start/end sigmas out of order")
        # the model's sigma range and the percentages of the rule window's ends are the same numbers at every step of a run
        ends = _PCT_CACHE.get((id(ms), start_sigma, end_sigma))
        if ends is None or ends[0] is not ms:
            if len(_PCT_CACHE) > 64:
                _PCT_CACHE.clear()
            sigma_max, sigma_min = ms.sigma_max.detach().item(), ms.sigma_min.detach().item()
            s0, s1 = min(sigma_max, start_sigma), min(max(sigma_min, end_sigma), sigma_max)
            ends = _PCT_CACHE[(id(ms), start_sigma, end_sigma)] = (ms, sigma_max, sigma_min, s0, s1, cls._pct_of(ms, s0), cls._pct_of(ms, s1))
        _ms, sigma_max, sigma_min, start_sigma, end_sigma, pct_start, pct_end = ends
        sigma = min(max(sigma, sigma_min), sigma_max)
        pct_curr = cls._pct_of(ms, sigma)
        pct_range_curr = (pct_curr - pct_start) / (pct_end - pct_start)
        if sigmas is None:
            # ComfyUI always passes transformer_options["sample_sigmas"]; without them the reference binds neither step_first nor
            # step_last (:188-193) and fails while building the tuple.  Same failure here rather than a silently different schedule.
            raise UnboundLocalError("local variable 'step_first' referenced before assignment")
        key = (id(sigmas), sigmas._version, start_sigma, end_sigma)
        sched = _SCHED_CACHE.get(key)
        if sched is None or sched.keep is not sigmas:
            if len(_SCHED_CACHE) > 64:
                _SCHED_CACHE.clear()
            sched = _SCHED_CACHE[key] = _Schedule(sigmas, start_sigma, end_sigma)
        if sched.error is not None:
            raise type(sched.error)(*sched.error.args)
        sigma_first, sigma_last, start_sigma, end_sigma = sched.sigma_first, sched.sigma_last, sched.start, sched.end
        pct_sigmas = (sigma_first - sigma) / (sigma_first - sigma_last)
        sigma = min(max(sigma, sigma_last), sigma_first)
        pct_enabled_sigmas = 1.0 if start_sigma == end_sigma else (start_sigma - sigma) / (start_sigma - end_sigma)
        steps = sched.steps
        step = sched.table.step(sigma)
        pct_steps = step / (steps - 1) if step is not None else None
        if sched.enabled is None:
            # fewer than two schedule points inside the rule's window: pct_enabled_steps is never bound in the reference (:181-187)
            raise UnboundLocalError("local variable 'pct_enabled_steps' referenced before assignment")
        step_first, step_last = sched.enabled
        pct_enabled_steps = (step - step_first) / (step_last - step_first)  # step None (sigma off the schedule) -> TypeError, as there
        return WCFGPercentages(sigma=sigma, sigma_min=sigma_min, sigma_max=sigma_max, sigma_first=sigma_first, sigma_last=sigma_last,
                               steps=steps, step=step, step_first=step_first, step_last=step_last, pct_sampling=pct_curr,
                               pct_enabled_sampling=pct_range_curr, pct_sigmas=pct_sigmas, pct_enabled_sigmas=pct_enabled_sigmas,
                               pct_steps=pct_steps, pct_enabled_steps=pct_enabled_steps)

    @staticmethod
    def _pct_of(ms, s: float) -> float:
        """``1.0 - (ms.timestep(tensor(s)) / 999).clamp(0, 1).item()`` (:148-153): the model's ``timestep`` is the one tensor call; the
        quotient of its 0-d result by 999 is formed as torch forms it (fp32 for integer and fp32 timesteps), on a host scalar."""
        t = ms.timestep(torch.tensor(s))
        if t.dtype in (torch.float64,):
            q = t.item() / 999
        elif t.dtype in (torch.float32, torch.int64, torch.int32):
            q = float(_np.float32(t.item()) / _np.float32(999.0))
        else:
            return 1.0 - (t / 999).clamp(0, 1).detach().item()
        return 1.0 - min(max(q, 0.0), 1.0)


class WCFGScales(NamedTuple):
    yl_scale: float = 1.0
    yh_scales: object = 1.0

    def get_scales(self, *_a, verbose: bool = False, **_k) -> "WCFGScales":
        if verbose:
            _say(f"WCFG:     {self.pretty_scales()}")
        return self

    def apply_scales(self, yl, yh):
        return wavelet_scaling(yl, yh, yl_scale=self.yl_scale, yh_scales=self.yh_scales)

    def get_and_apply_scales(self, pcts, yl, yh, *, verbose: bool = False):
        return self.get_scales(pcts, yh, verbose=verbose).apply_scales(yl, yh)

    def table(self, yh):
        """(yl_scale, per-band per-orientation scales) for the fused band kernel."""
        return float(self.yl_scale), expand_yh_scales(yh, yh_scales=1.0 if self.yh_scales is None else self.yh_scales)

    def pretty_scales(self):
        return f"low={self.yl_scale:.4f}, high={self.yh_scales!r}"


class WCFGScheduledScale(NamedTuple):
    """py/wavelet_cfg.py:265-323."""

    schedule: WCFGSchedule = WCFGSchedule.LINEAR
    schedule_mode: WCFGSchedMode = WCFGSchedMode.ENABLED_MODEL_SAMPLING
    schedule_offset: float = 0.0
    schedule_offset_after: float = 0.0
    schedule_multiplier: float = 1.0
    schedule_multiplier_after: float = 1.0
    reverse_schedule: bool = False
    reverse_schedule_after: bool = False
    schedule_min: float = 0.0
    schedule_max: float = 1.0

    @classmethod
    def build(cls, **kwargs) -> "WCFGScheduledScale":
        schedule = kwargs.pop("schedule", WCFGSchedule.LINEAR)
        if isinstance(schedule, str):
            schedule = getattr(WCFGSchedule, schedule.upper())
        mode = kwargs.pop("schedule_mode", WCFGSchedMode.ENABLED_MODEL_SAMPLING)
        if isinstance(mode, str):
            mode = getattr(WCFGSchedMode, mode.upper())
        return WCFGScheduledScale(schedule=schedule, schedule_mode=mode, **filter_dict(kwargs, cls._fields))

    def get_b_scale(self, pcts: WCFGPercentages) -> float:
        if self.reverse_schedule:
            pcts = pcts.invert()
        pct = pcts.pct_from_schedmode(self.schedule_mode)
        if pct is None:
            raise RuntimeError("Couldn't get percentage")
        shaped = self.schedule.interp(clamp_float((pct + self.schedule_offset) * self.schedule_multiplier))
        pct = clamp_float((shaped + self.schedule_offset_after) * self.schedule_multiplier_after,
                          minval=clamp_float(self.schedule_min), maxval=clamp_float(self.schedule_max))
        return clamp_float(1.0 - pct) if self.reverse_schedule_after else pct

    def pretty_non_default(self) -> str:
        return pretty_non_default(self, defaults=WCFGScheduledScale())


def _blend_scalar(a: float, b: float, t: float, mode: str) -> float:
    """py/utils.py:33-55 (blend_scalar): plain lerp, or the named blend evaluated in fp64."""
    if mode == "lerp":
        return a * (1.0 - t) + b * t
    if mode == "inject":
        return a + b * t
    if mode == "subtract_b":
        return a - b * t
    raise KeyError(mode)


class WCFGScalesRange(NamedTuple):
    """py/wavelet_cfg.py:326-420."""

    scales_start: WCFGScales = WCFGScales()
    scales_end: Optional[WCFGScales] = None
    scheduler: Optional[WCFGScheduledScale] = None
    blend_mode: str = "lerp"

    @classmethod
    def build(cls, **kwargs):
        start = kwargs.pop("scales_start", None)
        if start is None:
            start = {"yl_scale": kwargs.pop("yl_scale", 1.0), "yh_scales": kwargs.pop("yh_scales", 1.0)}
        end = filter_dict(kwargs.pop("scales_end", {}), WCFGScales._fields)
        if not end or end == start:
            return WCFGScales(yl_scale=start.get("yl_scale", 1.0), yh_scales=start.get("yh_scales", 1.0))
        blend_mode = kwargs.pop("blend_mode", "lerp")
        return WCFGScalesRange(scales_start=WCFGScales(**start), scales_end=WCFGScales(**end),
                               scheduler=WCFGScheduledScale.build(**kwargs), blend_mode=blend_mode)

    def get_scales(self, pcts: WCFGPercentages, yh, *, verbose: bool = False) -> WCFGScales:
        if self.scales_end is None or self.scheduler is None:
            return self.scales_start.get_scales()
        pct = self.scheduler.get_b_scale(pcts)
        start, end = self.scales_start, self.scales_end
        if self.blend_mode == "lerp" and (pct <= 0 or pct >= 1):
            return start if pct <= 0 else end
        s_tab = expand_yh_scales(yh, yh_scales=start.yh_scales)
        e_tab = expand_yh_scales(yh, yh_scales=end.yh_scales)
        out = WCFGScales(yl_scale=_blend_scalar(start.yl_scale, end.yl_scale, pct, self.blend_mode),
                         yh_scales=tuple(tuple(_blend_scalar(a, b, pct, self.blend_mode) for a, b in zip(bs, be)) for bs, be in zip(s_tab, e_tab)))
        if verbose:
            _say(f"WCFG:     {out.pretty_scales()}")
        return out

    def apply_scales(self, yl, yh):
        return self.scales_start.apply_scales(yl, yh)

    def get_and_apply_scales(self, pcts, yl, yh, *, verbose: bool = False):
        return self.get_scales(pcts, yh, verbose=verbose).apply_scales(yl, yh)

    def pretty_non_default(self) -> str:
        return pretty_non_default(self, defaults=WCFGScalesRange())


class WCFGScheduledFloat(NamedTuple):
    value_start: float
    value_end: Optional[float] = None
    scheduler: Optional[WCFGScheduledScale] = None

    @classmethod
    def build(cls, val, *, default_start=None, default_end=None, **_kw) -> "WCFGScheduledFloat":
        if isinstance(val, float):
            return WCFGScheduledFloat(value_start=val)
        if not isinstance(val, dict):
            raise TypeError("Bad type for scheduled float value")
        val = val.copy()
        v0, v1 = val.pop("value_start", default_start), val.pop("value_end", default_end)
        if not isinstance(v0, (float, int)):
            raise TypeError("Bad type for scheduled float start_value")
        if v1 is None:
            return WCFGScheduledFloat(value_start=val)
        if not isinstance(v1, (float, int)):
            raise TypeError("Bad type for scheduled float end_value")
        return WCFGScheduledFloat(value_start=float(v0), value_end=float(v1), scheduler=WCFGScheduledScale.build(**val))

    def get_value(self, pcts: WCFGPercentages) -> float:
        if self.value_end is None or self.scheduler is None:
            return self.value_start
        pct = self.scheduler.get_b_scale(pcts)
        return (1.0 - pct) * self.value_start + pct * self.value_end


class WCFGWaveletSettings(NamedTuple):
    """py/wavelet_cfg.py:468-503 — defaults are API: db4, level 5, symmetric."""

    wave: str = "db4"
    level: int = 5
    padding_mode: str = "symmetric"
    use_1d_dwt: bool = False
    use_dtcwt: bool = False
    biort: str = "near_sym_a"
    qshift: str = "qshift_a"
    inv_wave: Optional[str] = None
    inv_padding_mode: Optional[str] = None
    inv_biort: Optional[str] = None
    inv_qshift: Optional[str] = None

    @classmethod
    def build(cls, **kwargs) -> "WCFGWaveletSettings":
        return WCFGWaveletSettings(**filter_dict(kwargs, cls._fields))

    def make_wavelet(self, **kwargs) -> Wavelet:
        return Wavelet(wave=self.wave, level=self.level, mode=self.padding_mode, use_1d_dwt=self.use_1d_dwt, use_dtcwt=self.use_dtcwt,
                       biort=self.biort, qshift=self.qshift, inv_wave=self.inv_wave, inv_mode=self.inv_padding_mode,
                       inv_biort=self.inv_biort, inv_qshift=self.inv_qshift, **kwargs)

    def pretty_non_default(self) -> str:
        return pretty_non_default(self, defaults=DEFAULT_WAVELETSETTINGS)


DEFAULT_WAVELETSETTINGS = WCFGWaveletSettings()


class WCFGRule(NamedTuple):
    """py/wavelet_cfg.py:508-597."""

    start_sigma: float = math.inf
    end_sigma: float = 0.0
    verbose: bool = False
    blend_mode: str = "lerp"
    blend_strength: WCFGScheduledFloat = WCFGScheduledFloat(1.0)
    fallback_existing: bool = True
    target_mode: WCFGTarget = WCFGTarget.DENOISED
    diff: object = None
    cond: object = None
    uncond: object = None
    final: object = None
    wavelet: WCFGWaveletSettings = DEFAULT_WAVELETSETTINGS
    high_precision_mode: bool = True
    difference_blend_mode: str = "inject"
    difference_blend_strength: WCFGScheduledFloat = WCFGScheduledFloat(1.0)

    @classmethod
    def build(cls, **kwargs) -> "WCFGRule":
        target = kwargs.pop("target_mode", WCFGTarget.DENOISED)
        if isinstance(target, str):
            target = getattr(WCFGTarget, target.upper())
        diff = kwargs.pop("diff", None)
        if diff is None:
            diff = kwargs.pop("difference", None)

        def scales(v):
            return None if v is None else WCFGScalesRange.build(**v)

        cond, uncond, final = (kwargs.pop(k, None) for k in ("cond", "uncond", "final"))
        bs = kwargs.pop("blend_strength", 1.0)
        if not isinstance(bs, (float, int, dict)):
            raise TypeError("Bad type for blend_strength, must be float or dict")
        dbs = kwargs.pop("difference_blend_strength", 1.0)
        if not isinstance(dbs, (float, int, dict)):
            raise TypeError("Bad type for difference_blend_strength, must be float or dict")
        return WCFGRule(target_mode=target, diff=scales(diff), cond=scales(cond), uncond=scales(uncond), final=scales(final),
                        blend_strength=WCFGScheduledFloat(bs), difference_blend_strength=WCFGScheduledFloat(dbs),
                        wavelet=WCFGWaveletSettings.build(**kwargs), **filter_dict(kwargs, cls._fields))

    def make_wavelet(self, **kwargs) -> Wavelet:
        return self.wavelet.make_wavelet(**kwargs)

    def get_and_apply_scales(self, name: str, pcts, yl, yh, *, verbose: bool = False):
        return getattr(self, name).get_scales(pcts, yh).apply_scales(yl, yh)

    def scale_table(self, name: str, pcts, yh):
        spec = getattr(self, name)
        if spec is None:
            return 1.0, None
        return spec.get_scales(pcts, yh).table(yh)

    def pretty_non_default(self) -> str:
        return pretty_non_default(self, defaults=DEFAULT_RULE)


DEFAULT_RULE = WCFGRule()


class WCFGRules(NamedTuple):
    rules: Sequence = ()

    def __len__(self) -> int:
        return len(self.rules)

    def __getitem__(self, idx: int) -> WCFGRule:
        return self.rules[idx]

    def __bool__(self) -> bool:
        return bool(self.rules)

    def get_rule(self, sigma: float) -> Optional[WCFGRule]:
        for rule in self.rules:
            if rule.end_sigma <= sigma <= (math.inf if rule.start_sigma < 0 else rule.start_sigma):
                return rule
        return None

    @classmethod
    def build(cls, **params) -> "WCFGRules":
        params = params.copy()
        extra = params.pop("rules", ())
        return WCFGRules(rules=(WCFGRule.build(**params), *(WCFGRule.build(**p) for p in extra)))


class WCFGContext(NamedTuple):
    cond: torch.Tensor
    uncond: torch.Tensor
    x: torch.Tensor
    sigma: torch.Tensor
    wavelet: Wavelet
    dtype: torch.dtype
    op_kwargs: dict


class _BandShape:  # expand_yh_scales only needs the band count and the orientation count
    shape = (1, 1, 3, 1, 1)


def _to_dtype(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if t.dtype == dtype:
        return t.contiguous()
    if t.dtype == torch.float32 and dtype == torch.float64 and t.is_cuda:
        return hip_lib.cast_f32_f64(t.contiguous())
    return t.to(dtype).contiguous()


_LOWPASS_ARGS: dict = {}
_FUSED_CALLS: dict = {}


def _rule_is_static(rule: "WCFGRule") -> bool:
    """True when nothing in the rule is scheduled: its scale tables and strengths are the same numbers at every step."""
    for spec in (rule.diff, rule.cond, rule.uncond, rule.final):
        if spec is not None and not isinstance(spec, WCFGScales):
            return False
    return all(isinstance(v.value_start, (int, float)) and (v.value_end is None or v.scheduler is None)
               for v in (rule.blend_strength, rule.difference_blend_strength))


def _per_latent(t: torch.Tensor, sigma: torch.Tensor, *, divide: bool) -> torch.Tensor:
    """``t / sigma`` or ``t * sigma`` with one sigma per latent (``sigma`` shaped [B, 1, ...], py/wavelet_cfg.py:697-699,746-747,832-833)
    through the row kernel: (t - 0) / s and t * s + 0 are the reference's quotient / product exactly."""
    rows = t.shape[0]
    s = utils.as_f32(sigma.reshape(-1)).to(t.device)
    if s.numel() == 1 and rows > 1:
        s = s.expand(rows)
    elif s.numel() != rows:  # the kernel reads one value per row: refuse what torch's broadcast refuses, with its words
        raise RuntimeError(f"The size of tensor a ({rows}) must match the size of tensor b ({s.numel()}) at non-singleton dimension 0")
    s = s.contiguous()
    zero = torch.zeros_like(s)
    return hip_lib.row_affine(0 if divide else 1, utils.as_f32(t).contiguous(), rows, t.numel() // max(rows, 1), zero, s)


def _check_broadcast(a: tuple, b: tuple) -> None:
    """torch's refusal to broadcast two shapes, with its message (the first mismatch counted from the trailing dimension)."""
    n = max(len(a), len(b))
    pa, pb = (1,) * (n - len(a)) + a, (1,) * (n - len(b)) + b
    for d in range(n - 1, -1, -1):
        if pa[d] != pb[d] and pa[d] != 1 and pb[d] != 1:
            raise RuntimeError(f"The size of tensor a ({pa[d]}) must match the size of tensor b ({pb[d]}) at non-singleton dimension {d}")
    if pa != pb:
        raise hip_lib.SonarHipError(f"WaveletCFG: blending {a} with {b} needs broadcasting, which the HIP blend kernels do not do")


def _reconstructs(w) -> bool:
    """IDWT(DWT(t)) == t (after the crop) needs the analysis and synthesis banks of ONE wavelet (dmey is only approximately a wavelet)
    and extension modes that agree on the coefficient grid: pywt's non-periodised modes are interchangeable (the valid region never
    sees the extension), but a periodised transform keeps H/2 coefficients on a grid shifted by L/2 - 1 samples, so pairing it with
    any other mode returns a SHIFTED reconstruction for filters longer than two taps -- the reference computes exactly that, and such
    rules take the band-by-band path."""
    if getattr(w, "wave", None) is None or w.wave != w.inv_wave or w.wave == "dmey" or len(w.dec_lo) != len(w.rec_lo):
        return False
    return len(w.dec_lo) == 2 or (w.mode == "periodization") == (w.inv_mode == "periodization")


class WaveletCFG:
    """py/wavelet_cfg.py:626-842 — a ComfyUI ``sampler_cfg_function``."""

    # Rules that need the coefficient bands: the single-launch kernel with every band resident in LDS (fast path 1: 1.1-2.0 x the step's
    # 16N bytes of traffic) or level 1 in tile launches + the deeper levels resident (fast path 2: 2.2-2.9 x).  None: by precision --
    # fp32 arithmetic takes the single-launch kernel (round 5: 8 % faster than the tiles on difference-only rules, 12 % on rules that
    # also scale cond / uncond, at 1.5 x / 2.0 x of 16N instead of 2.3 x / 2.2 x), fp64 the tiles (its planes leave a CU one workgroup:
    # 231 against 223-233 us on difference rules, 331 against 275 on the others).  True / False force a route (bench.py and the tests
    # time and check both).
    single_launch_bands = None

    def _bands_first(self, ctx) -> bool:
        return ctx.dtype == torch.float32 if self.single_launch_bands is None else bool(self.single_launch_bands)

    def __init__(self, *, existing_cfg: Optional[Callable], rules: WCFGRules, operation_cond=None, operation_uncond=None,
                 operation_fallback_cfg=None, operation_wavelet_cfg=None, operation_result=None):
        self.wavelet_cache = {}
        self.rules = rules
        self.fallback_cfg_function = existing_cfg if existing_cfg is not None and (not rules or rules[0].fallback_existing) else self.basic_cfg_function
        self.operation_cond = operation_cond
        self.operation_uncond = operation_uncond
        self.operation_fallback_cfg = operation_fallback_cfg
        self.operation_wavelet_cfg = operation_wavelet_cfg
        self.operation_result = operation_result
        self._likely_rule = None  # the rule the previous call matched: the next call launches for it before its own sigma is known

    @staticmethod
    def basic_cfg_function(args: dict) -> torch.Tensor:
        """py/wavelet_cfg.py:657-660: x - (uncond + (cond - uncond) * scale)."""
        x, scale = args["input"], args["cond_scale"]
        uncond, cond = args["uncond_denoised"], args["cond_denoised"]
        mixed = hip_lib.blend("inject", utils.as_f32(uncond), hip_lib.blend("subtract_b", utils.as_f32(cond), utils.as_f32(uncond), 1.0), scale)
        return hip_lib.blend("subtract_b", utils.as_f32(x), mixed, 1.0)

    @staticmethod
    def maybe_op(t, mop, **kwargs):
        if mop is None:
            return t
        return mop(latent=t, **(kwargs if getattr(mop, "EXTENDED_LATENT_OPERATION", None) else {}))

    def get_context(self, *, rule: WCFGRule, args: dict) -> WCFGContext:
        """py/wavelet_cfg.py:677-727."""
        sigma_orig = sigma = args["sigma"]
        x = args["input"]
        if x.ndim == 3 and not rule.wavelet.use_1d_dwt:
            raise RuntimeError("Enable use_1d_dwt mode for 3D latents.")
        if x.ndim < 3:
            raise RuntimeError("Wavelet CFG can't handle latents with 2 or less dimensions.")
        if sigma.ndim != x.ndim:
            sigma = sigma.reshape(x.shape[0], *((1,) * (x.ndim - sigma.ndim)))
        if rule.target_mode in {WCFGTarget.NOISE, WCFGTarget.NOISE_NORM}:
            cond, uncond = args["cond"], args["uncond"]
            if rule.target_mode == WCFGTarget.NOISE_NORM:
                cond, uncond = _per_latent(cond, sigma, divide=True), _per_latent(uncond, sigma, divide=True)
        elif rule.target_mode == WCFGTarget.DENOISED:
            cond, uncond = args["cond_denoised"], args["uncond_denoised"]
        else:
            raise ValueError("Bad target mode")
        op_kwargs = {"sigma": sigma_orig, "cond": cond, "uncond": uncond, "cond_scale": args["cond_scale"], "raw_args": args}
        cond = self.maybe_op(cond, self.operation_cond, **op_kwargs)
        uncond = self.maybe_op(uncond, self.operation_uncond, **op_kwargs)
        eff_dtype = torch.float64 if rule.high_precision_mode else x.dtype
        wavelet = self.wavelet_cache.get(id(rule))
        if wavelet is None:
            wavelet = self.wavelet_cache[id(rule)] = rule.make_wavelet()
        wavelet = wavelet.to(device=x.device, dtype=eff_dtype)
        if rule.wavelet.use_1d_dwt:
            cond = cond.flatten(start_dim=2)
            uncond = uncond.flatten(start_dim=2)
        elif x.ndim > 4:
            cond = cond.flatten(start_dim=1, end_dim=cond.ndim - 3)
            uncond = uncond.flatten(start_dim=1, end_dim=uncond.ndim - 3)
        return WCFGContext(cond=cond, uncond=uncond, x=x, sigma=sigma, wavelet=wavelet, dtype=eff_dtype, op_kwargs=op_kwargs)

    @classmethod
    def wavelet_cfg_fused(cls, *, rule: WCFGRule, ctx: WCFGContext, pcts) -> Optional[torch.Tensor]:
        """x - IDWT(band(DWT(cond), DWT(uncond))) through ``sonar_wcfg_fused_*``; None when the shape / dtype is outside its scope."""
        w = ctx.wavelet
        if ctx.cond.ndim != 4 or ctx.cond.dtype != torch.float32 or ctx.uncond.dtype != torch.float32 or ctx.dtype not in (torch.float32, torch.float64):
            return None
        levels = w.level
        if levels < 1 or getattr(w, "use_dtcwt", False):
            return None  # the dual-tree transform goes band by band (wavelet_cfg_raw)
        static = _rule_is_static(rule)
        pr = _reconstructs(w)
        key = (id(rule), ctx.dtype, pr)
        hit = _FUSED_CALLS.get(key) if static else None
        if hit is not None and hit[0] is rule and hit[1] is w:
            call = hit[2]
        else:
            fake_yh = [_BandShape] * levels
            tabs = {name: rule.scale_table(name, pcts, fake_yh) for name in ("cond", "uncond", "diff", "final")}

            def row(name, j):
                table = tabs[name][1]
                if table is None or j >= len(table):
                    return (1.0, 1.0, 1.0)
                sc = table[j]
                sc = (float(sc),) * 3 if isinstance(sc, (int, float)) else tuple(float(v) for v in sc)
                return (sc + (1.0,) * 3)[:3]

            names = ("cond", "uncond", "diff", "final")
            call = hip_lib.FusedCall(
                levels=levels, dec_lo=w.dec_lo, dec_hi=w.dec_hi, mode=w.mode, rec_lo=w.rec_lo, rec_hi=w.rec_hi, inv_mode=w.inv_mode,
                yl_scales=[tabs[n][0] for n in names], yh_scales=[[row(n, j) for n in names] for j in range(levels)],
                blend_mode=rule.difference_blend_mode, strength=rule.difference_blend_strength.get_value(pcts), subtract_from_x=True,
                high_precision=ctx.dtype == torch.float64, perfect_reconstruction=pr)
            if static:  # the tables of an unscheduled rule are the same numbers at every step: converted once
                if len(_FUSED_CALLS) > 64:
                    _FUSED_CALLS.clear()
                _FUSED_CALLS[key] = (rule, w, call)
        return call(ctx.cond.contiguous(), ctx.uncond.contiguous(), ctx.x.contiguous())

    @classmethod
    def wavelet_cfg_bands(cls, *, rule: WCFGRule, ctx: WCFGContext, pcts) -> Optional[torch.Tensor]:
        """x - result with the coefficient bands resident in LDS (``sonar_wcfg_bands_*``, csrc/dwt_bands.h) for any scale tables, when the
        wavelet pair reconstructs and the blend is linear in the bands (lerp / inject / subtract_b): difference-only rules are ONE launch
        on cond - uncond; a rule that also scales cond / uncond / final is A . DWT(cond) + B . DWT(uncond) per band -- two launches, one
        tensor each.  None when the rule / wavelet / shape is outside that scope."""
        w = ctx.wavelet
        if ctx.cond.ndim != 4 or ctx.cond.dtype != torch.float32 or ctx.uncond.dtype != torch.float32 or ctx.dtype not in (torch.float32, torch.float64):
            return None
        levels = w.level
        mode = rule.difference_blend_mode
        if levels < 1 or getattr(w, "use_dtcwt", False) or not _reconstructs(w) or mode not in ("lerp", "inject", "subtract_b"):
            return None
        fake_yh = [_BandShape] * levels
        tabs = {name: rule.scale_table(name, pcts, fake_yh) for name in ("cond", "uncond", "diff", "final")}

        def row(name, j):
            table = tabs[name][1]
            if table is None or j >= len(table):
                return (1.0, 1.0, 1.0)
            sc = table[j]
            sc = (float(sc),) * 3 if isinstance(sc, (int, float)) else tuple(float(v) for v in sc)
            return (sc + (1.0,) * 3)[:3]

        t = float(rule.difference_blend_strength.get_value(pcts))
        sign = -1.0 if mode == "subtract_b" else 1.0
        keep = 1.0 - t if mode == "lerp" else 1.0  # blend(a, b, t) = keep * a + sign * t * b
        common = dict(levels=levels, dec_lo=w.dec_lo, dec_hi=w.dec_hi, rec_lo=w.rec_lo, rec_hi=w.rec_hi, mode=w.mode, inv_mode=w.inv_mode,
                      subtract_from_x=True, high_precision=ctx.dtype == torch.float64)
        cond, uncond, x = ctx.cond.contiguous(), ctx.uncond.contiguous(), ctx.x.contiguous()
        plain = all(float(tabs[n][0]) == 1.0 and all(v == 1.0 for j in range(levels) for v in row(n, j)) for n in ("cond", "uncond", "final"))
        if plain:
            return hip_lib.wcfg_bands(cond, uncond, x, yh_scales=[row("diff", j) for j in range(levels)], yl_scale=float(tabs["diff"][0]),
                                      ku=keep, kt=sign * t, **common)

        def coeffs(s_c, s_u, s_d, s_f):  # blend(s_u U, s_d (s_c C - s_u U), t) s_f = A C + B U
            return sign * t * s_d * s_c * s_f, (keep - sign * t * s_d) * s_u * s_f

        per_band = [[coeffs(*(row(n, j)[k] for n in ("cond", "uncond", "diff", "final"))) for k in range(3)] for j in range(levels)]
        yl_a, yl_b = coeffs(*(float(tabs[n][0]) for n in ("cond", "uncond", "diff", "final")))
        out = hip_lib.wcfg_bands(uncond, None, x, yh_scales=[[ab[1] for ab in lvl] for lvl in per_band], yl_scale=yl_b, ku=0.0, kt=1.0, **common)
        if out is None:
            return None
        return hip_lib.wcfg_bands(cond, None, out, out, yh_scales=[[ab[0] for ab in lvl] for lvl in per_band], yl_scale=yl_a, ku=0.0, kt=1.0, **common)

    @classmethod
    def wavelet_cfg_lowpass(cls, *, rule: WCFGRule, ctx: WCFGContext, pcts) -> Optional[torch.Tensor]:
        """x - result in ONE launch for rules that only scale the difference bands with one scale per level (the node's placeholder
        rule): by linearity and perfect reconstruction the step collapses to a low-pass pyramid of cond - uncond kept in LDS
        (``sonar_wcfg_lowpass_*``, csrc/dwt_lowpass.h).  None when the rule / wavelet / shape is outside that scope."""
        launch = cls._lowpass_launch(rule=rule, ctx=ctx, pcts=pcts)
        return None if launch is None else launch()

    @classmethod
    def _lowpass_launch(cls, *, rule: WCFGRule, ctx: WCFGContext, pcts):
        """``wavelet_cfg_lowpass`` up to the launch (``pcts`` is only read when the rule is scheduled)."""
        w = ctx.wavelet
        if ctx.cond.ndim != 4 or ctx.cond.dtype != torch.float32 or ctx.uncond.dtype != torch.float32 or ctx.dtype not in (torch.float32, torch.float64):
            return None
        levels = w.level
        if levels < 1 or not _reconstructs(w):
            return None
        static = _rule_is_static(rule)
        hit = _LOWPASS_ARGS.get(id(rule)) if static else None
        if hit is not None and hit[0] is rule:
            plan = hit[1]
        else:
            plan = cls._lowpass_plan(rule, pcts, levels)
            if static:
                if len(_LOWPASS_ARGS) > 64:
                    _LOWPASS_ARGS.clear()
                _LOWPASS_ARGS[id(rule)] = (rule, plan)
        if plan is None:
            return None
        g, ku, kt = plan
        return hip_lib.wcfg_lowpass_plan(ctx.cond.contiguous(), ctx.uncond.contiguous(), ctx.x.contiguous(), levels=levels, dec_lo=w.dec_lo,
                                         rec_lo=w.rec_lo, mode=w.mode, inv_mode=w.inv_mode, g=g, ku=ku, kt=kt, subtract_from_x=True,
                                         high_precision=ctx.dtype == torch.float64)

    def _speculate(self, args: dict):
        """While the sigma read is in flight: (rule, context, result) of the fast path for the rule the previous call matched, LAUNCHED
        before sigma's value is known, when that rule is unscheduled and nothing in its preparation depends on the value or runs user
        operations.  The kernels only write their own fresh output, so a call that turns out to match another rule (a window boundary)
        drops the tensor and takes the ordinary order; the GPU works instead of waiting for the host to learn a number that almost
        always selects the same rule.  None otherwise (and on any error: the ordinary path raises it where the reference does)."""
        rule = self._likely_rule
        if rule is None or self.operation_cond is not None or self.operation_uncond is not None or self.operation_wavelet_cfg is not None:
            return None
        x = args.get("input")
        if (not torch.is_tensor(x) or x.ndim != 4 or x.dtype != torch.float32 or rule.target_mode != WCFGTarget.DENOISED or rule.wavelet.use_1d_dwt
                or rule.blend_mode != "lerp" or not _rule_is_static(rule) or rule.blend_strength.get_value(None) != 1.0):
            return None
        try:
            ctx = self.get_context(rule=rule, args=args)
            launch = self._lowpass_launch(rule=rule, ctx=ctx, pcts=None)
            if launch is not None:
                result = launch()
            else:
                result = self.wavelet_cfg_bands(rule=rule, ctx=ctx, pcts=None) if self._bands_first(ctx) else None
                if result is None:
                    result = self.wavelet_cfg_fused(rule=rule, ctx=ctx, pcts=None)
        except Exception:  # noqa: BLE001 -- reported by the ordinary path, in the reference's order
            return None
        return None if result is None else (rule, ctx, result)

    @staticmethod
    def _lowpass_plan(rule: WCFGRule, pcts, levels: int):
        """(g, ku, kt) of ``sonar_wcfg_lowpass_*`` for this rule at these percentages, or None when the rule needs the bands."""
        fake_yh = [_BandShape] * levels
        for name in ("cond", "uncond", "final"):
            yl, table = rule.scale_table(name, pcts, fake_yh)
            if yl != 1.0 or (table is not None and any(float(v) != 1.0 for row in table for v in (row if isinstance(row, (tuple, list)) else (row,)))):
                return None
        yl, table = rule.scale_table("diff", pcts, fake_yh)
        d = []
        for j in range(levels):
            row = (1.0,) if table is None or j >= len(table) else table[j]
            row = tuple(float(v) for v in row) if isinstance(row, (tuple, list)) else (float(row),)
            if any(v != row[0] for v in row[:3]):
                return None  # per-orientation scales need the bands themselves
            d.append(row[0])
        g = [d[0], *(d[j] - d[j - 1] for j in range(1, levels)), float(yl) - d[-1]]
        t = float(rule.difference_blend_strength.get_value(pcts))
        mode = rule.difference_blend_mode
        if mode == "inject":
            ku, kt = 1.0, t
        elif mode == "lerp":
            ku, kt = 1.0 - t, t
        elif mode == "subtract_b":
            ku, kt = 1.0, -t
        else:
            return None
        return g, ku, kt

    @classmethod
    def wavelet_cfg_raw(cls, *, rule: WCFGRule, ctx: WCFGContext, pcts) -> torch.Tensor:
        """The transform-domain result in ``ctx.dtype`` at the reconstruction's own size (before cast/crop)."""
        condw = ctx.wavelet.forward(_to_dtype(ctx.cond, ctx.dtype))
        uncondw = ctx.wavelet.forward(_to_dtype(ctx.uncond, ctx.dtype))
        yh = condw[1]
        tabs = {name: rule.scale_table(name, pcts, yh) for name in ("cond", "uncond", "diff", "final")}
        strength = rule.difference_blend_strength.get_value(pcts)
        mode = rule.difference_blend_mode

        def band_scales(j):
            return [None if tabs[name][1] is None else tabs[name][1][j] if j < len(tabs[name][1]) else None for name in ("cond", "uncond", "diff", "final")]

        yl = hip_lib.wcfg_band(condw[0], uncondw[0], 1, *[[tabs[n][0]] for n in ("cond", "uncond", "diff", "final")], mode, strength, out=condw[0])
        if condw[0].ndim == 3:
            # 1-D transform: one detail band [B, C, l] per level; the reference's band scaling reaches only coefficient 0 of each row
            # there (wavelet_functions.wavelet_scaling), the blend every coefficient
            def head(s):
                return None if s is None else float(s[0] if isinstance(s, (tuple, list)) else s)

            out_yh = [hip_lib.wcfg_band_head(c, u, *[head(s) for s in band_scales(j)], mode, strength, out=c)
                      for j, (c, u) in enumerate(zip(condw[1], uncondw[1]))]
        else:
            # groups = orientations per band: 3 for the DWT's [B, C, 3, h, w], 6 for the dual-tree transform's [B, C, 6, h, w, 2]
            out_yh = [hip_lib.wcfg_band(c, u, c.shape[2], *band_scales(j), mode, strength, out=c) for j, (c, u) in enumerate(zip(condw[1], uncondw[1]))]
        return ctx.wavelet.inverse(yl, out_yh)

    @classmethod
    def wavelet_cfg(cls, *, rule: WCFGRule, ctx: WCFGContext, pcts) -> torch.Tensor:
        """py/wavelet_cfg.py:750-791."""
        return cls.wavelet_cfg_raw(rule=rule, ctx=ctx, pcts=pcts).to(dtype=ctx.x.dtype)

    def process_output(self, *, result: torch.Tensor, rule: WCFGRule, ctx: WCFGContext) -> torch.Tensor:
        """py/wavelet_cfg.py:729-748 (crop to x, DENOISED: x - result; NOISE_NORM: * sigma)."""
        x_shape = ctx.x.shape
        if rule.wavelet.use_1d_dwt:
            result = result[..., : ctx.cond.shape[2]].reshape(x_shape)
        elif ctx.x.ndim > 4:
            result = result[..., : x_shape[-2], : x_shape[-1]].reshape(x_shape)
        else:
            result = result[tuple(slice(None, sz) for sz in x_shape)]
        if rule.target_mode == WCFGTarget.DENOISED:
            result = hip_lib.blend("subtract_b", utils.as_f32(ctx.x), utils.as_f32(result), 1.0)
        elif rule.target_mode == WCFGTarget.NOISE_NORM:
            result = _per_latent(result, ctx.sigma, divide=False)
        return self.maybe_op(result, self.operation_wavelet_cfg, **ctx.op_kwargs)

    def __call__(self, args: dict) -> torch.Tensor:
        """py/wavelet_cfg.py:793-842."""
        sigma = args["sigma"]
        ready = None
        if sigma.is_cuda and sigma.dtype == torch.float32 and sigma.numel() > 0:
            # `sigma.max().item()`: one launch into pinned memory; what does not need the value is prepared while it is in flight
            token = hip_lib.max_to_host_begin(sigma)
            try:
                ready = self._speculate(args)
            finally:
                sigma_f = hip_lib.max_to_host_end(token)
        else:
            sigma_f = sigma.max().item()
        rule = self.rules.get_rule(sigma_f)
        self._likely_rule = rule  # None: outside every rule's window -- the next call most likely is too, and launches nothing ahead
        if rule is None:
            return self.fallback_cfg_function(args)
        if rule.verbose:
            _say(f"\nWCFG: Rule matched, sigma={sigma_f:.4f}, rule={rule.pretty_non_default()}")
        if ready is not None and ready[0] is rule:
            # already launched; the percentages (whose errors are the reference's errors, py/wavelet_cfg.py:155-222) are checked under the kernel
            low = ready[2]
            WCFGPercentages.build(ms=args["model"].model_sampling, start_sigma=rule.start_sigma, end_sigma=rule.end_sigma, sigma=sigma_f,
                                  sigmas=args.get("model_options", {}).get("transformer_options", {}).get("sample_sigmas"))
            return self.maybe_op(low, self.operation_result, **ready[1].op_kwargs).contiguous()
        model = args["model"]
        pcts = WCFGPercentages.build(ms=model.model_sampling, start_sigma=rule.start_sigma, end_sigma=rule.end_sigma, sigma=sigma_f,
                                     sigmas=args.get("model_options", {}).get("transformer_options", {}).get("sample_sigmas"))
        wcfg_blend = rule.blend_strength.get_value(pcts)
        if rule.blend_mode == "lerp" and wcfg_blend == 0:
            return self.maybe_op(self.fallback_cfg_function(args), self.operation_fallback_cfg, sigma=sigma, cond=args["cond_denoised"],
                                 uncond=args["uncond_denoised"], raw_args=args)
        ctx = self.get_context(rule=rule, args=args)
        plain = rule.blend_mode == "lerp" and wcfg_blend == 1.0
        x = ctx.x
        if (plain and rule.target_mode == WCFGTarget.DENOISED and x.ndim == 4 and x.dtype == torch.float32 and self.operation_wavelet_cfg is None
                and not rule.wavelet.use_1d_dwt):
            # fast path 0: difference-only rule with one scale per level -> low-pass pyramid in LDS, one launch, 16N bytes of traffic
            low = self.wavelet_cfg_lowpass(rule=rule, ctx=ctx, pcts=pcts)
            if low is not None:
                return self.maybe_op(low, self.operation_result, **ctx.op_kwargs).contiguous()
            # fast path 1 (`single_launch_bands`: the default with fp32 arithmetic): any scale tables with ALL coefficient bands resident in
            # LDS -- one launch for difference-only rules, two otherwise.  Least HBM traffic, but a plane's coefficients fill half a CU's
            # LDS and the launch runs at one or two workgroups per CU: slower than fast path 2 in fp64 at SDXL size
            if self._bands_first(ctx):
                bands = self.wavelet_cfg_bands(rule=rule, ctx=ctx, pcts=pcts)
                if bands is not None:
                    return self.maybe_op(bands, self.operation_result, **ctx.op_kwargs).contiguous()
            # fast path 2: level 1 in two LDS-staged tile launches (cond + uncond analysed together, band arithmetic before the store, the
            # synthesis writes x - result), the deeper levels in one launch with their coefficients resident in LDS
            fused = self.wavelet_cfg_fused(rule=rule, ctx=ctx, pcts=pcts)
            if fused is not None:
                return self.maybe_op(fused, self.operation_result, **ctx.op_kwargs).contiguous()
            # fast path 3: cast + crop + (x - result) fused in one kernel straight from the fp64/fp32 reconstruction
            raw = self.wavelet_cfg_raw(rule=rule, ctx=ctx, pcts=pcts)
            result = hip_lib.wcfg_output(x.contiguous(), raw, x.shape, True)
            return self.maybe_op(result, self.operation_result, **ctx.op_kwargs).contiguous()
        result = self.wavelet_cfg(rule=rule, ctx=ctx, pcts=pcts)
        if not plain:
            normal = self.maybe_op(self.fallback_cfg_function(args), self.operation_fallback_cfg, **ctx.op_kwargs)
            if rule.target_mode == WCFGTarget.DENOISED:
                normal = hip_lib.blend("subtract_b", utils.as_f32(ctx.x), utils.as_f32(normal), 1.0)
            elif rule.target_mode == WCFGTarget.NOISE_NORM:
                normal = _per_latent(normal, ctx.sigma, divide=True)
            # the reference blends BEFORE it crops (:825-836): a reconstruction larger than the latent (odd sizes) or of another rank
            # (flattened video / 1-D latents) does not broadcast against the fallback result, and torch says so
            _check_broadcast(tuple(normal.shape), tuple(result.shape))
            result = utils.BLENDING_MODES[rule.blend_mode](normal, result.contiguous(), wcfg_blend)
        result = self.process_output(result=result, ctx=ctx, rule=rule)
        return self.maybe_op(result, self.operation_result, **ctx.op_kwargs).contiguous()
