"""Sonar momentum samplers on MI355X (API of the reference's ``py/sonar.py``).

The per-step tensor work of the three samplers is fused into HIP kernels:
  * ``momentum_step``           -> ``sonar_momentum_euler_f32``: reads x, denoised, history; writes x', history'
                                   (both history updates of py/sonar.py:262-307 happen in registers)
  * ``SonarDPMPPSDE`` half steps -> ``sonar_dpmpp_stage{1,2}_f32``
``get_momentum_denoised`` / ``get_momentum_d`` stay available with the reference's signature and are
built from the blend kernels.  Sigmas and the DPM-Solver scalars are host floats computed in fp32
tensor arithmetic the way the reference computes them (0-d tensors), so the device scalars match.
"""
from __future__ import annotations

import importlib
from enum import Enum, auto
from sys import stderr
from typing import Any, Callable, NamedTuple, Optional

import torch
from torch import Tensor

from .. import hip_lib
from . import noise, utils

try:  # ComfyUI present: use its progress bar and sampler registry
    from tqdm.auto import trange
except ImportError:  # pragma: no cover
    def trange(n, disable=None):
        return range(n)


class HistoryType(Enum):
    ZERO = auto()
    RAND = auto()
    SAMPLE = auto()
    SAMPLE_NORM = auto()


class GuidanceType(Enum):
    LINEAR = auto()
    EULER = auto()


class GuidanceConfig(NamedTuple):
    guidance_type: GuidanceType = GuidanceType.LINEAR
    factor: float = 0.01
    start_step: int = 1
    end_step: int = 9999
    latent: Optional[Tensor] = None


class MomentumMode(Enum):
    CLASSIC = auto()
    NEW = auto()
    DENOISED = auto()


class SonarConfig(NamedTuple):
    """py/sonar.py:46-67 — field names are API (``sonar_params`` YAML keys)."""

    momentum: float = 0.95
    momentum_hist: float = 0.75
    direction: float = 1.0
    momentum_start_step: int = 0
    momentum_end_step: int = 9999
    always_update_history: bool = True
    momentum_mode: MomentumMode = MomentumMode.NEW
    init: HistoryType = HistoryType.ZERO
    noise_type: Optional[noise.NoiseType] = None
    custom_noise: Any = None
    rand_init_noise_type: Optional[noise.NoiseType] = None
    rand_init_noise_multiplier: float = 1.0
    guidance: Optional[GuidanceConfig] = None
    blend_mode: str = "lerp"
    momentum_blend_mode: Optional[str] = None
    history_blend_mode: Optional[str] = None
    guidance_blend_mode: Optional[str] = None

    def get_with_default(self, k: str, default: Any) -> Any:
        val = getattr(self, k)
        return val if val is not None else default


def to_d(x: Tensor, sigma, denoised: Tensor) -> Tensor:
    """k-diffusion ``to_d`` = (x - denoised) / sigma, one kernel."""
    return hip_lib.to_d(utils.as_f32(x), utils.as_f32(denoised), float(sigma))


def get_ancestral_step(sigma_from, sigma_to, eta: float = 1.0):
    """k-diffusion ancestral split (host scalars / 0-d tensors; ComfyUI is not vendored by the reference)."""
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to**2 * (sigma_from**2 - sigma_to**2) / sigma_from**2) ** 0.5)
    sigma_down = (sigma_to**2 - sigma_up**2) ** 0.5
    return sigma_down, sigma_up


class SonarBase:
    """py/sonar.py:70-320."""

    DEFAULT_NOISE_TYPE = noise.NoiseType.GAUSSIAN

    def __init__(self, cfg: SonarConfig) -> None:
        self.history_d: Optional[Tensor] = None
        self.cfg = cfg
        self.noise_sampler = None
        self._fresh_history = False
        base = cfg.blend_mode
        self.blend_name = base
        self.momentum_blend_name = cfg.get_with_default("momentum_blend_mode", base)
        self.history_blend_name = cfg.get_with_default("history_blend_mode", base)
        self.guidance_blend_name = cfg.get_with_default("guidance_blend_mode", base)
        self.blend = utils.BLENDING_MODES[base]
        self.momentum_blend = utils.BLENDING_MODES[self.momentum_blend_name]
        self.history_blend = utils.BLENDING_MODES[self.history_blend_name]
        self.guidance_blend = utils.BLENDING_MODES[self.guidance_blend_name]

    _cfg_fixups = (("momentum_mode", MomentumMode), ("init", HistoryType), ("noise_type", noise.NoiseType))

    @classmethod
    def get_config(cls, cfg: Optional[SonarConfig] = None, ext: Optional[dict] = None) -> SonarConfig:
        """py/sonar.py:104-131: ``sonar_params`` overrides with string -> enum fix-ups."""
        overrides = ext.copy() if ext is not None else {}
        for key, enum_class in cls._cfg_fixups:
            if key not in overrides:
                continue
            val = overrides[key]
            if isinstance(val, str):
                member = getattr(enum_class, val.strip().upper(), None)
                if member is None:
                    valid = ", ".join(enum_class.__members__.keys())
                    raise ValueError(f"Bad value for {key} of type enum {enum_class.__name__}, must be one of the following: {valid}")
                overrides[key] = member
            elif not isinstance(val, enum_class):
                raise TypeError(f"Bad parameter type for {key}: Must be valid string or instance of {enum_class.__name__}")
        if cfg is None:
            return SonarConfig(**overrides)
        return SonarConfig(**(cfg._asdict() | overrides))

    def set_noise_sampler(self, x: Tensor, sigmas: Tensor, noise_sampler: Optional[Callable], seed: Optional[int] = None) -> Callable:
        """py/sonar.py:133-167."""
        sigma_min, sigma_max = sigmas[sigmas > 0].min(), sigmas.max()
        if noise_sampler is not None and self.cfg.noise_type not in {None, self.DEFAULT_NOISE_TYPE}:
            print("Sonar: Warning: Noise sampler supplied, overriding noise type from settings", file=stderr)
        if self.cfg.custom_noise:
            noise_sampler = self.cfg.custom_noise.make_noise_sampler(x, sigma_min, sigma_max, seed=seed)
        elif noise_sampler is None:
            noise_sampler = noise.get_noise_sampler(self.cfg.noise_type or self.DEFAULT_NOISE_TYPE, x, sigma_min, sigma_max,
                                                    seed=seed, cpu=True, normalized=True)
        self.noise_sampler = noise_sampler
        return noise_sampler

    def draw_noise(self, sigma, sigma_next, *, defer: bool = True):
        """(noise as fp32, norm): ``self.noise_sampler(sigma, sigma_next)``.  A sampler built by this package's noise chains can hand the
        sum back with its final normalisation still pending (``deferred``; ``norm`` = the decision on the device) for the step kernels to
        apply while they read the noise; ``defer=False`` (or any other sampler) gives the finished tensor and ``norm`` None."""
        lazy = getattr(self.noise_sampler, "deferred", None) if defer else None
        nz, norm = lazy(sigma, sigma_next) if lazy is not None else (self.noise_sampler(sigma, sigma_next), None)
        if norm is not None and nz.dtype != torch.float32:
            nz, norm = hip_lib.apply_norm_(utils.as_f32(nz), norm), None
        nz = utils.as_f32(nz)
        utils.pop_stats(nz)
        return nz, norm

    # ---- scalar bookkeeping
    @property
    def history_ratios(self):
        """py/sonar.py:208-219."""
        direction, hist = self.cfg.direction, self.cfg.momentum_hist
        return (hist, 1.0 + abs(direction) * (1 - hist) if direction < 0 else 2.0 - direction, direction)

    def check_step(self, step: int, *, is_history: bool = False):
        cfg = self.cfg
        if is_history and cfg.always_update_history:
            return True
        return cfg.momentum_start_step <= step <= cfg.momentum_end_step

    def kernel_cfg(self, step: int) -> hip_lib.MomentumCfg:
        """Everything the fused step kernels need to know about this step."""
        cfg = self.cfg
        kc = hip_lib.MomentumCfg()
        kc.momentum = cfg.momentum
        kc.hist_ratio, kc.hist_scale, kc.md_scale = self.history_ratios
        kc.mode = hip_lib.MODE_IDS[cfg.momentum_mode.name]
        kc.momentum_blend = hip_lib.BLEND_IDS[self.momentum_blend_name]
        kc.history_blend = hip_lib.BLEND_IDS[self.history_blend_name]
        kc.use_momentum = int(self.check_step(step))
        hist_ok = self.check_step(step, is_history=True)
        kc.update_hist = int(cfg.momentum_hist != 1 and hist_ok)
        kc.init_kind = 0
        if self.history_d is None and hist_ok and cfg.init in (HistoryType.SAMPLE, HistoryType.SAMPLE_NORM):
            kc.init_kind = hip_lib.INIT_IDS[cfg.init.name]
        kc.h_in_fresh = 0
        return kc

    def _rand_history(self, x: Tensor) -> Tensor:
        """py/sonar.py:192-204 (RAND init)."""
        cfg = self.cfg
        ns = noise.get_noise_sampler(cfg.rand_init_noise_type, x, None, None, seed=self.extra_args.get("seed"), cpu=True, normalized=True)
        hist = ns(None, None)
        if cfg.rand_init_noise_multiplier != 1:
            hip_lib.scale_noise_(hist, cfg.rand_init_noise_multiplier, False, None)
        return hist

    def _rand_init_due(self, step: int) -> bool:
        return self.history_d is None and self.cfg.init == HistoryType.RAND and self.check_step(step, is_history=True)

    def prefetch_rand_history(self, x: Tensor, step: int) -> None:
        """Draws the RAND history now if this step will create it.  The reference draws it inside momentum_step, i.e. BEFORE the
        step's ancestral noise (py/sonar.py:262-320, 541-573); a sampler that fetches its noise first (to fuse the add into the step
        kernel) calls this ahead of the fetch so that the global generator is consumed in the reference's order."""
        if self._rand_init_due(step) and getattr(self, "_rand_pre", None) is None:
            self._rand_pre = self._rand_history(x)

    def _history_for_kernel(self, x: Tensor, step: int, kc: hip_lib.MomentumCfg) -> Optional[Tensor]:
        """RAND init creates the history inside the step, after the denoised mix read 'no history'."""
        if self._rand_init_due(step):
            kc.h_in_fresh = 1
            pre, self._rand_pre = getattr(self, "_rand_pre", None), None
            return pre if pre is not None else self._rand_history(x)
        return self.history_d

    # ---- reference-signature building blocks (unfused; the samplers below use the fused kernels)
    def init_hist_d(self, x: Tensor, denoised: Tensor, sigma, *, step: int) -> None:
        """py/sonar.py:169-206."""
        if self.history_d is not None or not self.check_step(step, is_history=True):
            return
        cfg = self.cfg
        src = x if cfg.momentum_mode != MomentumMode.DENOISED else denoised
        if cfg.init == HistoryType.ZERO:
            self.history_d = None
        elif cfg.init == HistoryType.SAMPLE:
            self.history_d = src
        elif cfg.init == HistoryType.SAMPLE_NORM:
            self.history_d = hip_lib.div_scalar(utils.as_f32(src), float(sigma))
        elif cfg.init == HistoryType.RAND:
            self.history_d = self._rand_history(x)
        else:
            raise ValueError("Sonar sampler: bad history type")

    def update_hist(self, momentum_d: Tensor, step: int) -> None:
        """py/sonar.py:227-236."""
        hd, cfg = self.history_d, self.cfg
        if cfg.momentum_hist == 1 or not self.check_step(step, is_history=True):
            return
        ratio, hd_scale, md_scale = self.history_ratios
        self.history_d = momentum_d if hd is None else self.history_blend(hip_lib.mul_scalar(momentum_d, md_scale), hip_lib.mul_scalar(hd, hd_scale), ratio)

    def momentum_mix(self, history: Optional[Tensor], item: Tensor, sigma, *, is_denoised: bool = False, momentum=None) -> Tensor:
        """py/sonar.py:238-260."""
        momentum = self.cfg.momentum if momentum is None else momentum
        denoised_mode = self.cfg.momentum_mode == MomentumMode.DENOISED
        if momentum == 1 or history is None or denoised_mode != is_denoised:
            return item
        return self.momentum_blend(hip_lib.mul_scalar(history, float(sigma)) if is_denoised else history, item, momentum)

    def get_momentum_denoised(self, x: Tensor, denoised: Tensor, sigma, *, step: int, momentum=None, update_history=True) -> Tensor:
        """py/sonar.py:262-283."""
        mixed = self.momentum_mix(self.history_d, denoised, sigma, is_denoised=True, momentum=momentum)
        if update_history:
            self.init_hist_d(x, denoised, sigma, step=step)
            self.update_hist(hip_lib.div_scalar(utils.as_f32(denoised), float(sigma)), step=step)
        return mixed if self.check_step(step) else denoised

    def get_momentum_d(self, x: Tensor, denoised: Tensor, sigma, *, step: int, momentum=None, d: Optional[Tensor] = None,
                       update_history=True) -> Tensor:
        """py/sonar.py:285-307 (the blend weight is always cfg.momentum: ``momentum`` only gates the early-out)."""
        cfg = self.cfg
        gate = cfg.momentum if momentum is None else momentum
        d = to_d(x, sigma, denoised) if d is None else d
        if gate == 1 or cfg.momentum_mode == MomentumMode.DENOISED:
            return d
        momentum_d = self.momentum_mix(self.history_d, d, sigma)
        if update_history:
            self.init_hist_d(x, denoised, sigma, step=step)
            self.update_hist(d if cfg.momentum_mode == MomentumMode.NEW else momentum_d, step=step)
        return momentum_d if self.check_step(step) else d

    # ---- fused step
    def momentum_step(self, step: int, x: Tensor, denoised: Tensor, sigma, sigma_down, *, noise_add: Optional[Tensor] = None,
                      noise_scale: float = 0.0, noise_norm: Optional[Tensor] = None) -> Tensor:
        """py/sonar.py:309-320 as ONE kernel launch; optional fused ancestral noise add (:563-566)."""
        sigma_t, down_t = torch.as_tensor(sigma, dtype=torch.float32), torch.as_tensor(sigma_down, dtype=torch.float32)
        dt = (down_t.cpu() - sigma_t.cpu()).item()  # fp32 subtraction, like the reference's 0-d tensors
        kc = self.kernel_cfg(step)
        h_in = self._history_for_kernel(x, step, kc)
        x32, den32 = utils.as_f32(x), utils.as_f32(denoised)
        x_out, h_out = hip_lib.momentum_euler(x32, den32, h_in, kc, float(sigma_t), dt, noise=noise_add, noise_scale=noise_scale, noise_norm=noise_norm)
        self.history_d = h_out
        return x_out


class SonarGuidanceMixin:
    """py/sonar.py:323-411 (guidance is SURVEY.md §8f rank 1: statistics and blends run as HIP kernels)."""

    def __init__(self, cfg: Optional[GuidanceConfig] = None) -> None:
        self.guidance = cfg
        self.ref_latent = self.prepare_ref_latent(cfg.latent) if cfg and cfg.latent is not None else None

    @staticmethod
    def prepare_ref_latent(latent: Optional[Tensor]) -> Optional[Tensor]:
        if latent is None:
            return None
        if not latent.is_cuda:
            # reference latents usually arrive from the graph on CPU; statistics are a one-off setup step
            latent = latent.to("cuda")
        out = utils.as_f32(latent)
        inner = out.shape[-1] * out.shape[-2]
        rows = out.numel() // inner
        avg, std = hip_lib.rowstats(out, rows, inner)
        return hip_lib.row_affine(0, out, rows, inner, avg, std).to(latent.dtype)

    def guidance_step(self, step_index: int, x: Tensor, denoised: Tensor) -> Tensor:
        g = self.guidance
        if g is None or g.factor == 0.0 or not g.start_step <= step_index <= g.end_step:
            return x
        if self.ref_latent.device != x.device:
            self.ref_latent = self.ref_latent.to(device=x.device)
        if g.guidance_type == GuidanceType.LINEAR:
            return self.guidance_linear(x, self.ref_latent, g.factor, blend=self.guidance_blend)
        if g.guidance_type == GuidanceType.EULER:
            sigma, sigma_next = self.sigmas[step_index], self.sigmas[step_index + 1]
            return self.guidance_euler(sigma, sigma_next, x, denoised, self.ref_latent, g.factor)
        raise ValueError("Sonar: Guidance: Unknown guidance type")

    @classmethod
    def guidance_shift(cls, t: Tensor, ref_latent: Tensor, *, dim=None) -> Tensor:
        if dim is None:
            dim = tuple(range(-(t.ndim - 1), 0))
        dims = sorted({d % t.ndim for d in dim})
        t32, inverse = utils.dims_last(utils.as_f32(t), dims)  # any dim tuple: the statistics of a transposed copy
        inner = 1
        for d in range(t32.ndim - len(dims), t32.ndim):
            inner *= t32.shape[d]
        rows = t32.numel() // inner
        avg, std = hip_lib.rowstats(t32, rows, inner)
        ref = utils.as_f32(ref_latent.to(t.device))
        if ref.shape != t.shape:
            ref = ref.expand(t.shape)
        ref, _ = utils.dims_last(ref, dims)
        return utils.dims_restore(hip_lib.row_affine(1, ref, rows, inner, avg, std), inverse)

    @classmethod
    def guidance_euler(cls, sigma, sigma_next, x, denoised, ref_latent, factor: float = 0.2, *, do_shift: bool = True) -> Tensor:
        if float(sigma) == float(sigma_next):
            return cls.guidance_linear(x, ref_latent, factor=factor, do_shift=do_shift)
        shifted = cls.guidance_shift(denoised, ref_latent) if do_shift else ref_latent
        d = to_d(x, sigma, shifted)
        # the reference forms dt on fp32 0-d tensors: the difference rounds to fp32, then the product with the (fp32-cast) factor does
        f32 = torch.float32
        dt = (torch.as_tensor(float(sigma_next), dtype=f32) - torch.as_tensor(float(sigma), dtype=f32)) * torch.as_tensor(factor, dtype=f32)
        return hip_lib.axpby_(d, float(dt), utils.as_f32(x), 1.0)

    @classmethod
    def guidance_linear(cls, x, ref_latent, factor: float = 0.2, *, blend=None, do_shift: bool = True) -> Tensor:
        blend = utils.BLENDING_MODES["lerp"] if blend is None else blend
        shifted = cls.guidance_shift(x, ref_latent) if do_shift else ref_latent
        return blend(x, shifted, factor)


class SonarWithGuidance(SonarBase, SonarGuidanceMixin):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        SonarGuidanceMixin.__init__(self, self.cfg.guidance)


class SonarSampler(SonarWithGuidance):
    """py/sonar.py:419-450."""

    def __init__(self, model, sigmas: Tensor, s_in: Tensor, extra_args: dict, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.model = model
        self.sigmas = sigmas
        self.sigmas_host = sigmas.detach().to("cpu", torch.float32)  # one D2H copy per sampling run, not per step
        self.s_in = s_in
        self.extra_args = extra_args

    def call_model(self, x: Tensor, sigma, *args, s_in=None, extra_args=None) -> Tensor:
        s_in = self.s_in if s_in is None else s_in
        extra_args = self.extra_args if extra_args is None else self.extra_args | extra_args
        return self.model(x, sigma * s_in, *args, **extra_args)

    def guidance_active(self, step_index: int) -> bool:
        g = self.guidance
        return not (g is None or g.factor == 0.0 or not g.start_step <= step_index <= g.end_step)

    @classmethod
    def _run(cls, sonar, x, sigmas, callback, disable):
        dtype = x.dtype  # a half / bfloat16 latent is carried in fp32 between the steps and handed back in its own dtype
        for i in trange(len(sigmas) - 1, disable=disable):
            x, sigma, sigma_hat, denoised = sonar.step(i, x)
            if callback is not None:
                callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigma_hat, "denoised": denoised})
        return x if x.dtype == dtype else x.to(dtype)


class SonarEuler(SonarSampler):
    """py/sonar.py:452-526."""

    def step(self, step_index: int, sample: Tensor):
        sigma, sigma_next = self.sigmas[step_index], self.sigmas[step_index + 1]
        h_sigma, h_next = self.sigmas_host[step_index], self.sigmas_host[step_index + 1]
        denoised = self.call_model(sample, sigma)
        result = self.momentum_step(step_index, sample, denoised, h_sigma, h_next)
        if h_next > 0:
            result = self.guidance_step(step_index, result, denoised)
        return result, sigma, sigma, denoised

    @classmethod
    def sampler(cls, model, x: Tensor, sigmas: Tensor, extra_args: Optional[dict] = None, callback=None, disable=None,
                noise_sampler: Optional[Callable] = None, sonar_config: Optional[SonarConfig] = None,
                sonar_params: Optional[dict] = None) -> Tensor:
        sonar_config = cls.get_config(sonar_config, sonar_params)
        extra_args = {} if extra_args is None else extra_args
        sonar = cls(model, sigmas, x.new_ones((x.shape[0],)), extra_args, sonar_config)
        sonar.set_noise_sampler(x, sigmas, noise_sampler, seed=extra_args.get("seed"))
        return cls._run(sonar, x, sigmas, callback, disable)


class SonarEulerAncestral(SonarSampler):
    """py/sonar.py:529-623."""

    def __init__(self, eta: float = 1.0, s_noise: float = 1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.eta = eta
        self.s_noise = s_noise

    def step(self, step_index: int, sample: Tensor):
        sigma = self.sigmas[step_index]
        h_sigma, h_next = self.sigmas_host[step_index], self.sigmas_host[step_index + 1]
        sigma_down, sigma_up = get_ancestral_step(h_sigma, h_next, eta=self.eta)
        denoised = self.call_model(sample, sigma)
        add_noise = h_next > 0
        if add_noise and not self.guidance_active(step_index):
            # the noise add rides in the step kernel: x' = md*dt + x + noise*(s_noise*sigma_up)
            self.prefetch_rand_history(sample, step_index)
            nz, norm = self.draw_noise(sigma, self.sigmas[step_index + 1])
            scale = float(torch.as_tensor(self.s_noise * sigma_up, dtype=torch.float32))
            result = self.momentum_step(step_index, sample, denoised, h_sigma, sigma_down, noise_add=nz, noise_scale=scale, noise_norm=norm)
        else:
            result = self.momentum_step(step_index, sample, denoised, h_sigma, sigma_down)
            if add_noise:
                result = self.guidance_step(step_index, result, denoised)
                nz, _ = self.draw_noise(sigma, self.sigmas[step_index + 1], defer=False)
                result = hip_lib.axpby_(nz, float(torch.as_tensor(self.s_noise * sigma_up, dtype=torch.float32)), result, 1.0)
        return result, sigma, sigma, denoised

    @classmethod
    def sampler(cls, model, x, sigmas, extra_args=None, callback=None, disable=None, sonar_config: Optional[SonarConfig] = None,
                sonar_params: Optional[dict] = None, eta=1.0, s_noise=1.0, noise_sampler: Optional[Callable] = None):
        sonar_config = cls.get_config(sonar_config, sonar_params)
        extra_args = {} if extra_args is None else extra_args
        sonar = cls(eta, s_noise, model, sigmas, x.new_ones((x.shape[0],)), extra_args, sonar_config)
        sonar.set_noise_sampler(x, sigmas, noise_sampler, seed=extra_args.get("seed"))
        return cls._run(sonar, x, sigmas, callback, disable)


class SonarDPMPPSDE(SonarSampler):
    """py/sonar.py:626-820: DPM-Solver++(SDE), r = 1/2, with momentum on both half steps."""

    DEFAULT_NOISE_TYPE = noise.NoiseType.BROWNIAN

    def __init__(self, eta: float = 1.0, s_noise: float = 1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.eta = eta
        self.s_noise = s_noise

    @staticmethod
    def sigma_fn(t: Tensor):
        return t.neg().exp()

    @staticmethod
    def t_fn(sigma: Tensor):
        return sigma.log().neg()

    def momentum_step(self, step_index: int, x: Tensor, denoised: Tensor, sigma, sigma_next, sigma_down) -> Tensor:
        """Host scalars follow py/sonar.py:649-735 in fp32 0-d tensor arithmetic; tensors go through two fused kernels."""
        if sigma_next == 0:
            return super().momentum_step(step_index, x, denoised, sigma, sigma_down)
        cfg = self.cfg
        adjusted = cfg.momentum + (1 - cfg.momentum) / 2 if self.history_d is not None else cfg.momentum
        f32 = lambda v: float(torch.as_tensor(v, dtype=torch.float32))  # noqa: E731
        r = 1 / 2
        t, t_next = self.t_fn(sigma), self.t_fn(sigma_next)
        h = t_next - t
        s = t + h * r
        fac = 1 / (2 * r)
        s_t, s_s = self.sigma_fn(t), self.sigma_fn(s)
        sd, su = get_ancestral_step(s_t, s_s, self.eta)
        s_ = self.t_fn(sd)
        # ---- stage 1
        kc = self.kernel_cfg(step_index)
        h_in = self._history_for_kernel(x, step_index, kc)
        nz, norm = self.draw_noise(s_t, s_s)
        x32 = utils.as_f32(x)
        x_2, md1, hist = hip_lib.dpmpp_stage1(
            x32, utils.as_f32(denoised), h_in, kc, f32(sigma), f32((t - s_).expm1()), f32(self.sigma_fn(s_) / s_t), adjusted == 1,
            noise=nz, noise_scale=f32(self.s_noise * su), noise_norm=norm,
        )
        self.history_d = hist
        denoised_2 = self.call_model(x_2, s_s)
        # ---- stage 2
        s_t_next = self.sigma_fn(t_next)
        sd, su = get_ancestral_step(s_t, s_t_next, self.eta)
        t_down = self.t_fn(sd)
        kc2 = self.kernel_cfg(step_index)
        fuse_noise = not self.guidance_active(step_index)
        nz2, norm2 = self.draw_noise(s_t, s_t_next) if fuse_noise else (None, None)
        x_out, dd, hist = hip_lib.dpmpp_stage2(
            x32, utils.as_f32(denoised_2), md1, self.history_d, kc2, f32(s_s), f32((t - t_down).expm1()),
            f32(self.sigma_fn(t_down) / s_t), fac, adjusted == 1, noise=nz2, noise_scale=f32(self.s_noise * su), want_dd=not fuse_noise,
            noise_norm=norm2,
        )
        self.history_d = hist
        if not fuse_noise:
            x_out = self.guidance_step(step_index, x_out, dd)
            nz2, _ = self.draw_noise(s_t, s_t_next, defer=False)
            x_out = hip_lib.axpby_(nz2, f32(self.s_noise * su), x_out, 1.0)
        return x_out

    def step(self, step_index: int, sample: Tensor):
        sigma = self.sigmas[step_index]
        h_sigma, h_next = self.sigmas_host[step_index], self.sigmas_host[step_index + 1]
        sigma_down, _sigma_up = get_ancestral_step(h_sigma, h_next, eta=self.eta)
        denoised = self.call_model(sample, sigma)
        result = self.momentum_step(step_index, sample, denoised, h_sigma, h_next, sigma_down)
        return result, sigma, sigma, denoised

    @classmethod
    def sampler(cls, model, x: Tensor, sigmas: Tensor, extra_args: Optional[dict] = None, callback=None, disable=None,
                sonar_config: Optional[SonarConfig] = None, sonar_params: Optional[dict] = None, eta=1.0, s_noise=1.0,
                noise_sampler=None) -> Tensor:
        sonar_config = cls.get_config(sonar_config, sonar_params)
        extra_args = {} if extra_args is None else extra_args
        sonar = cls(eta, s_noise, model, sigmas, x.new_ones((x.shape[0],)), extra_args, sonar_config)
        sonar.set_noise_sampler(x, sigmas, noise_sampler, seed=extra_args.get("seed"))
        return cls._run(sonar, x, sigmas, callback, disable)


def add_samplers() -> None:
    """py/sonar.py:823-847: register the three samplers with ComfyUI's k-diffusion sampler table."""
    try:
        from comfy.samplers import KSampler, k_diffusion_sampling
    except ImportError:
        return  # not running inside ComfyUI (tests, bench)
    added = 0
    for name, fn in (("sonar_euler", SonarEuler.sampler), ("sonar_euler_ancestral", SonarEulerAncestral.sampler),
                     ("sonar_dpmpp_sde", SonarDPMPPSDE.sampler)):
        if name in KSampler.SAMPLERS:
            continue
        try:
            KSampler.SAMPLERS.append(name)
            setattr(k_diffusion_sampling, f"sample_{name}", fn)
            added += 1
        except ValueError as exc:
            print(f"Sonar: Failed to add {name} to built in samplers list: {exc}")
    if added > 0:
        importlib.reload(k_diffusion_sampling)
