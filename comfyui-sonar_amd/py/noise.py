"""Noise composition layer (API of the reference's ``py/noise.py``): items, chains, the sampler
wrapper, scheduled / composite / blended noise and the ``NoiseType`` registry.

Host code only builds closures; every per-call tensor operation is a HIP kernel launch on the
current stream (accumulate, mask-mix, blend, normalise).  Object model, argument names and the
order in which sub-samplers are *called* (it fixes the RNG draw order in replay mode) follow the reference.
"""
from __future__ import annotations

import abc
import math
from functools import partial
from typing import Callable, Optional

import torch
import yaml

from .. import hip_lib
from . import utils
from .noise_generation import *  # noqa: F401,F403  (the reference re-exports the generators from here)
from .noise_generation import (
    DeviceRNG,
    current_batch_offset,
    BrownianNoiseGenerator,
    CollatzNoiseGenerator,
    DistroNoiseGenerator,
    GaussianNoiseGenerator,
    GreenTestNoiseGenerator,
    HighresPyramidNoiseGenerator,
    LaplacianNoiseGenerator,
    MixedNoiseGenerator,
    NoiseType,
    OneFNoiseGenerator,
    PerlinOldNoiseGenerator,
    PinkOldNoiseGenerator,
    PowerLawNoiseGenerator,
    PowerOldNoiseGenerator,
    PyramidNoiseGenerator,
    PyramidOldNoiseGenerator,
    StudentTNoiseGenerator,
    UniformNoiseGenerator,
    VoronoiNoiseGenerator,
    WaveletFilteredNoiseGenerator,
    WaveletNoiseGenerator,
)
from .utils import fallback, pop_stats, scale_noise

Tensor = torch.Tensor

_CLONED_KEYS = frozenset(("custom_noise", "custom_noise_opt", "noise", "noise_opt", "sonar_custom_noise", "sonar_custom_noise_opt"))


def _accumulate(total: Optional[Tensor], part: Tensor) -> Tensor:
    """result.add_(noise) of py/noise.py:192-193 as one fused kernel."""
    if total is None:
        return part
    pop_stats(total)
    pop_stats(part)
    return hip_lib.axpby_(total, 1.0, part, 1.0)


class CustomNoiseItemBase(abc.ABC):
    """py/noise.py:30-80."""

    def __init__(self, factor, *, yaml_parameters=None, **kwargs):
        if yaml_parameters:
            extra = yaml.safe_load(yaml_parameters)
            if extra is not None:
                if not isinstance(extra, dict):
                    raise ValueError("CustomNoiseItem: yaml_parameters must either be null or an object")
                kwargs["ns_kwargs"] = extra
        self.factor = factor
        self.keys = set(kwargs.keys())
        for key, val in kwargs.items():
            setattr(self, key, val.clone() if key in _CLONED_KEYS and hasattr(val, "clone") else val)

    def clone_key(self, k):
        return getattr(self, k)

    def clone(self):
        return self.__class__(self.factor, **{k: self.clone_key(k) for k in self.keys})

    def set_factor(self, factor):
        self.factor = factor
        return self

    def get_normalize(self, k, default=None):
        val = getattr(self, k, None)
        return default if val is None else val

    @abc.abstractmethod
    def make_noise_sampler(self, x: Tensor, sigma_min=None, sigma_max=None, seed=None, cpu=True, normalized=True, **kwargs):
        raise NotImplementedError


class CustomNoiseItem(CustomNoiseItemBase):
    """py/noise.py:83-134: one registry noise type with a factor."""

    def __init__(self, factor, **kwargs):
        super().__init__(factor, **kwargs)
        if getattr(self, "noise_type", None) is None:
            raise ValueError("Noise type required!")

    @torch.no_grad()
    def make_noise_sampler(self, x: Tensor, sigma_min=None, sigma_max=None, seed=None, cpu=True, normalized=True, **kwargs):
        opts = getattr(self, "ns_kwargs", {}).copy()
        o_sigma, o_sigma_next, o_min, o_max = (opts.pop(k, None) for k in ("override_sigma", "override_sigma_next", "override_sigma_min", "override_sigma_max"))
        ns = get_noise_sampler(
            self.noise_type, x, fallback(o_min, sigma_min), fallback(o_max, sigma_max),
            seed=opts.pop("seed", seed), cpu=opts.pop("cpu", cpu), factor=self.factor,
            normalized=opts.pop("normalized", self.get_normalize("normalize", normalized)), **opts, **kwargs,
        )
        if o_sigma is None and o_sigma_next is None:
            return ns
        return lambda sigma, sigma_next: ns(fallback(o_sigma, sigma), fallback(o_sigma_next, sigma_next))


def _for_latent_dtype(x: Tensor, build: Callable) -> Callable:
    """The reference draws noise in the latent's dtype (``torch.randn(..., dtype=x.dtype)``); the kernels here compute in fp32.  A half /
    bfloat16 latent gets a sampler built for its fp32 image whose output is rounded to the latent's dtype once, at the end -- every value is
    the fp32 value correctly rounded, where the reference rounds after every operation."""
    if not torch.is_tensor(x) or x.dtype not in (torch.float16, torch.bfloat16):
        return build(x)
    inner = build(x.to(torch.float32))
    dtype = x.dtype

    def noise_sampler(sigma, sigma_next):
        out = inner(sigma, sigma_next)
        utils.pop_stats(out)
        return out.to(dtype)

    return noise_sampler


class CustomNoiseChain:
    """py/noise.py:137-196."""

    def __init__(self, items=None):
        self.items = items if items is not None else []

    def clone(self):
        return CustomNoiseChain([i.clone() for i in self.items])

    def add(self, item):
        if item is None:
            raise ValueError("Attempt to add nil item")
        self.items.append(item)

    @property
    def factor(self):
        return sum(abs(i.factor) for i in self.items)

    def rescaled(self, scale=1.0):
        divisor = self.factor / scale
        divisor = divisor if divisor != 0 else 1.0
        result = self.clone()
        if divisor != 1:
            for i in result.items:
                i.set_factor(i.factor / divisor)
        return result

    @torch.no_grad()
    def make_noise_sampler(self, x: Tensor, sigma_min=None, sigma_max=None, seed=None, cpu=True, normalized=True) -> Callable:
        if torch.is_tensor(x) and x.dtype in (torch.float16, torch.bfloat16):
            return _for_latent_dtype(x, lambda x32: self.make_noise_sampler(x32, sigma_min, sigma_max, seed=seed, cpu=cpu, normalized=normalized))
        samplers = tuple(i.make_noise_sampler(x, sigma_min, sigma_max, seed=seed, cpu=cpu, normalized=False) for i in self.items)
        if not samplers or not all(samplers):
            raise ValueError("Failed to get noise sampler")
        factor = self.factor

        def summed(sigma, sigma_next):
            """(tensor, final): final = True -> already scaled / normalised (the one-item fused path); else the chain's plain sum."""
            # result = sum_i item_i * factor_i (py/noise.py:188-194).  An item that would only multiply by its factor hands back
            # (raw tensor, factor) instead (`unscaled`), and the multiply rides in the accumulation kernel: y*a + x*b rounds each
            # product before the add, exactly like mul_ followed by add_, so the sweep is saved without changing a bit.
            if normalized and len(samplers) == 1:  # one item, factor 1: draw + normalise with a single write of the tensor
                fused = getattr(samplers[0], "normalized_call", None)
                out = fused(factor, sigma, sigma_next) if fused is not None else None
                if out is not None:
                    return out, True
            total, first = None, None
            pending = None  # the previous item's fold, captured for this item's kernel (hip_lib.FoldPrefix)
            for idx, ns in enumerate(samplers):
                fold = getattr(ns, "accumulate", None) if idx else None
                if idx == 0 and len(samplers) > 1 and getattr(samplers[1], "accepts_prefix", False):
                    # the first item too can be evaluated by the second item's kernel: the sum is then written once, by that kernel
                    capture = getattr(ns, "fold_prefix", None)
                    pending = capture(None, 1.0, sigma, sigma_next) if capture is not None else None
                    if pending is not None:
                        first = (pending.y, pending.first_factor)
                        continue
                if fold is not None:
                    # a generator that can fold its values into the running sum does so (read + write of the sum) instead of writing a
                    # tensor for the accumulation kernel to read back; same arithmetic, same bits
                    y, ymul = first if total is None else (total, 1.0)
                    last = idx == len(samplers) - 1
                    if pending is None and not last and getattr(samplers[idx + 1], "accepts_prefix", False):
                        # ... and when the NEXT item's kernel can evaluate this one on the fly, not even that: one pass for both
                        capture = getattr(ns, "fold_prefix", None)
                        pending = capture(y, ymul, sigma, sigma_next) if capture is not None else None
                        if pending is not None:
                            total = y
                            continue
                    partials = hip_lib.new_partials(y.device) if last else None
                    done = fold(y, ymul, partials, sigma, sigma_next, **({} if pending is None else {"pre": pending}))
                    if pending is not None:
                        pending.apply()  # nothing to do when the kernel hosted it
                        pending = None
                    if done:
                        total = y
                        if partials is not None:
                            utils.attach_stats(total, partials)
                        continue
                elif pending is not None:
                    pending.apply()
                    pending = None
                raw = getattr(ns, "unscaled", None)
                pair = raw(sigma, sigma_next) if raw is not None else None
                part, f = pair if pair is not None else (ns(sigma, sigma_next), 1.0)
                tag = pop_stats(part)
                if total is None and first is None:
                    first = (part, f)
                    if len(samplers) == 1 and f == 1.0 and tag is not None:
                        utils.attach_stats(part, tag)  # the only item, used as it is: its statistics still describe it
                    continue
                y, ymul = (first if total is None else (total, 1.0))
                if idx == len(samplers) - 1:  # the last accumulation also reduces the statistics a normalisation (here or a layer up) needs
                    total, partials = hip_lib.axpby_stats_(y, ymul, part, f)
                    utils.attach_stats(total, partials)
                else:
                    total = hip_lib.axpby_(y, ymul, part, f)
            if total is None:
                total = first[0] if first[1] == 1.0 else scale_noise(first[0], first[1], normalized=False)
            return total, False

        def noise_sampler(sigma, sigma_next):
            total, final = summed(sigma, sigma_next)
            if final or (not normalized and factor == 1):
                return total  # nothing left to do; a statistics tag of the sum (if any) stays valid for a normalising layer above
            return scale_noise(total, factor, normalized=normalized)

        def deferred(sigma, sigma_next):
            """(tensor, norm) for a consumer that can apply the final normalisation while it reads the noise (the Sonar sampler steps,
            ``noise_norm`` of sonar_momentum_euler_f32 / sonar_dpmpp_stage*_f32): the sum with its statistics already reduced comes
            back as it is and ``norm`` holds the decision on the device (``hip_lib.norm_decision``) -- the read + write of
            ``scale_noise`` is saved.  ``norm`` None: the tensor is final (every other case)."""
            total, final = summed(sigma, sigma_next)
            if final or (not normalized and factor == 1):
                return total, None
            partials = utils.pop_stats(total) if normalized else None
            if partials is None or total.dtype != torch.float32 or not total.is_contiguous():
                if partials is not None:
                    utils.attach_stats(total, partials)
                return scale_noise(total, factor, normalized=normalized), None
            return total, hip_lib.norm_decision(partials, total.numel(), factor)

        if all(getattr(ns, "plan_static", False) for ns in samplers):
            # every item draws on the device from nothing but the RNG position: the step is traced once and then issued by one foreign
            # call (hip_lib.Planned: the same entry points with the same arguments, the same bits)
            planned = _planned(noise_sampler)
            planned.deferred = _planned(deferred)
            planned.plan_static = True
            return planned
        noise_sampler.deferred = deferred
        return noise_sampler


def _planned(fn, *guards):
    """``fn`` behind a prepared plan keyed to this module's RNG bookkeeping and shard position."""
    return hip_lib.Planned(fn, take=DeviceRNG.take, rewind=DeviceRNG.rewind, guards=(current_batch_offset, *guards))


class NoiseSampler:
    """py/noise.py:199-257: wraps a generator factory; applies the item factor / normalisation."""

    def __init__(self, x: Tensor, sigma_min=None, sigma_max=None, seed=None, cpu: bool = False, transform: Callable = lambda t: t,
                 normalized=False, factor: float = 1.0, *, make_noise_sampler: Callable, **kwargs):
        self.factor = factor
        self.normalized = normalized
        self.transform = transform
        self.device = x.device
        self.dtype = x.dtype
        try:
            self.noise_sampler = make_noise_sampler(
                x,
                sigma_min=transform(torch.as_tensor(sigma_min)) if sigma_min is not None else None,
                sigma_max=transform(torch.as_tensor(sigma_max)) if sigma_max is not None else None,
                seed=seed, cpu=cpu, normalized=False, **kwargs,
            )
        except TypeError as exc:
            if "unexpected keyword" not in str(exc) and "positional argument" not in str(exc):
                raise
            self.noise_sampler = make_noise_sampler(x)
        self._planned = _planned(self._call, lambda: (self.factor, self.normalized)) if self.plan_static else None

    @property
    def plan_static(self) -> bool:
        """True when a call's launches depend on nothing but the RNG position (a device-mode generator that ignores its sigmas): the step
        may be replayed from a prepared plan (``hip_lib.Planned``)."""
        gen = self.noise_sampler
        return bool(getattr(gen, "PLAN_STATIC", False)) and getattr(gen, "cpu", True) is False and self.dtype == torch.float32

    @classmethod
    def simple(cls, f):
        return lambda *args, **kwargs: cls(*args, **kwargs, make_noise_sampler=lambda x, *_a, **_k: lambda _s, _sn: f(x))

    @classmethod
    def wrap(cls, f):
        return lambda *args, **kwargs: cls(*args, **kwargs, make_noise_sampler=f)

    def unscaled(self, *args):
        """(noise before the factor, factor) when this wrapper would only multiply (not normalised); None otherwise."""
        if self.normalized:
            return None
        args = tuple(self.transform(torch.as_tensor(s)) if s is not None else s for s in args)
        noise = self.noise_sampler(*args)
        if not hasattr(noise, "to") or noise.dtype != self.dtype or noise.device != self.device or noise.dtype != torch.float32:
            return scale_noise(noise, self.factor, normalized=False).to(dtype=self.dtype, device=self.device), 1.0
        return noise, float(self.factor)

    def accumulate(self, y, y_mul, partials, *args, pre=None) -> bool:
        """y <- y * y_mul + noise * factor in place, the noise never written out, when this wrapper would only multiply and its generator can
        fold (``generate_into``); ``partials`` (nullable) receives the statistics of the new y.  False: the caller takes the ordinary route
        (and ``pre``, the previous item's captured fold, is still the caller's to apply)."""
        if self.normalized or self.dtype != torch.float32 or y.dtype != torch.float32 or y.device != self.device or not y.is_contiguous():
            return False
        into = getattr(self.noise_sampler, "generate_into", None)
        if into is None:
            return False
        args = tuple(self.transform(torch.as_tensor(s)) if s is not None else s for s in args)
        return bool(into(y, float(y_mul), float(self.factor), partials, *args, **({} if pre is None else {"pre": pre})))

    @property
    def accepts_prefix(self) -> bool:
        """True when ``accumulate`` takes ``pre=`` (the previous chain item's fold, ``hip_lib.FoldPrefix``) and applies it in its own pass."""
        return (not self.normalized and self.dtype == torch.float32 and bool(getattr(self.noise_sampler, "accepts_prefix", False))
                and getattr(self.noise_sampler, "generate_into", None) is not None)

    def fold_prefix(self, y, y_mul, *args):
        """``accumulate`` captured instead of launched: the descriptor of y <- y * y_mul + noise * factor for the next item's kernel (the
        generator's keys are taken now, in chain order); None when this sampler cannot fold that way.  ``y`` None: this is the chain's
        first item -- the descriptor allocates the running sum (``pre.y``), which starts as the raw values (``pre.first_factor``: the
        multiplier the next fold applies to it)."""
        if self.normalized or self.dtype != torch.float32:
            return None
        if y is not None and (y.dtype != torch.float32 or y.device != self.device or not y.is_contiguous()):
            return None
        make = getattr(self.noise_sampler, "fold_prefix", None)
        if make is None:
            return None
        args = tuple(self.transform(torch.as_tensor(s)) if s is not None else s for s in args)
        if y is None:
            # the chain's first item: the sum starts as its RAW values and the next fold multiplies them by the factor (``unscaled``)
            pre = make(None, 1.0, 1.0, *args)
            if pre is not None:
                pre.first_factor = float(self.factor)
            return pre
        return make(y, float(y_mul), float(self.factor), *args)

    def normalized_call(self, factor, *args):
        """This sampler's output (its own factor must be 1, no normalisation of its own) followed by scale_noise(factor,
        normalized=True), through the generator's fused path; None when that does not apply.  Used by a single-item chain."""
        if self.normalized or self.factor != 1.0 or self.dtype != torch.float32:
            return None
        fused = getattr(self.noise_sampler, "generate_normalized", None)
        if fused is None:
            return None
        gen = self.noise_sampler
        if getattr(gen, "normalized", False):  # the generator would normalise first, then the chain again: keep that order
            return None
        args = tuple(self.transform(torch.as_tensor(s)) if s is not None else s for s in args)
        return fused(factor, *args)

    def deferred(self, *args):
        """(noise, norm) like ``CustomNoiseChain``'s ``deferred``: for a generator whose normalised form costs a separate pass over the
        tensor (pyramid), the raw draw comes back with the decision on the device and the consuming sampler step applies it while it reads
        the noise.  ``norm`` None: the tensor is final (every other generator)."""
        raw = getattr(self.noise_sampler, "generate_raw_stats", None) if self.normalized and self.dtype == torch.float32 else None
        if raw is not None:
            gen = self.noise_sampler
            gen.pre_hook()
            got = raw(*(self.transform(torch.as_tensor(s)) if s is not None else s for s in args))
            if got is not None and got[0].dtype == torch.float32 and got[0].device == self.device:
                noise, partials = got
                return noise, hip_lib.norm_decision(partials, noise.numel(), self.factor)
        return self(*args), None

    def __call__(self, *args, **kwargs):
        if self._planned is not None and not kwargs and len(args) == 2:
            return self._planned(*args)
        return self._call(*args, **kwargs)

    def _call(self, *args, **kwargs):
        args = tuple(self.transform(torch.as_tensor(s)) if s is not None else s for s in args)
        fused = getattr(self.noise_sampler, "generate_normalized", None) if self.normalized and not kwargs else None
        noise = fused(self.factor, *args) if fused is not None else None
        if noise is None:
            noise = self.noise_sampler(*args, **kwargs)
            noise = scale_noise(noise, self.factor, normalized=self.normalized)
        if hasattr(noise, "to") and (noise.dtype != self.dtype or noise.device != self.device):
            noise = noise.to(dtype=self.dtype, device=self.device)
        return noise


class AdvancedNoiseBase(CustomNoiseItemBase):
    """py/noise.py:260-283."""

    ns_factory_arg_keys = ()

    @property
    def ns_factory(self):
        raise NotImplementedError

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.ns_factory is None:
            raise NotImplementedError("ns_factory not implemented")
        ns_kwargs = {k: getattr(self, k) for k in self.ns_factory_arg_keys if getattr(self, k, None) is not None}
        self.sampler_factory = NoiseSampler.wrap(partial(self.ns_factory, **ns_kwargs))

    @torch.no_grad()
    def make_noise_sampler(self, *args, **kwargs):
        return self.sampler_factory(*args, factor=self.factor, **kwargs)


class AdvancedPyramidNoise(AdvancedNoiseBase):
    """py/noise.py:286-297."""

    ns_factory_arg_keys = ("discount", "iterations", "upscale_mode")
    pyramid_variants_map = {"pyramid": PyramidNoiseGenerator, "pyramid_old": PyramidOldNoiseGenerator, "highres_pyramid": HighresPyramidNoiseGenerator}

    @property
    def ns_factory(self):
        return self.pyramid_variants_map[self.variant]


class Advanced1fNoise(AdvancedNoiseBase):
    ns_factory_arg_keys = ("alpha", "hfac", "wfac", "k", "use_sqrt", "base_power")

    @property
    def ns_factory(self):
        return OneFNoiseGenerator


class AdvancedPowerLawNoise(AdvancedNoiseBase):
    ns_factory_arg_keys = ("alpha", "div_max_dims", "use_sign")

    @property
    def ns_factory(self):
        return PowerLawNoiseGenerator


# --------------------------------------------------------------------------------------------------
def _prep_mask(mask: Tensor, x: Tensor) -> Tensor:
    """py/noise.py:516-522: bilinear-resize the mask to the latent size, repeat to the batch."""
    m = utils.as_f32(mask.to(x.device, copy=True)).reshape((-1, 1, *mask.shape[-2:])).contiguous()
    if tuple(m.shape[-2:]) != tuple(x.shape[-2:]):
        m = utils.scale_samples(m, x.shape[-1], x.shape[-2], mode="bilinear")
    n, b = m.shape[0], x.shape[0]
    if n > b:
        m = m[:b]
    elif n < b:
        m = m.repeat(-(-b // n), 1, 1, 1)[:b]
    return m.contiguous()


class AdvancedWaveletNoise(AdvancedNoiseBase):
    """py/noise.py:392-444."""

    ns_factory_arg_keys = ("octave_scale_mode", "octave_rescale_mode", "post_octave_rescale_mode", "initial_amplitude", "persistence",
                           "octaves", "octave_height_factor", "octave_width_factor", "height_factor", "width_factor", "min_height",
                           "min_width", "update_blend", "update_blend_function")

    @property
    def ns_factory(self):
        return WaveletNoiseGenerator

    def clone_key(self, k):
        if k == "custom_noise" and getattr(self, "custom_noise", None) is not None:
            return self.custom_noise.clone()
        return super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        if x.ndim < 4:
            raise ValueError("Can only handle 4+ dimensional latents")
        height, width = x.shape[-2:]
        result = super().make_noise_sampler(x, *args, normalized=normalized, **kwargs)
        wavelet_ng = result.noise_sampler
        max_h = int(max(height, *(od.height for od in wavelet_ng.octave_data))) if wavelet_ng.octave_data else height
        max_w = int(max(width, *(od.width for od in wavelet_ng.octave_data))) if wavelet_ng.octave_data else width
        internal = None
        if getattr(self, "custom_noise", None) is not None:
            ref_x = x.new_zeros(*x.shape[:-2], max_h, max_w) if (max_w != width or max_h != height) else x
            internal = self.custom_noise.make_noise_sampler(ref_x, *args, normalized=self.normalize_noise, **kwargs)
        wavelet_ng.set_internal_noise_sampler(internal)
        return result


class GuidedNoise(CustomNoiseItemBase):
    """py/noise.py:536-623: a noise chain (or zeros) pulled towards a reference latent by the sampler's guidance arithmetic."""

    def __init__(self, factor, *, guidance_factor, ref_latent, method, normalize_noise, normalize_result, noise=None):
        super().__init__(factor, normalize_noise=normalize_noise, normalize_result=normalize_result, ref_latent=ref_latent.clone(),
                         noise=noise.clone() if noise is not None else None, method=method, guidance_factor=guidance_factor)

    def clone_key(self, k):
        if k == "noise" and self.noise is None:
            return None
        if k in {"noise", "ref_latent"}:
            return getattr(self, k).clone()
        return super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        from .sonar import SonarGuidanceMixin  # sonar imports this module

        factor, guidance_factor = self.factor, self.guidance_factor
        normalize_noise, normalize_result = (self.get_normalize(f"normalize_{k}", normalized) for k in ("noise", "result"))
        ns = None if self.noise is None else self.noise.make_noise_sampler(x, *args, normalized=normalize_noise, **kwargs)
        x_zeros = torch.zeros_like(x) if ns is None else None
        ref_latent = self.ref_latent.to(x, copy=True)
        if ref_latent.shape[-2:] != x.shape[-2:]:
            # F.interpolate(ref_latent, size=..., mode="bicubic", align_corners=True)  (py/noise.py:583-588)
            src = utils.as_f32(ref_latent).contiguous()
            sized = torch.empty((*src.shape[:-2], *x.shape[-2:]), dtype=torch.float32, device=src.device)
            ref_latent = hip_lib.resample_acc_(sized, src, 1.0, "bicubic_aligned", accumulate=False).to(ref_latent.dtype)
        ref_latent = ref_latent.contiguous()

        def base(s, sn):
            if ns is None:
                return x_zeros.clone()
            out = ns(s, sn)
            utils.pop_stats(out)
            return out

        if self.method == "linear":
            def noise_sampler(s, sn):
                return scale_noise(SonarGuidanceMixin.guidance_linear(base(s, sn), ref_latent, guidance_factor, do_shift=ns is not None),
                                   factor, normalized=normalize_result)
        elif self.method == "euler":
            def noise_sampler(s, sn):
                return scale_noise(SonarGuidanceMixin.guidance_euler(s, sn, base(s, sn), x, ref_latent, guidance_factor, do_shift=ns is not None),
                                   factor, normalized=normalize_result)
        else:
            raise ValueError("Bad method")
        return noise_sampler


class RepeatedNoise(CustomNoiseItemBase):
    """py/noise.py:681-760: a ring of cached noise tensors, re-served flipped / rolled / negated.  Host logic (the u32 draws come
    from a seeded CPU generator like the reference's); flips and rolls are index permutations done by torch on the device tensor,
    the arithmetic (negation, normalisation) by the HIP kernels.  Like the reference, a tensor served without a permutation is
    returned as a plain clone, without the factor."""

    def __init__(self, factor, *, noise, **kwargs):
        super().__init__(factor, noise=noise.clone(), **kwargs)

    def clone_key(self, k):
        return self.noise.clone() if k == "noise" else super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        factor, repeat_length, max_recycle, permute = self.factor, self.repeat_length, self.max_recycle, self.permute
        normalize = self.get_normalize("normalize", normalized)
        ns = self.noise.make_noise_sampler(x, *args, normalized=False, **kwargs)
        items: list = []
        u32_max = 0xFFFF_FFFF
        seed = kwargs.get("seed")
        if seed is None:
            seed = torch.randint(-u32_max, u32_max, (1,), device="cpu", dtype=torch.int64).item()
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        last_idx = -1

        def fresh(s, sn):
            t = ns(s, sn)
            pop_stats(t)
            return t

        def noise_sampler(s, sn):
            nonlocal last_idx
            rands = torch.randint(u32_max, (4,), generator=gen, dtype=torch.uint32).tolist()
            skip_permute = permute == "disabled"
            if len(items) < repeat_length:
                idx = len(items)
                noise = fresh(s, sn)
                items.append((1, noise))
                skip_permute = permute != "always"
            else:
                idx = rands[0] % repeat_length
                if idx == last_idx:
                    idx = (idx + 1) % repeat_length
                count, noise = items[idx]
                if count >= max_recycle:
                    noise = fresh(s, sn)
                    items[idx] = (1, noise)
                    skip_permute = permute != "always"
                else:
                    items[idx] = (count + 1, noise)
            last_idx = idx
            if skip_permute:
                return noise.clone()
            dims = noise.ndim
            if rands[1] % 2 == 0:
                if rands[2] <= u32_max // 5:
                    noise = noise.clone()
                    if rands[2] & 1 == 1:
                        hip_lib.mul_scalar(noise, -1.0, out=noise)
                else:
                    noise = torch.flip(noise, tuple({rands[2] % dims, rands[3] % dims}))
            else:
                dim = rands[2] % dims
                noise = torch.roll(noise, rands[3] % noise.shape[dim], dims=(dim,)).clone()
            return scale_noise(noise.contiguous(), factor, normalized=normalize)

        return noise_sampler


class RandomNoise(CustomNoiseItemBase):
    """py/noise.py:1022-1073: one (or the sum of ``mix_count`` distinct) randomly chosen item(s) of a chain per call; the choice comes
    from torch's global CPU generator like the reference's."""

    def __init__(self, factor, *, noise, mix_count, normalize):
        if len(noise.items) == 0:
            raise ValueError("RandomNoise requires ta least one noise item")
        super().__init__(factor, noise=noise.clone(), mix_count=mix_count, normalize=normalize)

    def clone_key(self, k):
        return self.noise.clone() if k == "noise" else super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        factor = self.factor
        samplers = tuple(ni.make_noise_sampler(x, *args, normalized=False, **kwargs) for ni in self.noise.items)
        count = len(samplers)
        mix_count = min(self.mix_count, count)
        normalize = self.get_normalize("normalize", normalized or mix_count > 1)

        def noise_sampler(s, sn):
            if mix_count == 1:
                return scale_noise(samplers[torch.randint(count, (1,)).item()](s, sn), factor, normalized=normalize)
            seen = set()
            while len(seen) < mix_count:
                seen.add(torch.randint(count, (1,)).item())
            idxs = tuple(seen)
            noise = samplers[idxs[0]](s, sn)
            for i in idxs[1:]:
                noise = _accumulate(noise, samplers[i](s, sn))
            return scale_noise(noise, factor, normalized=normalize)

        return noise_sampler


class ChannelNoise(CustomNoiseItemBase):
    """py/noise.py:1076-1131: a different chain item per channel (wrap / repeat / zero when there are fewer items than channels)."""

    def __init__(self, factor, *, noise, insufficient_channels_mode, normalize):
        if len(noise.items) == 0:
            raise ValueError("ChannelNoise requires at least one noise item")
        if insufficient_channels_mode not in {"wrap", "repeat", "zero"}:
            raise ValueError("Bad insufficient_channels_mode")
        super().__init__(factor, noise=noise.clone(), insufficient_channels_mode=insufficient_channels_mode, normalize=normalize)

    def clone_key(self, k):
        return self.noise.clone() if k == "noise" else super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        factor, mode = self.factor, self.insufficient_channels_mode
        c = x.shape[1]
        chosen = list(self.noise.items[:c])
        given = len(chosen)
        while len(chosen) < c:
            chosen.append(chosen[len(chosen) % given] if mode == "wrap" else chosen[given - 1] if mode == "repeat" else None)
        samplers = []
        for ch, item in enumerate(chosen):
            xs = x[:, ch:ch + 1, ...].contiguous()
            if item is None:
                samplers.append(lambda _s, _sn, xs=xs: torch.zeros_like(xs))
            else:
                samplers.append(item.make_noise_sampler(xs, *args, normalized=False, **kwargs))
        normalize = self.get_normalize("normalize", normalized)

        def noise_sampler(s, sn):
            noise = torch.cat(tuple(ns(s, sn) for ns in samplers), dim=1)  # channel interleave: a copy, no arithmetic
            return scale_noise(noise, factor, normalized=normalize)

        return noise_sampler


class LatentOperationFilteredNoise(CustomNoiseItemBase):
    """py/noise.py:1665-1698: a chain's noise passed through LATENT_OPERATIONs (each gets ``latent`` and the current ``sigma``)."""

    def clone_key(self, k):
        if k == "noise" and self.noise is not None:
            return self.noise.clone()
        return super().clone_key(k)

    def make_noise_sampler(self, x, sigma_min, sigma_max, *args, normalized=True, **kwargs):
        factor = self.factor
        normalize = self.get_normalize("normalize", normalized)
        ns = self.noise.make_noise_sampler(x, *args, sigma_min=sigma_min, sigma_max=sigma_max, normalized=self.normalize_noise, **kwargs)
        ops = self.operations

        def noise_sampler(sigma, sigma_next):
            noise = ns(sigma, sigma_next)
            for op in ops:
                pop_stats(noise)  # an operation may change the values in place
                noise = op(latent=noise, sigma=sigma)
            return scale_noise(noise.contiguous(), factor, normalized=normalize)

        return noise_sampler


class RippleFilteredNoise(CustomNoiseItemBase):
    """py/noise.py:1134-1202: the chain's noise times a sin / cos gain profile along one dimension (or the flattened trailing ones),
    rolled a little further every call."""

    def __init__(self, factor, *, noise, **kwargs):
        super().__init__(factor, noise=noise.clone(), **kwargs)

    def clone_key(self, k):
        return self.noise.clone() if k == "noise" else super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        factor = self.factor
        dim = self.dim + x.ndim if self.dim < 0 else self.dim
        if dim < 0 or dim >= x.ndim:
            raise ValueError("Dimension out of range")
        dim_els = math.prod(x.shape[dim:]) if self.flatten else x.shape[dim]
        inner = 1 if self.flatten else math.prod(x.shape[dim + 1:])
        mode_fun = torch.sin if self.mode.startswith("sin") else torch.cos
        follow_sign = self.mode.endswith("_copysign")
        # setup arithmetic on `dim_els` host values, once per sampler
        scaler = mode_fun(torch.linspace(self.offset, self.offset + math.pi * self.period, steps=dim_els, dtype=torch.float32))
        scaler = 1.0 + torch.where(scaler < 0, scaler * self.amplitude_low, scaler * self.amplitude_high)
        if self.flatten:  # torch's roll(dims=dim) on the broadcast-shaped scaler only moves the axis `dim`
            scaler = scaler.reshape(x.shape[dim], -1)
        ns = self.noise.make_noise_sampler(x, *args, normalized=self.normalize_noise, **kwargs)
        roll = self.roll
        normalize = self.get_normalize("normalize", normalized)
        counter = 0

        def noise_sampler(s, sn):
            nonlocal counter
            noise = ns(s, sn)
            to_roll = int(roll * counter)
            counter += 1
            table = scaler.roll(to_roll, dims=0).reshape(-1).contiguous().to(noise.device)
            result = scale_noise(noise, factor, normalized=normalize)
            return hip_lib.mul_table_(result.contiguous(), table, inner, follow_sign)

        return noise_sampler


class ResizedNoise(CustomNoiseItemBase):
    """py/noise.py:1410-1519: the inner chain runs at another latent size (a crop or a rescaled copy of x as its reference) and its noise is
    cropped / rescaled back -- ``scale_samples`` (the HIP resampler) and window indexing only."""

    def __init__(self, factor, *, custom_noise, **kwargs):
        if len(custom_noise.items) == 0:
            raise ValueError("ResizedNoise requires at least one noise item")
        super().__init__(factor, custom_noise=custom_noise.clone(), **kwargs)

    def clone_key(self, k):
        return self.custom_noise.clone() if k == "custom_noise" else super().clone_key(k)

    def working_size(self, xh: int, xw: int):
        """(rows, columns) the inner chain runs at, plus the crop offsets in latent cells.  `absolute` / `relative` sizes and the
        offsets are given in pixels (floor-divided by the compression), `percentage` sizes as fractions of the latent."""
        comp = self.spatial_compression
        if self.spatial_mode == "percentage":
            size = (max(1, int(xh * self.height)), max(1, int(xw * self.width)))
        elif self.spatial_mode in ("absolute", "relative"):
            base = (xh, xw) if self.spatial_mode == "relative" else (0, 0)
            size = (int(base[0] + self.height // comp), int(base[1] + self.width // comp))
        else:
            raise ValueError("Bad spatial_mode")
        return size, (self.crop_offset_vertical // comp, self.crop_offset_horizontal // comp)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        if x.ndim < 3:
            raise ValueError("ResizedNoise can only handle 3+ dimensional latents")
        factor, normalize = self.factor, self.get_normalize("normalize", normalized)
        xh, xw = x.shape[-2:]
        (nh, nw), (off_y, off_x) = self.working_size(xh, xw)
        if (nh, nw) == (xh, xw):  # nothing to resize: the inner chain normalises, this item only multiplies
            inner = self.custom_noise.make_noise_sampler(x, *args, normalized=normalize, **kwargs)
            return lambda *a, **k: scale_noise(inner(*a, **k), factor, normalized=False)

        def window(t, rows, cols):
            return utils.crop_samples(t, cols, rows, mode=self.crop_mode, offset_width=off_x, offset_height=off_y).contiguous()

        def resample(mode):
            return lambda t, rows, cols: utils.scale_samples(t, cols, rows, mode=mode)

        up, down = resample(self.upscale_mode), resample(self.downscale_mode)
        # route = (how the reference latent reaches the working size, how the noise comes back), chosen by which of the two is larger
        covers = (xh >= nh) + (xw >= nw)  # in how many axes the latent is at least the working size
        if covers == 2:
            to_work, back = (window if self.initial_reference == "prefer_crop" else down), up
        elif covers == 1:
            to_work, back = up, up
        else:
            to_work, back = up, (down if self.downscale_strategy == "scale" else window)
        inner = self.custom_noise.make_noise_sampler(to_work(x, nh, nw), *args, normalized=False, **kwargs)

        def noise_sampler(*a, **k):
            noise = scale_noise(inner(*a, **k), factor, normalized=normalize)
            pop_stats(noise)
            return back(noise.contiguous(), xh, xw)

        return noise_sampler


class NormalizeToScaleNoise(CustomNoiseItemBase):
    """py/noise.py:1205-1300: the chain's noise rescaled into a value range (simple: min / max over ``dims``; advanced: negatives and
    positives separately, over the whole tensor or per batch item), then the mean / std adjustments, then the usual scaling."""

    def __init__(self, factor, *, noise, min_negative_value: float, max_negative_value: float, min_positive_value: float, max_positive_value: float,
                 mode: str, **kwargs):
        if mode == "simple":
            if min_negative_value >= max_positive_value:
                raise ValueError("In simple mode, min_negative_value can't be greater or equal to max_positive_value")
        elif mode == "advanced":
            if min_negative_value >= max_negative_value:
                raise ValueError("In advanced mode, min_negative_value can't be greater or equal to max_negative value")
            if min_positive_value >= max_positive_value:
                raise ValueError("In advanced mode, min_positive_value can't be greater or equal to max_positive value")
        else:
            raise ValueError("Bad mode")
        super().__init__(factor, noise=noise.clone(), min_negative_value=min_negative_value, max_negative_value=max_negative_value,
                         min_positive_value=min_positive_value, max_positive_value=max_positive_value, mode=mode, **kwargs)

    def clone_key(self, k):
        return self.noise.clone() if k == "noise" else super().clone_key(k)

    @staticmethod
    def _trailing(noise: Tensor, dims, what: str):
        """(rows, inner) of a reduction over ``dims``, which must be the trailing dimensions (the node's '-3, -2, -1' and its suffixes)."""
        nd = noise.ndim
        want = sorted(d % nd for d in dims) if len(dims) else list(range(nd))
        if want != list(range(nd - len(want), nd)):
            raise hip_lib.SonarHipError(f"NormalizeToScaleNoise: {what} must be the trailing dimensions on the HIP path (got {tuple(dims)})")
        inner = math.prod(noise.shape[nd - len(want):])
        return noise.numel() // max(inner, 1), inner

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        std_dims, std_multiplier = self.std_dims, self.std_multiplier
        mean_dims, mean_multiplier = self.mean_dims, self.mean_multiplier
        factor, mode = self.factor, self.mode
        ns = self.noise.make_noise_sampler(x, *args, normalized=self.normalize_noise, **kwargs)
        normalize = self.get_normalize("normalize", normalized)

        def noise_sampler(s, sn):
            noise = ns(s, sn)
            pop_stats(noise)
            noise = utils.as_f32(noise).contiguous()
            if mode == "simple":
                noise = utils.normalize_to_scale(noise, self.min_negative_value, self.max_positive_value, dim=self.dims)
            else:
                # whole tensor (no dims, or fewer than two dimensions) or one batch item at a time (:1282-1286): rows of the same kernel
                rows = 1 if (noise.ndim < 2 or not self.dims) else noise.shape[0]
                noise = hip_lib.signed_rescale(noise, rows, noise.numel() // max(rows, 1), self.min_negative_value, self.max_negative_value,
                                               self.min_positive_value, self.max_positive_value)
            if mean_multiplier != 0:
                rows, inner = self._trailing(noise, mean_dims, "mean_dims")
                mean, _ = hip_lib.rowstats(noise, rows, inner)
                noise = hip_lib.row_affine(0, noise, rows, inner, mean * mean_multiplier, torch.ones_like(mean))  # (x - m k) / 1
            if std_multiplier != 0:
                rows, inner = self._trailing(noise, std_dims, "std_dims")
                _, std = hip_lib.rowstats(noise, rows, inner)
                adj = (std - 1.0) * std_multiplier + 1.0
                adj = torch.where(adj == 0, torch.full_like(adj, 1e-07), adj)
                noise = hip_lib.row_affine(0, noise, rows, inner, torch.zeros_like(adj), adj)
            return scale_noise(noise, factor, normalized=normalize)

        return noise_sampler


class PerDimNoise(CustomNoiseItemBase):
    """py/noise.py:1822-1893: noise assembled along one dimension from separate calls of the chain (slices / concatenation only)."""

    def clone_key(self, k):
        return self.noise.clone() if k == "noise" else super().clone_key(k)

    def make_noise_sampler(self, x, sigma_min, sigma_max, *args, normalized=True, **kwargs):
        factor = self.factor
        normalize = self.get_normalize("normalize", normalized)
        offset, chunk_size = self.offset, self.chunk_size
        dim = self.dim + x.ndim if self.dim < 0 else self.dim
        if dim < 0 or dim >= x.ndim:
            raise ValueError("Dimension out of range")
        dim_size = x.shape[dim]
        if self.shrink_dim:
            if offset + chunk_size > dim_size:
                raise ValueError("Offset or chunk size incompatible with tensor")
            x = x[tuple(slice(offset, offset + chunk_size) if d == dim else slice(None) for d in range(x.ndim))].contiguous()
        ns = self.noise.make_noise_sampler(x, *args, sigma_min=sigma_min, sigma_max=sigma_max, normalized=self.normalize_noise, **kwargs)
        trim = tuple(slice(-dim_size, None) if d == dim else slice(None) for d in range(x.ndim))
        if self.shrink_dim:
            def noise_sampler(sigma, sigma_next):
                noise = torch.cat(tuple(ns(sigma, sigma_next) for _ in range(dim_size)), dim=dim)[trim]
                return scale_noise(noise.contiguous(), factor, normalized=normalize)
        else:
            n_chunks = math.ceil(dim_size / chunk_size)
            temp_shape = list(x.shape)
            temp_shape[dim] = int(n_chunks * chunk_size)

            def noise_sampler(sigma, sigma_next):
                result = x.new_zeros(temp_shape)
                sel = [slice(None)] * x.ndim
                for idx in range(0, dim_size, chunk_size):
                    sel[dim] = slice(idx, idx + chunk_size)
                    result[tuple(sel)] = ns(sigma, sigma_next)[tuple(sel)]
                return scale_noise(result[trim].contiguous(), factor, normalized=normalize)

        return noise_sampler


class ModulatedNoise(CustomNoiseItemBase):
    """py/noise.py:762-1019: noise shaped by the local busyness of a reference latent (or of the sampler's x).  ``intensity`` and
    ``frequency`` run as HIP kernels (std over the modulation dims -> broadcast gain -> [LDS-resident rfft2 x boost x irfft2] ->
    L2-norm ratio mix; no scalar is read back).  ``spectral_signum``: forward transform over the modulation dims (LDS-resident rfft2
    and / or a channel DFT), log amplitude, per-sample quantiles by radix select, soft clamp of the outlying bins, inverse."""

    MODULATION_DIMS = (-3, (-2, -1), (-3, -2, -1))

    def __init__(self, factor, *, noise, normalize_result, normalize_noise, normalize_ref, modulation_type="none", modulation_strength=2.0,
                 modulation_dims=3, ref_latent_opt=None):
        super().__init__(factor, normalize_result=normalize_result, normalize_noise=normalize_noise, normalize_ref=normalize_ref,
                         noise=noise.clone(), modulation_dims=modulation_dims, modulation_type=modulation_type,
                         modulation_strength=modulation_strength, ref_latent_opt=None if ref_latent_opt is None else ref_latent_opt.clone())
        if modulation_type in {"intensity", "frequency", "spectral_signum"} and modulation_dims not in (1, 2, 3):
            raise ValueError("Bad modulation_dims")

    def clone_key(self, k):
        if k == "ref_latent_opt":
            return None if self.ref_latent_opt is None else self.ref_latent_opt.clone()
        if k == "noise":
            return self.noise.clone()
        return super().clone_key(k)

    def _spectral_signum_sampler(self, x, args, kwargs, factor, strength, normalize_noise, normalize_result, normalize_ref):
        from .sonar import get_ancestral_step

        ns = self.noise.make_noise_sampler(x, *args, normalized=normalize_noise, **kwargs)
        ref_latent = None if self.ref_latent_opt is None else self.ref_latent_opt.to(x, copy=True).contiguous()
        which = self.modulation_dims - 1

        def noise_sampler(s, sn):
            _down, sigma_up = get_ancestral_step(utils.tensor_item(s), utils.tensor_item(sn), eta=1.0)
            # the reference still normalises the (unused) first argument in place, sampler's x included (:1005-1008)
            pop_stats(scale_noise(x if ref_latent is None else ref_latent, normalized=normalize_ref))
            noise = ns(s, sn)
            pop_stats(noise)
            out = _spectral_signum(utils.as_f32(noise), float(sigma_up), strength, which)
            return scale_noise(out, factor, normalized=normalize_result)

        return noise_sampler

    @staticmethod
    def _frequency_boost(h: int, w: int, strength: float, device) -> Tensor:
        """Half-spectrum form of the magnitude boost 1 + (1 - exp(-((ky/h)^2 + (kx/w)^2) b^2)) (:838-848, unshifted index grid)."""
        from .noise_generation import _half_gain

        ky = (torch.arange(h, dtype=torch.float32)[:, None] / h) ** 2
        kx = (torch.arange(w, dtype=torch.float32)[None, :] / w) ** 2
        return _half_gain(2.0 - torch.exp(-(ky + kx) * abs(strength) ** 2)).to(device)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        from .sonar import get_ancestral_step  # sonar imports this module

        factor, strength, mtype = self.factor, float(self.modulation_strength), self.modulation_type
        normalize_noise, normalize_result, normalize_ref = (self.get_normalize(f"normalize_{k}", normalized) for k in ("noise", "result", "ref"))
        if mtype == "spectral_signum":
            return self._spectral_signum_sampler(x, args, kwargs, factor, strength, normalize_noise, normalize_result, normalize_ref)
        if mtype not in {"intensity", "frequency"}:
            ns = self.noise.make_noise_sampler(x, *args, normalized=normalize_result or normalize_noise, **kwargs)

            def plain_sampler(s, sn):
                return scale_noise(ns(s, sn), factor, normalized=False)

            return plain_sampler
        ns = self.noise.make_noise_sampler(x, *args, normalized=normalize_noise, **kwargs)
        ref_latent = None if self.ref_latent_opt is None else self.ref_latent_opt.to(x, copy=True).contiguous()
        if x.ndim < 3:
            raise ValueError("ModulatedNoise needs at least 3 dimensions")
        d3, d2, d1 = x.shape[-3:]
        outer, mid, inner = x.numel() // (d3 * d2 * d1), d3, d2 * d1
        which = self.modulation_dims - 1  # 0: dim -3, 1: dims (-2, -1), 2: dims (-3, -2, -1)
        boost = None
        if mtype == "frequency":
            if not hip_lib.power_supported(d2, d1):
                raise hip_lib.SonarHipError(f"ModulatedNoise frequency mode: plane {d2}x{d1} is beyond the spectral kernels (sides of at most 2048)")
            boost = self._frequency_boost(d2, d1, strength, x.device)

        def noise_sampler(s, sn):
            _down, sigma_up = get_ancestral_step(utils.tensor_item(s), utils.tensor_item(sn), eta=1.0)
            k = float(sigma_up)  # s_noise = 1
            # the reference normalises this tensor IN PLACE, sampler's x included (:1005-1008)
            ref = scale_noise(x if ref_latent is None else ref_latent, normalized=normalize_ref)
            pop_stats(ref)
            ref = utils.as_f32(ref).contiguous()
            # std is shift invariant: the reference's `x - x.mean()` only moves rounding
            if which == 0:
                stdv, bcast = hip_lib.std_mid(ref, outer, mid, inner), 2
            elif which == 1:
                stdv, bcast = hip_lib.rowstats(ref, outer * mid, inner)[1], 1
            else:
                stdv, bcast = hip_lib.rowstats(ref, outer, mid * inner)[1], 0
            noise = ns(s, sn)
            pop_stats(noise)
            noise = noise.contiguous()
            shaped, parts = hip_lib.bcast_gain(noise, stdv, outer, mid, inner, bcast, abs(strength), k)
            den = parts
            if boost is not None:
                den = hip_lib.new_partials(x.device)
                shaped = hip_lib.spectral_filter(shaped, boost, den)
                pop_stats(shaped)
            out = hip_lib.ratio_mix(shaped, strength, noise, k * (1.0 - strength), parts, k * k, den, out=shaped)
            return scale_noise(out, factor, normalized=normalize_result)

        return noise_sampler


def _spectral_signum(noise: Tensor, k: float, intensity: float, which: int, percentile: float = 5.0) -> Tensor:
    """py/noise.py:938-1015 (spectral_modulate_noise) on device.  ``which``: 0 = fftn over dim -3 (a channel DFT), 1 = over (-2, -1)
    (rfft2 per plane), 2 = over (-3, -2, -1) (rfft2, then the channel DFT on the half-spectrum).  The per-sample quantiles of
    |log amplitude| run over the FULL spectrum (the half-spectrum's dropped columns come back through their Hermitian partners)."""
    if noise.ndim != 4:
        raise hip_lib.SonarHipError("ModulatedNoise spectral_signum: [B, C, H, W] latents on the HIP path")
    b, c, h, w = noise.shape
    add = hip_lib.mul_scalar(noise.contiguous(), k)  # additive_noise = noise * s_noise * sigma_up
    if which == 0:
        z = hip_lib.cdft_mid(add, b, c, h * w, inverse=False)          # complex [B, C, H, W]
        la, full = hip_lib.spectral_logamp(z.reshape(b * c, h, w), b * c, 1, h, w)
        gain, plane_elems = 1.0 / c, h * w
    else:
        if not hip_lib.power_supported(h, w):
            raise hip_lib.SonarHipError(f"ModulatedNoise spectral_signum: plane {h}x{w} is beyond the transform kernels (lines of at most 2048)")
        z = hip_lib.rfft2(add)                                          # complex [B, C, H, W/2+1]; any plane size
        wz = w // 2 + 1
        if which == 2:
            z = hip_lib.cdft_mid(z, b, c, h * wz, inverse=False)
        la, full = hip_lib.spectral_logamp(z.reshape(b * c, h, wz), b * c, c if which == 2 else 1, h, w)
        # torch's ifftn divides by the transformed element count; the inverse kernel below is ortho (1 / sqrt(H W))
        gain, plane_elems = 1.0 / math.sqrt(h * w) / (c if which == 2 else 1), h * wz
    # torch.quantile(|log_amp|.flatten(1), q, dim=1): one row per SAMPLE
    q = torch.stack([hip_lib.abs_quantile_rows(full, b, c * h * w, qq) for qq in (percentile * 0.01, 1 - percentile * 0.01, 1.0)], dim=1).contiguous()
    if b != 1 and b != c:
        # the reference expands the [B] quantile vector as [B, 1, 1] against [B, C, H, W]: it only lines up for B = 1 or B = C
        raise RuntimeError(f"The expanded size of the tensor ({c}) must match the existing size ({b}) at non-singleton dimension 1.  "
                           f"Target sizes: [{b}, {c}, {h}, {w}].  Tensor sizes: [{b}, 1, 1]")
    hip_lib.spectral_signum_mask_(z, la, q[:1].contiguous() if b == 1 else q, b * c, c, plane_elems, intensity, gain, channel_sym=which == 2)
    if which == 0:
        return hip_lib.cdft_mid(z, b, c, h * w, inverse=True, real_out=True)
    if which == 2:
        z = hip_lib.cdft_mid(z, b, c, h * (w // 2 + 1), inverse=True)
    ones = torch.ones(h, w // 2 + 1, dtype=torch.float32, device=noise.device)
    return hip_lib.power_irfft2(z, ones, (b, c, h, w))


class CompositeNoise(CustomNoiseItemBase):
    """py/noise.py:470-533: dst*(1-mask) + src*mask; dst is sampled before src."""

    def __init__(self, factor, *, dst_noise, src_noise, normalize_dst, normalize_src, normalize_result, mask):
        super().__init__(factor, dst_noise=dst_noise.clone(), src_noise=src_noise.clone(), normalize_dst=normalize_dst,
                         normalize_src=normalize_src, normalize_result=normalize_result, mask=mask.clone())

    def clone_key(self, k):
        return getattr(self, k).clone() if k in {"mask", "src_noise", "dst_noise"} else super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        n_src, n_dst, n_res = (self.get_normalize(f"normalize_{k}", normalized) for k in ("src", "dst", "result"))
        ns_dst = self.dst_noise.make_noise_sampler(x, *args, normalized=n_dst, **kwargs)
        ns_src = self.src_noise.make_noise_sampler(x, *args, normalized=n_src, **kwargs)
        mask = _prep_mask(self.mask, x)  # [B,1,H,W]
        b, c = x.shape[:2]
        # the kernel indexes the mask as flat[i % mask_n]: expand over channels once, at setup
        mask_full = mask.expand(b, c, *mask.shape[-2:]).contiguous() if (b > 1 and c > 1) else mask
        factor = self.factor

        def noise_sampler(s, sn):
            dst = ns_dst(s, sn)
            src = ns_src(s, sn)
            pop_stats(dst)
            pop_stats(src)
            mixed = hip_lib.mask_mix(dst, src, mask_full, out=dst)
            return scale_noise(mixed, factor, normalized=n_res)

        return noise_sampler


class ScheduledNoise(CustomNoiseItemBase):
    """py/noise.py:626-678: the wrapped noise inside [end_sigma, start_sigma], the fallback (or zeros) outside."""

    def __init__(self, factor, *, noise, start_sigma, end_sigma, normalize, fallback_noise=None):
        super().__init__(factor, noise=noise.clone(), start_sigma=start_sigma, end_sigma=end_sigma, normalize=normalize,
                         fallback_noise=None if fallback_noise is None else fallback_noise.clone())

    def clone_key(self, k):
        if k == "noise":
            return self.noise.clone()
        if k == "fallback_noise":
            return None if self.fallback_noise is None else self.fallback_noise.clone()
        return super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        factor, start_sigma, end_sigma = self.factor, self.start_sigma, self.end_sigma
        normalize = self.get_normalize("normalize", normalized)
        ns = self.noise.make_noise_sampler(x, *args, normalized=False, **kwargs)
        if self.fallback_noise:
            ns_out = self.fallback_noise.make_noise_sampler(x, *args, normalized=False, **kwargs)
        else:
            def ns_out(_s, _sn):  # zeros; with normalisation on this yields NaN exactly like the reference
                return torch.zeros_like(x)

        def noise_sampler(s, sn):
            if s is None or sn is None:
                raise ValueError("ScheduledNoise requires sigma, sigma_next to be passed")
            inside = end_sigma <= float(s) <= start_sigma
            return scale_noise((ns if inside else ns_out)(s, sn), factor, normalized=normalize)

        return noise_sampler


class BlendedNoise(CustomNoiseItemBase):
    """py/noise.py:1302-1407."""

    def __init__(self, factor, *, normalize, blend_function, custom_noise_1=None, custom_noise_2=None, custom_noise_mask=None,
                 noise_2_percent=0.5):
        if custom_noise_1 is None and (custom_noise_mask is not None or noise_2_percent != 1):
            raise ValueError("When custom_noise_1 is not attached noise_2_percent must be set to 1")
        if custom_noise_2 is None and (custom_noise_mask is not None or noise_2_percent != 0):
            raise ValueError("When custom_noise_2 is not attached noise_2_percent must be set to 0")
        if custom_noise_mask is None and noise_2_percent == 1 and custom_noise_1 is None:
            custom_noise_1, custom_noise_2, noise_2_percent = custom_noise_2, None, 0.0
        super().__init__(factor, noise_2_percent=noise_2_percent, blend_function=blend_function, custom_noise_1=custom_noise_1.clone(),
                         custom_noise_2=None if custom_noise_2 is None else custom_noise_2.clone(),
                         custom_noise_mask=None if custom_noise_mask is None else custom_noise_mask.clone(), normalize=normalize)

    def clone_key(self, k):
        if k in {"custom_noise_1", "custom_noise_2", "custom_noise_mask"}:
            v = getattr(self, k)
            return None if v is None else v.clone()
        return super().clone_key(k)

    def make_noise_sampler(self, x, *args, normalized=True, **kwargs):
        factor, pct, blend_function = self.factor, self.noise_2_percent, self.blend_function
        normalize = self.get_normalize("normalize", normalized)

        def sub(chain):
            return None if chain is None else chain.make_noise_sampler(x, *args, normalized=False, **kwargs)

        ns_1, ns_2, ns_mask = sub(self.custom_noise_1), sub(self.custom_noise_2), sub(self.custom_noise_mask)

        def noise_sampler(s, sn):
            n1 = ns_1(s, sn)
            n2 = None if ns_2 is None else ns_2(s, sn)
            weight = pct
            if ns_mask is not None:
                weight = (utils.normalize_to_scale(ns_mask(s, sn), 0.0, 1.0) + pct).clamp_(0.0, 1.0)
            if n2 is not None:
                pop_stats(n1)
                n1 = blend_function(n1, n2, weight)
            return scale_noise(n1, factor, normalized=normalize)

        return noise_sampler


class WaveletFilteredNoise(CustomNoiseItemBase):
    """py/noise.py:1521-1593."""

    def clone_key(self, k):
        if k in {"noise", "noise_high"} and getattr(self, k) is not None:
            return getattr(self, k).clone()
        return super().clone_key(k)

    def make_noise_sampler(self, x, sigma_min, sigma_max, *args, normalized=True, **kwargs):
        factor = self.factor
        normalize = self.get_normalize("normalize", normalized)

        def sub(chain):
            if chain is None:
                return None
            return chain.make_noise_sampler(x, *args, sigma_min=sigma_min, sigma_max=sigma_max, normalized=self.normalize_noise, **kwargs)

        ns_low, ns_high = sub(self.noise), sub(getattr(self, "noise_high", None))
        opts = getattr(self, "ns_kwargs", {}).copy()
        blends = {}
        for key in ("yl_blend_function", "yh_blend_function"):
            fn = opts.pop(key, utils.BLENDING_MODES["lerp"])
            blends[key] = utils.BLENDING_MODES[fn] if isinstance(fn, str) else fn
        gen = WaveletFilteredNoiseGenerator(x, *args, sigma_min=sigma_min, sigma_max=sigma_max, normalized=False, noise_sampler=ns_low,
                                            noise_sampler_high=ns_high, **blends, **(kwargs | opts))
        return lambda sigma, sigma_next: scale_noise(gen(sigma, sigma_next), factor, normalized=normalize)


# --------------------------------------------------------------------------------------------------
def _scaled(f: float):
    def fn(t):
        pop_stats(t)
        return hip_lib.scale_noise_(t, f, False, None)
    fn.scale_factor = f  # a pure multiply: MixedNoiseGenerator folds it into its accumulation kernel
    return fn


def _mix(name, *entries, output=None):
    return NoiseSampler.wrap(partial(MixedNoiseGenerator, name=name, noise_mix=tuple(entries), output_fun=output))


def _pyramid_mix(name, mode=None, discount=0.6):
    extra = {} if mode is None else {"upscale_mode": mode}
    return _mix(name, (PyramidNoiseGenerator, {"discount": discount, **extra}, _scaled(0.2)),
                (PyramidNoiseGenerator, {"discount": discount, **extra}, _scaled(-0.8)))


# py/noise.py:2244-2457 — NoiseType -> factory (presets are API)
NOISE_SAMPLERS: dict[NoiseType, Callable] = {
    NoiseType.BROWNIAN: NoiseSampler.wrap(BrownianNoiseGenerator),
    NoiseType.DISTRO: NoiseSampler.wrap(DistroNoiseGenerator),
    NoiseType.GAUSSIAN: NoiseSampler.wrap(GaussianNoiseGenerator),
    NoiseType.UNIFORM: NoiseSampler.wrap(UniformNoiseGenerator),
    NoiseType.PERLIN: NoiseSampler.wrap(PerlinOldNoiseGenerator),
    NoiseType.STUDENTT: NoiseSampler.wrap(StudentTNoiseGenerator),
    NoiseType.ONEF_PINKISH: NoiseSampler.wrap(partial(OneFNoiseGenerator, alpha=-0.5)),
    NoiseType.ONEF_GREENISH: NoiseSampler.wrap(partial(OneFNoiseGenerator, alpha=0.5)),
    NoiseType.ONEF_PINKISHGREENISH: _mix("onef_pinkishgreenish", (OneFNoiseGenerator, {"alpha": 0.5}, None),
                                         (OneFNoiseGenerator, {"alpha": -0.5}, None), output=_scaled(0.5)),
    NoiseType.ONEF_PINKISH_MIX: _mix("onef_pinkish_mix", (OneFNoiseGenerator, {"alpha": -0.5}, _scaled(-1.0)),
                                     (OneFNoiseGenerator, {"alpha": -0.5}, None), output=_scaled(0.5)),
    NoiseType.ONEF_GREENISH_MIX: _mix("onef_greenish_mix", (OneFNoiseGenerator, {"alpha": 0.5}, _scaled(-1.0)),
                                      (OneFNoiseGenerator, {"alpha": 0.5}, None), output=_scaled(0.5)),
    NoiseType.WHITE: NoiseSampler.wrap(partial(PowerLawNoiseGenerator, alpha=0.0, use_sign=True)),
    NoiseType.GREY: NoiseSampler.wrap(partial(PowerLawNoiseGenerator, alpha=0.0)),
    NoiseType.VELVET: NoiseSampler.wrap(partial(PowerLawNoiseGenerator, alpha=1.0, use_sign=True, div_max_dims=(-3, -2, -1))),
    NoiseType.VIOLET: NoiseSampler.wrap(partial(PowerLawNoiseGenerator, alpha=0.5, use_sign=True, div_max_dims=(-3, -2, -1))),
    NoiseType.PINK_OLD: NoiseSampler.wrap(PinkOldNoiseGenerator),
    NoiseType.LAPLACIAN: NoiseSampler.wrap(LaplacianNoiseGenerator),
    NoiseType.HIGHRES_PYRAMID: NoiseSampler.wrap(HighresPyramidNoiseGenerator),
    NoiseType.PYRAMID: NoiseSampler.wrap(PyramidNoiseGenerator),
    NoiseType.RAINBOW_MILD: _mix("rainbow_mild", (GreenTestNoiseGenerator, {}, _scaled(0.55)), (GreenTestNoiseGenerator, {}, _scaled(0.7)),
                                 output=_scaled(1.15)),
    NoiseType.RAINBOW_INTENSE: _mix("rainbow_intense", (GreenTestNoiseGenerator, {}, _scaled(0.75)), (GreenTestNoiseGenerator, {}, _scaled(0.5)),
                                    output=_scaled(1.15)),
    NoiseType.GREEN_TEST: NoiseSampler.wrap(GreenTestNoiseGenerator),
    NoiseType.POWER_OLD: NoiseSampler.wrap(PowerOldNoiseGenerator),
    NoiseType.COLLATZ: NoiseSampler.wrap(CollatzNoiseGenerator),
    NoiseType.PYRAMID_OLD: NoiseSampler.wrap(PyramidOldNoiseGenerator),
    NoiseType.PYRAMID_BISLERP: NoiseSampler.wrap(partial(PyramidNoiseGenerator, upscale_mode="bislerp")),
    NoiseType.HIGHRES_PYRAMID_BISLERP: NoiseSampler.wrap(partial(HighresPyramidNoiseGenerator, upscale_mode="bislerp")),
    NoiseType.PYRAMID_AREA: NoiseSampler.wrap(partial(PyramidNoiseGenerator, upscale_mode="area")),
    NoiseType.HIGHRES_PYRAMID_AREA: NoiseSampler.wrap(partial(HighresPyramidNoiseGenerator, upscale_mode="area")),
    NoiseType.PYRAMID_OLD_BISLERP: NoiseSampler.wrap(partial(PyramidOldNoiseGenerator, upscale_mode="bislerp")),
    NoiseType.PYRAMID_OLD_AREA: NoiseSampler.wrap(partial(PyramidOldNoiseGenerator, upscale_mode="area")),
    NoiseType.PYRAMID_DISCOUNT5: NoiseSampler.wrap(partial(PyramidNoiseGenerator, discount=0.5)),
    NoiseType.PYRAMID_MIX: _pyramid_mix("pyramid_mix"),
    NoiseType.PYRAMID_MIX_BISLERP: _pyramid_mix("pyramid_mix_bislerp", "bislerp", 0.5),
    NoiseType.PYRAMID_MIX_AREA: _pyramid_mix("pyramid_mix_area", "area", 0.5),
    NoiseType.WAVELET: NoiseSampler.wrap(WaveletNoiseGenerator),
    NoiseType.VORONOI_FUZZ: NoiseSampler.wrap(VoronoiNoiseGenerator),
    NoiseType.VORONOI_MIX: NoiseSampler.wrap(VoronoiNoiseGenerator),
}


def get_noise_sampler(noise_type, x: Tensor, sigma_min, sigma_max, seed=None, cpu: bool = True, factor: float = 1.0,
                      normalized=False, **kwargs) -> Callable:
    """py/noise.py:2460-2489."""
    if noise_type is None:
        noise_type = NoiseType.GAUSSIAN
    elif isinstance(noise_type, str):
        noise_type = NoiseType[noise_type.upper()]
    if noise_type == NoiseType.BROWNIAN and (sigma_min is None or sigma_max is None):
        raise ValueError("Must pass sigma min/max when using brownian noise")
    factory = NOISE_SAMPLERS.get(noise_type)
    if factory is None:
        raise ValueError("Unknown noise sampler")
    return _for_latent_dtype(x, lambda x32: factory(x32, sigma_min, sigma_max, seed=seed, cpu=cpu, factor=factor, normalized=normalized, **kwargs))
