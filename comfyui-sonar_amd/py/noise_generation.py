"""Noise generators of the Sonar hot path on MI355X (API of the reference's ``py/noise_generation.py``).

Two RNG modes, selected by the reference's own ``cpu`` flag:
  * ``cpu=True``  (reference default, *replay*): the base random draws come from the torch CPU
    generator in exactly the reference's order and are copied to the device; all arithmetic after the
    draw runs in HIP kernels.  Results match the reference's CPU path within fp32 tolerance.
  * ``cpu=False`` (*generate*): draws come from the in-kernel Philox4x32-10 counter RNG, keyed by
    (seed, stream, global element index), fused with the generator arithmetic and with the
    normaliser's statistics.  Values are independent of how a batch is sharded over GPUs.
"""
from __future__ import annotations

import math
import os
import threading
from enum import Enum, auto
from typing import NamedTuple, Optional

import torch

from .. import hip_lib
from . import utils
from .utils import attach_stats, fallback, scale_noise, tensor_to

Tensor = torch.Tensor


class NoiseType(Enum):
    """py/noise_generation.py:31-80 (names are API: node dropdowns and YAML use them)."""

    BROWNIAN = auto()
    COLLATZ = auto()
    DISTRO = auto()
    GAUSSIAN = auto()
    GREEN_TEST = auto()
    GREY = auto()
    HIGHRES_PYRAMID = auto()
    HIGHRES_PYRAMID_AREA = auto()
    HIGHRES_PYRAMID_BISLERP = auto()
    LAPLACIAN = auto()
    ONEF_GREENISH = auto()
    ONEF_GREENISH_MIX = auto()
    ONEF_PINKISH = auto()
    ONEF_PINKISH_MIX = auto()
    ONEF_PINKISHGREENISH = auto()
    PERLIN = auto()
    PINK_OLD = auto()
    POWER_OLD = auto()
    PYRAMID = auto()
    PYRAMID_AREA = auto()
    PYRAMID_BISLERP = auto()
    PYRAMID_DISCOUNT5 = auto()
    PYRAMID_MIX = auto()
    PYRAMID_MIX_AREA = auto()
    PYRAMID_MIX_BISLERP = auto()
    PYRAMID_OLD = auto()
    PYRAMID_OLD_AREA = auto()
    PYRAMID_OLD_BISLERP = auto()
    RAINBOW_INTENSE = auto()
    RAINBOW_MILD = auto()
    STUDENTT = auto()
    UNIFORM = auto()
    VELVET = auto()
    VIOLET = auto()
    VORONOI_FUZZ = auto()
    VORONOI_MIX = auto()
    WAVELET = auto()
    WHITE = auto()

    @classmethod
    def get_names(cls, default=GAUSSIAN, skip=None):
        if default is not None:
            if isinstance(default, int):
                default = cls(default)
            yield default.name.lower()
        for nt in cls:
            if nt == default or (skip and nt in skip):
                continue
            yield nt.name.lower()


class NoiseError(Exception):
    pass


# --------------------------------------------------------------------------------------------------
# on-device RNG bookkeeping (generate mode)
class DeviceRNG:
    """Hands out Philox stream ids from the state of torch's default generator of the current ROCm device -- the generator the
    reference's ``cpu=False`` draws consume.  (seed, stream) = (its seed, its Philox offset / 4); taking ``count`` streams advances
    the offset like a draw would, without launching anything.  So ``torch.manual_seed(s)`` (every call of it, same value or not),
    ``torch.cuda.set_rng_state`` and ``torch.cuda.manual_seed`` rewind device-generated noise exactly as they rewind the
    reference's, and ranks that seed alike stay in step (shard invariance)."""

    _lock = threading.Lock()

    @classmethod
    def take(cls, count: int = 1) -> tuple[int, int]:
        gens = torch.cuda.default_generators
        if not gens:  # the runtime is not initialised yet: the wrapper's lazy initialisation fills the table
            torch.cuda.current_device()
            gens = torch.cuda.default_generators
        gen = gens[hip_lib._CUR_DEVICE()]
        with cls._lock:
            offset = gen.get_offset()
            gen.set_offset(offset + 4 * int(count))  # Philox offsets move in units of 4
        seed = gen.initial_seed()
        rec = hip_lib._recorder
        if rec is not None and rec.thread == threading.get_ident():  # this call is being traced into a prepared plan: its stream ids are relative to this position
            rec.on_take(seed, offset // 4, int(count))
        return seed, offset // 4

    @classmethod
    def rewind(cls, stream: int, count: int | None = None) -> bool:
        """Put the position back to ``stream`` (a prepared plan that took ``count`` streams there and then could not issue the step).
        With ``count`` given the position only moves back if nobody took streams since -- it still reads ``stream + count``: another
        sampler on another thread may have drawn in between, and handing ITS ids out again would make two samplers draw the same
        noise, whereas a gap of unused ids costs nothing.  Returns whether the position moved."""
        gen = torch.cuda.default_generators[torch.cuda.current_device()]
        with cls._lock:
            if count is not None and gen.get_offset() != 4 * (int(stream) + int(count)):
                return False
            gen.set_offset(4 * int(stream))
            return True


class _Shard(threading.local):
    batch_offset = 0


_SHARD = _Shard()


class shard_offset:
    """Context manager: the latents generated inside are latents [batch_offset, batch_offset + B) of a
    larger logical batch (one rank's shard) -> per-latent draws use global indices."""

    def __init__(self, batch_offset: int):
        self.batch_offset = int(batch_offset)

    def __enter__(self):
        self.prev = _SHARD.batch_offset
        _SHARD.batch_offset = self.batch_offset
        return self

    def __exit__(self, *exc):
        _SHARD.batch_offset = self.prev
        return False


def current_batch_offset() -> int:
    return _SHARD.batch_offset


# --------------------------------------------------------------------------------------------------
class NoiseGenerator:
    """py/noise_generation.py:87-179."""

    name = "unknown"
    MIN_DIMS = 1
    MAX_DIMS = 0

    def __init__(self, x, **kwargs):
        if x.ndim < self.MIN_DIMS:
            raise ValueError(f"Noise generator {self.name} requires at least {self.MIN_DIMS} dimension(s) but got input with shape {x.shape}")
        if self.MAX_DIMS > 0 and x.ndim > self.MAX_DIMS:
            raise ValueError(f"Noise generator {self.name} requires at most {self.MAX_DIMS} dimension(s) but got input with shape {x.shape}")
        defaults = self.ng_params()
        merged = defaults | kwargs
        for key in defaults:
            setattr(self, key, merged.pop(key))
        self.options = merged  # unknown keys (seed, sigma_min, ...) are kept, not rejected
        self.update_x(x)

    @classmethod
    def ng_params(cls):
        return {"normalized": True, "force_normalize": None, "normalize_dims": None, "cpu": True, "generator": None}

    def update_x(self, x):
        self.shape = x.shape
        if x.ndim in {4, 5}:
            self.batch, self.channels = x.shape[:2]
            self.height, self.width = x.shape[-2:]
            self.frames = x.shape[-3] if x.ndim == 5 else None
        else:
            self.batch = self.channels = self.frames = self.height = self.width = None
        self.device = x.device
        self.gen_device = torch.device("cpu") if self.cpu else self.device
        self.layout = x.layout
        self.dtype = x.dtype
        if not x.is_cuda:
            raise hip_lib.SonarHipError(
                f"Noise generator {self.name}: the latent lives on {x.device}; this implementation only runs on a ROCm device"
            )
        if x.dtype != torch.float32:
            raise hip_lib.SonarHipError(f"Noise generator {self.name}: float32 latents only (got {x.dtype})")

    # ---- draws
    def device_key(self, streams: int = 1) -> tuple[int, int]:
        """(seed, first stream id) for `streams` consecutive on-device draws."""
        return DeviceRNG.take(streams)

    def latent_elem_offset(self, per_latent: int) -> int:
        return current_batch_offset() * per_latent

    def rand_like(self, *, fun=torch.randn, cpu=None, to_device=True, shape=None, dtype=None, layout=None, device=None,
                  generator=None, partials=None):
        """py/noise_generation.py:133-155.  ``cpu`` draws on the host generator (replay), otherwise Philox on device."""
        cpu = fallback(cpu, self.cpu)
        shape = tuple(fallback(shape, self.shape))
        if cpu:
            noise = fun(*shape, generator=fallback(generator, self.generator), dtype=fallback(dtype, self.dtype),
                        layout=fallback(layout, self.layout), device="cpu")
            return tensor_to(noise, self.device) if to_device else noise
        if fun not in (torch.randn, torch.rand):
            raise NotImplementedError("on-device draws support torch.randn / torch.rand")
        seed, stream = self.device_key()
        per_latent = math.prod(shape[1:]) if len(shape) > 1 else 1
        offs = self.latent_elem_offset(per_latent) if shape[0] == self.shape[0] else 0
        if fun is torch.randn:
            return hip_lib.philox_normal(shape, self.device, seed, stream, offs, partials)
        return hip_lib.philox_uniform(shape, self.device, seed, stream, offs, partials=partials)

    def output_hook(self, noise):
        """py/noise_generation.py:157-165."""
        if noise.device != self.device:
            noise = tensor_to(noise, self.device)
        return scale_noise(noise, normalized=self.normalized and (self.force_normalize is None or self.force_normalize is True),
                           normalize_dims=self.normalize_dims)

    def pre_hook(self):
        pass

    def generate(self, *args):
        raise NotImplementedError

    def _plain_output(self) -> bool:
        """True when ``__call__`` hands back ``generate``'s device draw untouched (no normalisation in the output hook), so a generator
        may fold its values straight into a chain's running sum (``generate_into``: y <- y * y_mul + generate() * x_mul)."""
        return not self.cpu and not (self.normalized and self.force_normalize in (None, True)) and self.normalize_dims is None

    def __call__(self, *args, **kwargs):
        self.pre_hook()
        fused = getattr(self, "generate_normalized", None)
        if (fused is not None and not kwargs and self.normalized and self.force_normalize in (None, True) and self.normalize_dims is None):
            # generate + the output hook's scale_noise(normalized=True) as the generator's fused path (device draws: one write)
            noise = fused(1.0, *args)
            if noise is not None:
                return noise
        return self.output_hook(self.generate(*args, **kwargs))

    def __str__(self):
        params = ", ".join(f"{k}={getattr(self, k)!s}" for k in self.ng_params())
        return f"<NoiseGenerator({self.name}): device={self.device}, shape={self.shape}, dtype={self.dtype}, {params}>"


class FramesToChannelsNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:182-209: 5-D video latents are handled as (B, C*F, H, W)."""

    MIN_DIMS = 4
    MAX_DIMS = 5

    def get_adjusted_shape(self):
        c = self.channels * self.frames if self.frames else self.channels
        return (self.batch, c, self.height, self.width)

    def fix_output_frames(self, noise):
        if not self.frames:
            return noise
        partials = utils.pop_stats(noise)
        return attach_stats(noise.reshape(self.batch, self.channels, self.frames, self.height, self.width), partials)

    def rand_like(self, *args, shape=None, **kwargs):
        noise = super().rand_like(*args, shape=shape, **kwargs)
        if shape is not None:
            return noise
        adjusted = self.get_adjusted_shape()
        return noise.reshape(*adjusted) if tuple(noise.shape) != adjusted else noise


class MixedNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:212-249: sum of transformed sub-generators."""

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"name": "mixed_noise", "normalized": True, "pass_args": frozenset(("cpu",)),
                                      "noise_mix": (), "output_fun": None}

    def __init__(self, x, *args, **kwargs):
        lo = hi = None
        self.name = kwargs["name"]
        for entry in kwargs["noise_mix"]:
            klass = entry[0] if isinstance(entry, (tuple, list)) else entry
            lo = klass.MIN_DIMS if lo is None else max(lo, klass.MIN_DIMS)
            hi = klass.MAX_DIMS if hi is None else min(hi, klass.MAX_DIMS)
        self.MIN_DIMS, self.MAX_DIMS = lo, hi
        super().__init__(x, *args, **kwargs)
        passed = {k: v for k, v in kwargs.items() if k in self.pass_args}
        self.ng_list = [(klass(x, **klass_kwargs, **passed), transform) for klass, klass_kwargs, transform in self.noise_mix]

    @property
    def PLAN_STATIC(self) -> bool:  # noqa: N802 -- the generators' class attribute, here a function of the parts
        return all(getattr(gen, "PLAN_STATIC", False) for gen, _ in self.ng_list)

    def generate(self, *args):
        # total = sum_i transform_i(part_i), then output_fun.  A transform that only multiplies (``scale_factor``: the presets' _scaled)
        # rides in the accumulation kernel -- y * a + x * b rounds each product before the add, exactly like the multiply pass followed
        # by the add -- and so does an output multiply by a power of two (it commutes with every rounding); the last accumulation also
        # leaves the statistics the caller's normalisation would otherwise sweep the tensor for.  Same bits, three to four passes less.
        out_factor = getattr(self.output_fun, "scale_factor", None) if self.output_fun is not None else 1.0
        fold_out = out_factor is not None and out_factor != 0.0 and math.frexp(out_factor)[0] in (0.5, -0.5)
        g = out_factor if fold_out else 1.0
        total, pending = None, 1.0  # pending: the multiplier the running sum still owes (rides in the next accumulation)
        count = len(self.ng_list)
        for idx, (gen, transform) in enumerate(self.ng_list):
            # one part alive at a time: drawn just before its accumulation, dropped after it (several 134 MB tensors at batch 512 otherwise)
            part = gen(*args)
            utils.pop_stats(part)
            factor = getattr(transform, "scale_factor", None) if transform is not None else 1.0
            if factor is None:
                part, factor = transform(part), 1.0
                utils.pop_stats(part)
            factor = float(factor) * g
            if total is None:
                total, pending = part, factor
                continue
            fusable = all(p.dtype == torch.float32 and p.is_cuda and p.is_contiguous() for p in (total, part)) and part.shape == total.shape
            if fusable:
                if idx == count - 1 and (fold_out or self.output_fun is None):
                    total, partials = hip_lib.axpby_stats_(total, pending, part, factor)
                    return attach_stats(total, partials)
                hip_lib.axpby_(total, pending, part, factor)
            else:
                if pending != 1.0:
                    total = hip_lib.scale_noise_(total, pending, False, None)
                if factor != 1.0:
                    part = hip_lib.scale_noise_(part, factor, False, None)
                total = hip_lib.axpby_(total, 1.0, part, 1.0)
            pending = 1.0
            del part
        if pending != 1.0:
            total = hip_lib.scale_noise_(total, pending, False, None)
        return self.output_fun(total) if self.output_fun is not None and not fold_out else total


class GaussianNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:252-260."""

    name = "gaussian"
    PLAN_STATIC = True  # device-mode calls depend on nothing but the RNG position: a prepared plan may replay them (hip_lib.Planned)

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"normalized": False}

    def generate(self, *_args):
        if self.cpu:
            return self.rand_like()
        partials = hip_lib.new_partials(self.device)
        return attach_stats(self.rand_like(partials=partials), partials)

    def generate_normalized(self, factor, *_args):
        """generate() + scale_noise(factor, normalized=True), the tensor written once (device draws only)."""
        if self.cpu or self.normalize_dims is not None:
            return None
        seed, stream = self.device_key()
        shape = tuple(self.shape)
        return hip_lib.philox_noise(False, shape, self.device, seed, stream, self.latent_elem_offset(math.prod(shape[1:])), factor)

    accepts_prefix = True  # generate_into(..., pre=): the previous chain item's fold rides in this generator's kernel

    def generate_into(self, y, y_mul, x_mul, partials, *_args, pre=None):
        if not self._plain_output() or tuple(y.shape) != tuple(self.shape):
            return False
        self.pre_hook()
        seed, stream = self.device_key()
        hip_lib.philox_normal_acc_(y, y_mul, x_mul, seed, stream, self.latent_elem_offset(math.prod(self.shape[1:])), partials, pre=pre)
        return True

    def fold_prefix(self, y, y_mul, x_mul, *_args):
        """``generate_into`` captured instead of launched (``hip_lib.FoldPrefix``): the next chain item's kernel applies it in its own pass.
        ``y`` None: this item is the chain's first -- the running sum is allocated here and starts as this generator's raw values."""
        fresh = y is None
        if not self._plain_output() or (not fresh and tuple(y.shape) != tuple(self.shape)):
            return None
        if fresh:
            y = torch.empty(tuple(self.shape), dtype=torch.float32, device=self.device)
        self.pre_hook()
        seed, stream = self.device_key()
        return hip_lib.FoldPrefix(hip_lib.PREFIX_NORMAL, y, y_mul, x_mul, seed, stream, self.latent_elem_offset(math.prod(self.shape[1:])),
                                  fresh=fresh)


class UniformNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:496-514: (U[0,1) - sub_fac) * mul_fac + mean_fac."""

    name = "uniform"
    PLAN_STATIC = True

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"normalized": False, "sub_fac": 0.5, "mul_fac": 3.46, "mean_fac": 0.0}

    def generate(self, *_args):
        if self.cpu:
            return hip_lib.affine_(self.rand_like(fun=torch.rand), self.sub_fac, self.mul_fac, self.mean_fac)
        seed, stream = self.device_key()
        partials = hip_lib.new_partials(self.device)
        per_latent = math.prod(self.shape[1:])
        out = hip_lib.philox_uniform(tuple(self.shape), self.device, seed, stream, self.latent_elem_offset(per_latent),
                                     sub=self.sub_fac, mul=self.mul_fac, add=self.mean_fac, partials=partials)
        return attach_stats(out, partials)

    def generate_normalized(self, factor, *_args):
        if self.cpu or self.normalize_dims is not None:
            return None
        seed, stream = self.device_key()
        shape = tuple(self.shape)
        return hip_lib.philox_noise(True, shape, self.device, seed, stream, self.latent_elem_offset(math.prod(shape[1:])), factor,
                                    sub=self.sub_fac, mul=self.mul_fac, add=self.mean_fac)


class PerlinOldNoiseGenerator(FramesToChannelsNoiseGenerator):
    """py/noise_generation.py:289-493.  ``generate`` always evaluates the lattice with grid == output
    size, i.e. one pixel per cell sampled at the cell centre, and adds the [C,H,W] lattice term to every
    latent of the batch; the kernels implement exactly that case (``sonar_perlin_*``)."""

    name = "perlin_old"
    PLAN_STATIC = True

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"div_fac": 2.0, "iterations": 2, "blend_mode": "lerp"}

    def generate(self, *_args):
        if self.blend_mode not in hip_lib.BLEND_IDS:
            raise KeyError(self.blend_mode)
        b, c, h, w = self.get_adjusted_shape()
        partials = hip_lib.new_partials(self.device)
        two_pi = 2.0 * math.pi
        if self.cpu:
            base = self.rand_like(fun=torch.rand)  # drawn first (:480), then one lattice per iteration (:465-469)
            angles = torch.stack([
                torch.empty(c, h + 1, w + 1, dtype=torch.float32).uniform_(to=two_pi, generator=self.generator)
                for _ in range(self.iterations)
            ]) if self.iterations > 0 else torch.empty(0, c, h + 1, w + 1)
            terms = hip_lib.perlin_terms(tensor_to(angles, self.device).contiguous(), self.blend_mode)
            out = hip_lib.perlin_apply(base.contiguous(), terms, self.div_fac, partials)
        else:
            out = self._device_generate(partials, None)
        return self.fix_output_frames(attach_stats(out, partials))

    def _device_generate(self, partials, fused_factor):
        b, c, h, w = self.get_adjusted_shape()
        seed, stream = self.device_key(2)
        # the lattice is shared by every latent (and every rank): no batch offset in its counter.  Device draws have no
        # bit-parity reference: the angles are drawn inside the terms kernel and the iterations summed there ([C,H,W], 256 KiB),
        # so the streaming kernels read one table
        terms = hip_lib.perlin_lattice(max(self.iterations, 0), c, h, w, self.device, self.blend_mode, seed, stream + 1)
        offs = self.latent_elem_offset(c * h * w)
        if fused_factor is None:
            return hip_lib.perlin_generate((b, c, h, w), terms, self.div_fac, seed, stream, offs, partials)
        return hip_lib.perlin_noise((b, c, h, w), terms, self.div_fac, seed, stream, offs, fused_factor)

    def generate_normalized(self, factor, *_args):
        """generate() followed by scale_noise(factor, normalized=True), the tensor written once (device draws only)."""
        if self.cpu or self.normalize_dims is not None or self.blend_mode not in hip_lib.BLEND_IDS:
            return None
        return self.fix_output_frames(self._device_generate(None, factor))

    accepts_prefix = True

    def generate_into(self, y, y_mul, x_mul, partials, *_args, pre=None):
        if not self._plain_output() or tuple(y.shape) != tuple(self.shape) or self.blend_mode not in hip_lib.BLEND_IDS:
            return False
        self.pre_hook()
        b, c, h, w = self.get_adjusted_shape()
        seed, stream = self.device_key(2)  # the same keys, in the same order, as _device_generate
        terms = hip_lib.perlin_lattice(max(self.iterations, 0), c, h, w, self.device, self.blend_mode, seed, stream + 1)
        hip_lib.perlin_generate_acc_(y.view(b, c, h, w), y_mul, x_mul, terms, self.div_fac, seed, stream, self.latent_elem_offset(c * h * w), partials,
                                     pre=pre)
        return True

    def fold_prefix(self, y, y_mul, x_mul, *_args):
        """``generate_into`` with the lattice built and the keys taken, the fold itself left to the next chain item's kernel (``y`` None:
        the chain's first item, see ``GaussianNoiseGenerator.fold_prefix``)."""
        fresh = y is None
        if not self._plain_output() or (not fresh and tuple(y.shape) != tuple(self.shape)) or self.blend_mode not in hip_lib.BLEND_IDS:
            return None
        if fresh:
            y = torch.empty(tuple(self.shape), dtype=torch.float32, device=self.device)
        self.pre_hook()
        b, c, h, w = self.get_adjusted_shape()
        seed, stream = self.device_key(2)
        terms = hip_lib.perlin_lattice(max(self.iterations, 0), c, h, w, self.device, self.blend_mode, seed, stream + 1)
        return hip_lib.FoldPrefix(hip_lib.PREFIX_PERLIN, y, y_mul, x_mul, seed, stream, self.latent_elem_offset(c * h * w),
                                  terms=terms, div_fac=self.div_fac, fresh=fresh, view=(b, c, h, w))


def _level_ratios(seed: int, stream: int):
    """The pyramid levels' shrink ratios r in [2, 4) of a device-mode draw (py/noise_generation.py:628: ``torch.rand(1) * 2 + 2`` per level): a
    host-side splitmix64 sequence keyed by the call's (seed, stream) -- the same on every rank, no tensor work for a handful of scalars."""
    state = (seed * 0x9E3779B97F4A7C15 + stream) & 0xFFFFFFFFFFFFFFFF

    def draw() -> float:
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        return (z >> 11) * (2.0 / 9007199254740992.0) + 2.0

    return draw


class PyramidNoiseGenerator(FramesToChannelsNoiseGenerator):
    """py/noise_generation.py:609-649: Gaussian base plus bilinearly upsampled Gaussian levels whose
    sizes shrink by a random ratio r in [2,4) per level (cumulative), weighted discount**i."""

    name = "pyramid"
    PLAN_STATIC = True

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"discount": 0.7, "upscale_mode": "bilinear", "iterations": 10}

    def _plan(self, h, w, draw_r):
        sizes, cw, ch = [], w, h
        for i in range(self.iterations):
            r = draw_r()
            cw, ch = max(1, int(cw / (r**i))), max(1, int(ch / (r**i)))
            sizes.append((ch, cw))
            yield i, ch, cw
            if cw == 1 or ch == 1:
                break

    def generate(self, *_args):
        mode = self.upscale_mode
        if mode not in hip_lib.UPSCALE_MODES:
            raise NotImplementedError(f"pyramid upscale_mode {mode!r} is not on the HIP path")
        b, c, h, w = self.get_adjusted_shape()
        partials = hip_lib.new_partials(self.device)
        if self.cpu:
            noise = self.rand_like().contiguous()
            pending = None
            for i, ch, cw in self._plan(h, w, lambda: torch.rand(1, generator=self.generator).item() * 2 + 2):
                if pending is not None:
                    hip_lib.resample_acc_(noise, *pending)
                level = tensor_to(torch.randn(b, c, ch, cw, dtype=torch.float32), self.device)
                pending = (level, self.discount**i, mode, True)
            if pending is not None:
                hip_lib.resample_acc_(noise, *pending, partials)  # last level also reduces the statistics
            else:
                partials = None
            return self.fix_output_frames(attach_stats(noise, partials))
        return self.fix_output_frames(self._device_generate(partials, None))

    accepts_prefix = True  # generate_into(..., pre=): the previous chain item's fold rides in the plane kernel

    def generate_into(self, y, y_mul, x_mul, partials, *_args, pre=None):
        """y <- y * y_mul + generate() * x_mul in the plane kernel's own pass (``sonar_pyramid_generate_acc_f32``); a shape that kernel
        cannot run gets the same values, from the same keys, through generate + the accumulation kernel."""
        mode = self.upscale_mode
        if (not self._plain_output() or tuple(y.shape) != tuple(self.shape) or mode not in hip_lib.UPSCALE_MODES
                or y.dtype != torch.float32 or not y.is_contiguous()):
            return False
        self.pre_hook()
        b, c, h, w = self.get_adjusted_shape()
        keys = self.device_key(2 + self.iterations)
        seed, stream = keys
        if w % 4 == 0 and mode in hip_lib.PYRAMID_FUSED_MODES:
            if hip_lib.pyramid_generate_acc_(y.view(b, c, h, w), y_mul, x_mul, self._auto_levels(h, w, seed, stream), mode,
                                             seed, stream, self.latent_elem_offset(c * h * w), partials, pre=pre):
                return True
        if pre is not None:
            pre.apply()
        part = self._device_generate(None, None, keys=keys)
        utils.pop_stats(part)
        hip_lib.axpby_(y, y_mul, part.view(y.shape), x_mul)
        if partials is not None:
            hip_lib.stats(y, partials)
        return True

    def _auto_levels(self, h, w, seed, stream):
        """The all-device level list [(None, h_i, w_i, discount**i)] of ``_plan`` with ``_level_ratios(seed, stream)``, computed by the
        library (``sonar_pyramid_levels``: the same doubles in the same order) so that a prepared plan can recompute it per call."""
        return hip_lib.AutoLevels(h, w, self.iterations, self.discount, seed, stream)

    def _device_generate(self, partials, fused_factor, keys=None):
        mode = self.upscale_mode
        b, c, h, w = self.get_adjusted_shape()
        seed, stream = self.device_key(2 + self.iterations) if keys is None else keys
        plane_offset = current_batch_offset() * c
        offs = self.latent_elem_offset(c * h * w)

        def run(levels):
            if fused_factor is not None:
                return hip_lib.pyramid_noise((b, c, h, w), self.device, levels, mode, seed, stream, offs, fused_factor)
            out = hip_lib.pyramid_generate((b, c, h, w), self.device, levels, mode, seed, stream, offs, partials)
            return None if out is None else attach_stats(out, partials)

        fusable = w % 4 == 0 and mode in hip_lib.PYRAMID_FUSED_MODES  # nearest / bicubic: the unfused kernels below
        if fusable:
            # every level drawn on device: full-resolution levels fold into the base draw, the small grids are drawn by the
            # plane kernel (no launches, no HBM round trip for them); if that kernel cannot run this shape, explicit grids
            out = run(self._auto_levels(h, w, seed, stream))
            if out is not None:
                return out
        # From here on the level grids are tensors whose SIZES this call computed on the host from (seed, stream): a prepared plan that
        # recorded these launches would replay one call's sizes for every later call (round 6: scratch/fuzz_plans_r6.py found exactly that
        # on widths that are not multiples of four -- replayed values that were not the ordinary path's).  A trace that comes through here
        # yields no plan; the step stays on the ordinary path.
        rec = hip_lib._recorder
        if rec is not None and rec.thread == threading.get_ident():
            rec.fail("pyramid level grids with sizes computed per call on the host (a shape the plane kernel does not take)")
        plan = list(self._plan(h, w, _level_ratios(seed, stream)))  # shared by all ranks
        levels = []
        for i, ch, cw in plan:
            if (ch, cw) == (h, w) and not any(lv[0] is None for lv in levels):
                levels.append((None, h, w, self.discount**i))  # full-resolution level folded into the base draw
            else:
                grid = hip_lib.philox_normal((b * c, ch, cw), self.device, seed, stream + 2 + i, plane_offset * ch * cw)
                levels.append((grid, ch, cw, self.discount**i))
        if fusable:
            return run(levels)
        # rare odd widths / modes: same values through the unfused kernels
        out = hip_lib.philox_normal((b, c, h, w), self.device, seed, stream, offs)
        for grid, ch, cw, wt in levels:
            if grid is None:
                grid = hip_lib.philox_normal((b, c, h, w), self.device, seed, stream + 1, offs)
            hip_lib.resample_acc_(out, grid, wt, mode, True)
        if fused_factor is not None:
            return hip_lib.scale_noise_(out, fused_factor, True, hip_lib.stats(out))
        return attach_stats(out, hip_lib.stats(out, partials))

    def generate_normalized(self, factor, *_args):
        if self.cpu or self.normalize_dims is not None or self.upscale_mode not in hip_lib.UPSCALE_MODES:
            return None
        return self.fix_output_frames(self._device_generate(None, factor))

    def generate_raw_stats(self, *_args):
        """(raw noise, its statistics partials) -- what ``generate_normalized`` computes before its in-place normalisation pass, for a
        consumer that applies the normalisation itself while reading the noise (``NoiseSampler.deferred``); None when not applicable."""
        if self.cpu or self.normalize_dims is not None or self.upscale_mode not in hip_lib.UPSCALE_MODES:
            return None
        partials = hip_lib.new_partials(self.device)
        out = self._device_generate(partials, None)
        part = utils.pop_stats(out)
        return None if part is None else (self.fix_output_frames(out), part)


class HighresPyramidNoiseGenerator(FramesToChannelsNoiseGenerator):
    """py/noise_generation.py:517-564: uniform base plus Gaussian levels drawn at up to 15x the
    resolution and scaled DOWN to the latent size (replay mode only; the draws are ~60x the latent)."""

    name = "highres_pyramid"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.noise_generator is None:
            self.noise_generator = UniformNoiseGenerator(*args, **(kwargs | {"normalized": self.normalize_noise}))

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"normalized": True, "discount": 0.7, "upscale_mode": "bilinear", "iterations": 4,
                                      "noise_generator": None, "normalize_noise": False}

    def generate(self, s, sn):
        mode = self.upscale_mode
        if mode not in hip_lib.UPSCALE_MODES:
            raise NotImplementedError(f"highres_pyramid upscale_mode {mode!r} is not on the HIP path")
        b, c, h, w = self.get_adjusted_shape()
        noise = self.noise_generator(s, sn).reshape(b, c, h, w)
        utils.pop_stats(noise)
        rs = torch.rand(self.iterations, dtype=torch.float32, generator=self.generator).cpu() * 2 + 2
        ch, cw = h, w
        sizes = []
        for i in range(self.iterations):
            r = rs[i].item()
            ch, cw = min(h * 15, int(ch * (r**i))), min(w * 15, int(cw * (r**i)))
            sizes.append((ch, cw, self.discount**i))
            if ch >= h * 15 or cw >= w * 15:
                break
        if not self.cpu and noise.dtype == torch.float32 and noise.is_contiguous():
            # on-device draws: the levels (up to 225 x the latent's elements each) are never built -- a level value is a counter-based
            # normal keyed by its global element index and only the taps the shrinking interpolation reads are drawn
            seed, stream = self.device_key(max(len(sizes), 1))
            plane_offset = current_batch_offset() * c
            if hip_lib.levels_sampled((b, c, h, w), self.device, [(lh, lw, wt, 1.0) for lh, lw, wt in sizes], mode, seed, stream, plane_offset,
                                      out=noise) is None:
                for i, (lh, lw, wt) in enumerate(sizes):  # area off whole multiples: the windows overlap, the level is drawn (tile streams:
                    # the cheaper generator when every value is needed) and pooled
                    level = hip_lib.philox_normal((b, c, lh, lw), self.device, seed, stream + i, plane_offset * lh * lw)
                    hip_lib.resample_acc_(noise, level, wt, mode, True)
                    del level
            return self.fix_output_frames(noise)
        for lh, lw, wt in sizes:
            if self.cpu:
                level = tensor_to(torch.randn(b, c, lh, lw, generator=self.generator), self.device)
            else:
                seed, stream = self.device_key()
                level = hip_lib.philox_normal((b, c, lh, lw), self.device, seed, stream, current_batch_offset() * c * lh * lw)
            hip_lib.resample_acc_(noise, level, wt, mode, True)
        return self.fix_output_frames(noise)


class PyramidOldNoiseGenerator(FramesToChannelsNoiseGenerator):
    """py/noise_generation.py:567-606: levels at 2x..32x resolution, normal(std=0.5**i), scaled down."""

    name = "pyramid_old"
    PLAN_STATIC = True

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"discount": 0.8, "iterations": 5, "upscale_mode": "nearest-exact", "normalized": False}

    def generate(self, *_args):
        mode = self.upscale_mode
        if mode not in hip_lib.UPSCALE_MODES:
            raise NotImplementedError(f"pyramid_old upscale_mode {mode!r} is not on the HIP path")
        b, c, h, w = self.get_adjusted_shape()
        if not self.cpu:
            # on-device draws: a level value is a counter-based normal keyed by its global element index, so only the taps the shrinking
            # interpolation reads are drawn (the levels are up to 32 x 32 times the latent: 1 GiB for four SDXL latents); area modes
            # average whole blocks of independent normals -- the block mean is drawn directly, as one normal of the mean's variance
            seed, stream = self.device_key(max(self.iterations, 1))
            plane_offset = current_batch_offset() * c
            levels = [(h * (2 << i), w * (2 << i), self.discount**i, 0.5**i) for i in range(self.iterations)]
            out = hip_lib.levels_sampled((b, c, h, w), self.device, levels, mode, seed, stream, plane_offset)
            if out is None:
                out = torch.zeros((b, c, h, w), dtype=torch.float32, device=self.device)
                for i, (lh, lw, weight, sd) in enumerate(levels):
                    level = hip_lib.level_normal((b, c, lh, lw), self.device, sd, seed, stream + i, plane_offset)
                    hip_lib.resample_acc_(out, level, weight, mode, True)
                    del level
            return self.fix_output_frames(out)
        noise = torch.zeros((b, c, h, w), dtype=torch.float32, device=self.device)
        r = 1
        for i in range(self.iterations):
            r *= 2
            level = tensor_to(torch.normal(mean=0, std=0.5**i, size=(b, c, h * r, w * r), dtype=torch.float32, generator=self.generator), self.device)
            hip_lib.resample_acc_(noise, level, self.discount**i, mode, True)
        return self.fix_output_frames(noise)


class LaplacianNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:789-802: randn / div_fac + a Laplace(loc, scale) variate (torch.distributions.Laplace.rsample: a uniform on
    (eps - 1, 1) from the global generator, then loc - scale * sign(u) * log1p(-|u|))."""

    name = "laplacian"
    PLAN_STATIC = True

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"loc": 0, "scale": 1.0, "div_fac": 4.0}

    def generate(self, *_args):
        noise = self.rand_like()
        utils.pop_stats(noise)
        eps = float(torch.finfo(torch.float32).eps)
        if self.cpu:
            u = tensor_to(torch.empty(tuple(self.shape), dtype=torch.float32).uniform_(eps - 1.0, 1.0), self.device)
        else:
            seed, stream = self.device_key()
            u = hip_lib.philox_uniform(tuple(self.shape), self.device, seed, stream, self.latent_elem_offset(math.prod(self.shape[1:])),
                                       sub=0.0, mul=2.0 - eps, add=eps - 1.0)
            utils.pop_stats(u)
        return hip_lib.laplace_add_(noise.contiguous(), u.contiguous(), self.div_fac, self.loc, self.scale)


class PowerOldNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:1259-1287.  The reference draws a normal tensor only to take its batch size, scales a UNIFORM draw by
    k / (batch index + 1)^alpha per latent and standardises every [H, W] plane ((x - mean) / std, unbiased); same steps here."""

    name = "power_old"
    PLAN_STATIC = True

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"alpha": 2, "k": 1, "normalized": False}

    def generate(self, *_args):
        utils.pop_stats(self.rand_like())  # consumed and dropped, like the reference (keeps the generator's call order)
        noise = self.rand_like(fun=torch.rand)
        utils.pop_stats(noise)
        noise = noise.contiguous()
        b = noise.shape[0]
        per_latent = noise.numel() // max(b, 1)
        first = current_batch_offset() + 1  # batch shards keep their global latent index
        key = (first, b, self.k, self.alpha, str(self.device))
        cached = getattr(self, "_density", None)
        if cached is None or cached[0] != key:  # a blocking upload: once per batch shape, not per call
            density = (self.k / torch.arange(first, first + b, dtype=torch.float32) ** self.alpha).to(self.device)
            cached = self._density = (key, density, torch.zeros_like(density))
        _, density, zeros = cached
        noise = hip_lib.row_affine(1, noise, b, per_latent, zeros, density)
        hw = noise.shape[-1] * noise.shape[-2]
        rows = noise.numel() // hw
        mean, std = hip_lib.rowstats(noise, rows, hw)
        return hip_lib.row_affine(0, noise, rows, hw, mean, std)


class StudentTNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:652-677: a Student-t variate (torch.distributions.StudentT.rsample: a normal, then the chi-square through
    torch._standard_gamma(df / 2)), tails clamped at the per-latent quantile of |noise| and compressed by sign(x) |x|^pow_fac.
    Replay mode takes both base draws from the host generator in that order; on-device draws build the chi-square of an INTEGER df as
    a sum of squared normals (other df: replay mode only)."""

    name = "studentt"

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"loc": 0, "scale": 0.2, "df": 1, "quantile_fac": 0.75, "pow_fac": 0.5, "nq_fac": 1.0, "normalized": False}

    def generate(self, *_args):
        shape = tuple(self.shape)
        if self.cpu:
            noise = tensor_to(torch.empty(shape, dtype=torch.float32).normal_(), self.device)
            gamma = tensor_to(torch._standard_gamma(torch.full(shape, 0.5 * self.df, dtype=torch.float32)), self.device)
        else:
            df = int(self.df)
            if df != self.df or not 1 <= df <= 16:
                raise NotImplementedError("studentt: on-device draws need an integer df in 1..16 (use cpu noise for other values)")
            noise = self.rand_like()
            utils.pop_stats(noise)
            gamma = torch.empty_like(noise)
            for i in range(df):  # chi2(df) = sum of df squared normals; gamma(df / 2) = chi2 / 2
                z = self.rand_like()
                utils.pop_stats(z)
                hip_lib.sq_acc_(gamma, z.contiguous(), 0.5, i == 0)
        noise = hip_lib.studentt_(noise.contiguous(), gamma.contiguous(), self.loc, self.scale, self.df)
        b = noise.shape[0]
        inner = noise.numel() // max(b, 1)
        nq = hip_lib.abs_quantile_rows(noise, b, inner, self.quantile_fac)
        return hip_lib.clamp_signpow_rows_(noise, b, inner, nq, self.nq_fac, self.pow_fac)


def _off_path(type_name: str):
    class _OffPath(NoiseGenerator):
        name = type_name

        def __init__(self, x, **kwargs):
            raise NotImplementedError(
                f"noise type {type_name!r} is outside the MI355X hot path of this build (SURVEY.md §8: out of scope); "
                "there is deliberately no CPU fallback"
            )

    _OffPath.__name__ = f"{type_name.title().replace('_', '')}NoiseGenerator"
    return _OffPath


DistroNoiseGenerator = _off_path("distro")
VoronoiNoiseGenerator = _off_path("voronoi")
CollatzNoiseGenerator = _off_path("collatz")
ScatternetFilteredNoiseGenerator = _off_path("scatternet_filtered")


class PowerLawNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:762-786: spatial "power law" (white / grey / velvet / violet presets):
    (sign(x) or x) * |x|**alpha, optionally divided by the max |.| over ``div_max_dims``."""

    name = "powerlaw"
    PLAN_STATIC = True

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"alpha": 2.0, "div_max_dims": None, "use_sign": False, "use_div_max_abs": True}

    def generate(self, *_args):
        noise = hip_lib.powerlaw_(self.rand_like().contiguous(), self.alpha, self.use_sign)
        if self.div_max_dims is not None:
            dims = self.div_max_dims
            dims = tuple(range(noise.ndim)) if dims == () else sorted({d % noise.ndim for d in ((dims,) if isinstance(dims, int) else dims)})
            if list(dims) != list(range(dims[0], dims[-1] + 1)):
                raise hip_lib.SonarHipError("powerlaw: div_max_dims must be adjacent dimensions on the HIP path")
            outer = math.prod(noise.shape[: dims[0]])
            mid = math.prod(noise.shape[d] for d in dims)
            inner = math.prod(noise.shape[dims[-1] + 1:])
            peak = hip_lib.amax_mid(noise, outer, mid, inner, self.use_div_max_abs)
            hip_lib.div_mid_(noise, outer, mid, inner, peak)
        return noise


class WaveletFilteredNoiseGenerator(FramesToChannelsNoiseGenerator):
    """py/noise_generation.py:1908-2032: DWT -> optional blend with a second noise's bands -> per-band scaling -> IDWT."""

    name = "waveletfilter"
    MIN_DIMS = 4
    MAX_DIMS = 5

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from .wavelet_functions import Wavelet

        inv = {k: self.options[k] for k in ("inv_mode", "inv_biort", "inv_qshift", "inv_wave") if k in self.options}
        self.wavelet = Wavelet(wave=self.wave, level=self.level, mode=self.mode, use_1d_dwt=self.use_1d_dwt, use_dtcwt=self.use_dtcwt,
                               biort=self.biort, qshift=self.qshift, device=self.gen_device, **inv)

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {
            "mode": "periodization", "level": 3, "wave": "haar", "use_1d_dwt": False, "use_dtcwt": False, "qshift": "qshift_a",
            "biort": "near_sym_a", "yl_scale": 1.0, "yh_scales": 1.0, "two_step_inverse": False, "preblend_yl_scale_low": None,
            "preblend_yh_scales_low": None, "preblend_yl_scale_high": None, "preblend_yh_scales_high": None,
            "yl_blend_function": utils.BLENDING_MODES["lerp"], "yh_blend_function": utils.BLENDING_MODES["lerp"],
            "yl_blend_high": 0.0, "yh_blend_high": 1.0, "noise_sampler": None, "noise_sampler_high": None,
        }

    def generate(self, *args):
        from .wavelet_functions import wavelet_blend, wavelet_scaling

        shape = self.get_adjusted_shape()
        noise = self.rand_like() if self.noise_sampler is None else self.noise_sampler(*args)
        utils.pop_stats(noise)
        noise = noise.reshape(*shape).contiguous()
        need_flat = self.use_1d_dwt and noise.ndim > 3  # :1982-1986 -- the 1-D transform runs over the flattened plane
        flat = (lambda t: t.flatten(start_dim=2)) if need_flat else (lambda t: t)
        yl, yh = self.wavelet.forward(flat(noise))
        if self.noise_sampler_high is not None:
            high = self.noise_sampler_high(*args)
            utils.pop_stats(high)
            yl_h, yh_h = self.wavelet.forward(flat(high.reshape(*shape).contiguous()))
            if self.preblend_yl_scale_high is not None or self.preblend_yh_scales_high is not None:
                yl_h, yh_h = wavelet_scaling(yl_h, yh_h, fallback(self.preblend_yl_scale_high, 1.0), fallback(self.preblend_yh_scales_high, 1.0), in_place=True)
            if self.preblend_yl_scale_low is not None or self.preblend_yh_scales_low is not None:
                yl, yh = wavelet_scaling(yl, yh, fallback(self.preblend_yl_scale_low, 1.0), fallback(self.preblend_yh_scales_low, 1.0), in_place=True)
            yl, yh = wavelet_blend((yl, yh), (yl_h, yh_h), yl_factor=self.yl_blend_high, yh_factor=self.yh_blend_high,
                                   blend_function=self.yl_blend_function, yh_blend_function=self.yh_blend_function)
        yl, yh = wavelet_scaling(yl, yh, self.yl_scale, self.yh_scales, in_place=True)
        result = self.wavelet.inverse(yl, yh, two_step_inverse=self.two_step_inverse)
        if need_flat:  # pixel counts the transform cannot return exactly (odd H*W) fail here, like the reference's reshape
            result = result.reshape(*shape)
        if tuple(result.shape) != tuple(shape):
            result = result[tuple(slice(0, d) for d in shape)].contiguous()
        return self.fix_output_frames(result)


class PinkOldNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:707-718 (a scalar gain; the reference itself calls it wrong)."""

    name = "pink_old"
    PLAN_STATIC = True

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"alpha": 2.0, "k": 1.0, "freq": 1.0}

    def generate(self, *_args):
        noise = self.rand_like()
        return hip_lib.mul_scalar(noise, self.k / self.freq**self.alpha, out=noise)


def _half_gain(gain: torch.Tensor) -> torch.Tensor:
    """Re(ifft2(fft2(x) * G)) for real x equals irfft2(rfft2(x) * Gs) with Gs(k) = (G(k) + G(-k)) / 2 (the odd part of G
    only feeds the discarded imaginary output); returns the [H, W/2+1] half of Gs."""
    mirrored = torch.roll(torch.flip(gain, dims=(-2, -1)), shifts=(1, 1), dims=(-2, -1))
    return ((gain + mirrored) * 0.5)[..., : gain.shape[-1] // 2 + 1].contiguous()


class _SpectralGainNoiseGenerator(FramesToChannelsNoiseGenerator):
    """Shared by OneF / GreenTest: white noise -> per-plane LDS-resident rfft2 x gain -> irfft2 (one read, one write)."""

    MIN_DIMS = 4
    MAX_DIMS = 5
    PLAN_STATIC = True

    def spectral_gain(self) -> torch.Tensor:  # full [H, W] real gain, host (setup arithmetic as the reference writes it)
        raise NotImplementedError

    def filtered(self, partials=None):
        noise = self.rand_like()
        utils.pop_stats(noise)
        if not hip_lib.power_supported(self.height, self.width):
            raise hip_lib.SonarHipError(f"{self.name}: plane {self.height}x{self.width} is beyond the spectral kernels (sides of at most 2048)")
        return hip_lib.spectral_filter(noise.contiguous(), self.device_gain(), partials)

    def device_gain(self) -> torch.Tensor:
        """The half-spectrum gain on the device.  It depends on the plane size and the generator's parameters only, so it is built (host
        arithmetic + one blocking upload: ~2.5 ms for a 128 x 128 plane) when one of them changed, not on every call."""
        key = (self.height, self.width, str(self.device)) + tuple(repr(getattr(self, k, None)) for k in sorted(self.ng_params()))
        cached = getattr(self, "_device_gain", None)
        if cached is None or cached[0] != key:
            cached = self._device_gain = (key, _half_gain(self.spectral_gain().to(torch.float32)).to(self.device))
        return cached[1]


class GreenTestNoiseGenerator(_SpectralGainNoiseGenerator):
    """py/noise_generation.py:680-704: ifft2(fft2(x) / sqrt(sqrt(fy^p + fx^q))) * scale / std.  The reference takes the std of
    the COMPLEX ifft2 output; its imaginary part is rounding noise for the (even) default gain, so the real std is used."""

    name = "green_test"

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"scale_fac": 1.0, "x_pow": 2, "y_pow": 2, "power_base": 1}

    def spectral_gain(self):
        fy = torch.fft.fftfreq(self.height)[:, None] ** self.y_pow
        fx = torch.fft.fftfreq(self.width) ** self.x_pow
        power = torch.sqrt(fy + fx)
        power[0, 0] = self.power_base
        return 1.0 / torch.sqrt(power)

    def generate(self, *_args):
        partials = hip_lib.new_partials(self.device)
        noise = self.filtered(partials)
        hip_lib.std_scale_(noise, self.scale_fac / (self.width * self.height), partials)
        return self.fix_output_frames(noise)


class OneFNoiseGenerator(_SpectralGainNoiseGenerator):
    """py/noise_generation.py:720-759.  The reference transforms over ALL dims (fftn / ifftn); its gain depends on (h, w)
    only, so the batch / channel transforms cancel exactly and the result is the per-plane 2-D filter computed here."""

    name = "onef"

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"alpha": 2.0, "k": 1.0, "hfac": 1.0, "wfac": 1.0, "base_power": 1.0, "use_sqrt": True}

    def spectral_gain(self):
        fx, fy = torch.meshgrid(torch.fft.fftfreq(self.height, self.hfac), torch.fft.fftfreq(self.width, self.wfac), indexing="ij")
        power = (fx**2 + fy**2) ** (-self.alpha / 2.0)
        if self.k != 0:
            power = self.k / power
        power[0, 0] = self.base_power
        if not self.use_sqrt:
            return 1.0 / power
        # the reference divides the spectrum by sqrt(power) in COMPLEX arithmetic (py/noise_generation.py:752-757): where the power is
        # negative the divisor is imaginary, the (Hermitian) spectrum of the real noise turns anti-Hermitian there, and those frequencies
        # vanish from the real part it returns -- a gain of zero
        return torch.where(power < 0, torch.zeros_like(power), 1.0 / torch.sqrt(power.clamp_min(0.0)))

    def generate(self, *_args):
        partials = hip_lib.new_partials(self.device)
        return self.fix_output_frames(attach_stats(self.filtered(partials), partials))


# --------------------------------------------------------------------------------------------------
# Brownian-interval noise.  The reference delegates to ComfyUI's k-diffusion BrownianTreeNoiseSampler (torchsde's
# BrownianTree; un-vendored, not installed, no reference tests at that boundary): "parity unpinned".  Same contract here --
# W(t) is ONE Brownian path per element and seed, a call returns (W(t1) - W(t0)) / sqrt|t1 - t0| with the reference's sign
# convention, so every call is N(0,1) and repeated / nested / abutting intervals are mutually consistent -- but the path is
# this build's own: counter-based normals (sonar_brownian_bridge_f32) on a virtual Brownian tree (default; BrownianPath's tree mode) or
# as bridges between the times asked for (SONAR_BROWNIAN_TREE=0), so values differ from torchsde's.
class BrownianPath:
    """W(t) on [t_lo, t_hi] as a linear combination of per-node standard normals: host side of sonar_brownian_*_f32.

    The path is defined point by point, in the order times are asked for (torchsde's BrownianInterval grows its tree the same way): a new
    time t between the nearest known times a < t < b is the Brownian bridge W(t) = ((b - t) W(a) + (t - a) W(b)) / (b - a) +
    sqrt((t - a)(b - t) / (b - a)) z(node), node = the point's creation number; beyond every known time it is an independent increment
    from the outermost one (the two-sided motion continues outside [t_lo, t_hi]).  By the Markov property the joint law of all known
    points is exactly a Brownian motion's.  ``bridge[t]`` = (a, b, fa, fb, sd, node) is how a point was made -- all a sampler run needs,
    O(1) per point; ``coefficients(t)`` expands W(t) over the node normals (fp64) on demand, for the evaluation without kept tensors.

    ORDER DEPENDENCE (unlike ComfyUI's BrownianTree, whose dyadic tree makes W(t) a fixed function of the seed up to its tolerance):
    a point's value depends on which points were known when it was first asked for, i.e. on the HISTORY of queries of this instance,
    not only on (seed, t).  One sampler run is self-consistent (every increment it sees comes from one path), and two runs that ask for
    the same times in the same order see the same path; two runs over the same range with different step counts, or a resumed run
    that re-creates the sampler, see different (equally valid) paths at the times they share -- as they do with torchsde's
    BrownianInterval without its dyadic pre-tree.  ``tests/test_abi_and_host.py::test_brownian_path_depends_on_the_query_order``.

    TREE MODE (``tree_depth`` = D > 0; what samplers get by default since round 5, ``BROWNIAN_TREE_DEPTH`` below): the virtual Brownian
    tree -- what ComfyUI's BrownianTree is, up to its tolerance.  A time inside (t_lo, t_hi) is snapped to the grid of 2**D cells (D = 24: 6e-8 of the range), and a grid
    point is defined through its dyadic ancestors only, top down: the midpoint of [t_lo, t_hi], then the midpoint of the half that holds
    it, ... each a bridge between the ends of ITS dyadic interval with the node id of its place in the tree (heap index: 1, then 2h /
    2h + 1).  W(t) is then a fixed function of (seed, t) -- no history: any query order, any step count, a sampler re-created half way
    through a run all see the same path -- at the price of up to D + 1 normals per element and evaluation where the path of bridges
    (``tree_depth`` = 0) needs one (cfg5's shard: 0.34 ms per call instead of 0.10).  Times outside [t_lo, t_hi] keep
    the bridge path's extensions (history-dependent there)."""

    ROOT = 0              # node ids are creation numbers; they stay below 2**40 (the kernel's stream-id field has 48 bits)
    MEMO = 512            # expansions remembered (each can hold every earlier node: a run of n monotone queries makes them O(n) long)
    MAX_TREE_DEPTH = 36

    def __init__(self, t_lo: float, t_hi: float, tree_depth: int = 0):
        self.t_lo, self.t_hi = float(t_lo), float(t_hi)
        if not self.t_hi > self.t_lo:
            raise ValueError("Brownian noise needs sigma_min < sigma_max")
        self.tree_depth = int(tree_depth)
        if not 0 <= self.tree_depth <= self.MAX_TREE_DEPTH:
            raise ValueError(f"Brownian noise: tree depth 0 (off) .. {self.MAX_TREE_DEPTH}")
        # grid times are floats: a cell narrower than a few ulps of the largest time (a narrow range at a large offset, a very deep tree)
        # would make neighbouring grid points collide -- the tree is only as deep as the range resolves
        ulp = math.ulp(max(abs(self.t_lo), abs(self.t_hi)))
        while self.tree_depth > 1 and (self.t_hi - self.t_lo) / (1 << self.tree_depth) < 4.0 * ulp:
            self.tree_depth -= 1
        self.times = [self.t_lo, self.t_hi]  # sorted; W(lo) = 0, W(hi) ~ N(0, hi - lo)
        self.bridge: dict = {}               # t -> (a, b or None, fa, fb, sd, node)
        # tree mode: the dyadic nodes own the ids 1 .. 2**D - 1; creation numbers (extensions beyond the range) start above them
        self._next = 1 << self.tree_depth if self.tree_depth else 1
        self._memo: dict = {}

    # ---- tree mode: grid index g in [0, 2**D] <-> time
    def _grid_time(self, g: int) -> float:
        cells = 1 << self.tree_depth
        return self.t_lo if g <= 0 else self.t_hi if g >= cells else self.t_lo + g * ((self.t_hi - self.t_lo) / cells)

    def grid_index(self, t: float) -> int:
        """Tree mode: the index in [0, 2**D] of the grid point a time inside [t_lo, t_hi] stands for."""
        cells = 1 << self.tree_depth
        return min(cells, max(0, int(round((float(t) - self.t_lo) / (self.t_hi - self.t_lo) * cells))))

    def resolve(self, t: float) -> float:
        """The time a query of t stands for: t itself, or in tree mode the nearest grid point when t lies inside the range."""
        t = float(t)
        if not self.tree_depth:
            return t
        cells = 1 << self.tree_depth
        if not self.t_lo < t < self.t_hi:
            # within half a grid cell beyond an end: the end itself (a sigma_max handed over as a float32 tensor and queried as a float64
            # one differs in the eighth digit; without the snap that query is an extension beyond the range -- a history-dependent point)
            half = 0.5 * (self.t_hi - self.t_lo) / cells
            if self.t_lo - half < t <= self.t_lo:
                return self.t_lo
            if self.t_hi <= t < self.t_hi + half:
                return self.t_hi
            return t
        return self._grid_time(int(round((t - self.t_lo) / (self.t_hi - self.t_lo) * cells)))

    def _define_dyadic(self, t: float) -> None:
        """t (a grid time) and every dyadic ancestor that is not a point yet, top down."""
        import bisect

        cells = 1 << self.tree_depth
        g = int(round((t - self.t_lo) / (self.t_hi - self.t_lo) * cells))
        lo, hi, h = 0, cells, 1
        while hi - lo > 1:
            m = (lo + hi) // 2
            tm = self._grid_time(m)
            if tm not in self.bridge:
                a, b = self._grid_time(lo), self._grid_time(hi)
                fb = (tm - a) / (b - a)
                self.bridge[tm] = (a, b, 1.0 - fb, fb, math.sqrt((tm - a) * (b - tm) / (b - a)), h)
                bisect.insort(self.times, tm)
            if m == g:
                return
            lo, hi, h = (lo, m, 2 * h) if g < m else (m, hi, 2 * h + 1)
        raise AssertionError("grid point not reached")  # (unreachable: every interior grid index is some interval's midpoint)

    def define(self, t: float) -> None:
        """Make t a point of the path (a bridge between its neighbours, or an extension beyond the outermost known time)."""
        t = self.resolve(t)
        if t in self.bridge or t == self.t_lo or t == self.t_hi:
            return
        if self.tree_depth and self.t_lo < t < self.t_hi:
            self._define_dyadic(t)
            return
        import bisect

        i = bisect.bisect_left(self.times, t)
        node = self._next
        if node >= (1 << 40):
            raise RuntimeError("Brownian noise: out of node ids")
        self._next += 1
        if 0 < i < len(self.times):
            a, b = self.times[i - 1], self.times[i]
            fb = (t - a) / (b - a)
            made = (a, b, 1.0 - fb, fb, math.sqrt((t - a) * (b - t) / (b - a)), node)
        else:
            a = self.times[-1] if i else self.times[0]
            made = (a, None, 1.0, 0.0, math.sqrt(abs(t - a)), node)
        self.times.insert(i, t)
        self.bridge[t] = made

    def coefficients(self, t: float) -> dict:
        """{node id: coefficient} with W(t) = sum coefficient * z(node); defines the point if it is new."""
        t = self.resolve(t)
        self.define(t)
        memo = self._memo
        if len(memo) > self.MEMO:
            memo.clear()
        base = {self.t_lo: {}, self.t_hi: {self.ROOT: math.sqrt(self.t_hi - self.t_lo)}}
        stack = [t]
        while stack:  # iterative: a monotone run nests its points n deep
            u = stack[-1]
            if u in memo or u in base:
                stack.pop()
                continue
            a, b, fa, fb, sd, node = self.bridge[u]
            need = [p for p in (a, b) if p is not None and p not in memo and p not in base]
            if need:
                stack.extend(need)
                continue
            ca = memo.get(a, base.get(a))
            cb = {} if b is None else memo.get(b, base.get(b))
            out = {k: fa * ca.get(k, 0.0) + fb * cb.get(k, 0.0) for k in ca.keys() | cb.keys()}
            out[node] = sd
            memo[u] = out
            stack.pop()
        return memo[t] if t in memo else base[t]

    def increment(self, t0: float, t1: float):
        """(node ids, coefficients) of (W(t_max) - W(t_min)) / sqrt(t_max - t_min); the smaller time is defined first."""
        t0, t1 = self.resolve(t0), self.resolve(t1)
        ta, tb = (t0, t1) if t0 <= t1 else (t1, t0)
        if ta == tb:
            raise ValueError("Brownian noise needs two distinct times")
        self.define(ta)
        self.define(tb)
        ca = dict(self.coefficients(ta))  # a copy: the memo may be cleared by the second expansion
        cb = self.coefficients(tb)
        scale = 1.0 / math.sqrt(tb - ta)
        terms = {k: (cb.get(k, 0.0) - ca.get(k, 0.0)) * scale for k in ca.keys() | cb.keys()}
        terms = {k: v for k, v in terms.items() if abs(v) > 1e-12}
        ids = sorted(terms)
        return ids, [terms[k] for k in ids]


def _env_tree_depth() -> int:
    v = os.environ.get("SONAR_BROWNIAN_TREE", "1").strip().lower()
    if v in ("0", "off", "false"):
        return 0
    if v in ("", "1", "on", "true"):
        return 24
    try:
        depth = int(v)
    except ValueError:  # a malformed value must not make the package fail to import
        return 24
    return depth if 0 <= depth <= BrownianPath.MAX_TREE_DEPTH else 24


# Depth of the virtual Brownian tree (BrownianPath, TREE MODE) for samplers made from now on: 24 (BrownianTree's tolerance of 1e-6 on a
# sigma range of ~15) unless the environment says otherwise -- SONAR_BROWNIAN_TREE=<depth>, or =0 for the path of bridges between the
# times asked for (3.4 x cheaper per call, but a function of the query history; the default of rounds 2-4).
BROWNIAN_TREE_DEPTH = _env_tree_depth()


class BrownianTreeNoiseSampler:
    """Interface of k-diffusion's BrownianTreeNoiseSampler (x, sigma_min, sigma_max, seed, transform, cpu) -> (sigma, sigma_next);
    ``tree_depth``: BrownianPath's tree mode for this sampler (None: the module's BROWNIAN_TREE_DEPTH)."""

    def __init__(self, x: Tensor, sigma_min, sigma_max, seed=None, transform=lambda t: t, cpu: bool = False, tree_depth: Optional[int] = None):
        if not x.is_cuda:
            raise hip_lib.SonarHipError("Brownian noise: the latent must live on a ROCm device")
        self.transform = transform
        t0, t1 = float(transform(torch.as_tensor(sigma_min))), float(transform(torch.as_tensor(sigma_max)))
        self.sign = 1.0 if t0 <= t1 else -1.0
        self.path = BrownianPath(min(t0, t1), max(t0, t1), BROWNIAN_TREE_DEPTH if tree_depth is None else tree_depth)
        self.shape, self.device = tuple(x.shape), x.device
        if seed is None:
            seed = int(torch.randint(0, 2**63 - 1, []).item())
        self.latent_seeds = None
        try:
            seeds = [int(v) for v in seed]
            if len(seeds) != x.shape[0]:
                raise ValueError("Brownian noise: one seed per batch item expected")
            self.seed = 0
            self.latent_seeds = torch.tensor([v & (2**63 - 1) for v in seeds], dtype=torch.int64, device=self.device)
        except TypeError:
            self.seed = int(seed)
        self.elem_offset = current_batch_offset() * (x.numel() // x.shape[0])  # batch shards draw their own global elements
        self._points: dict = {}  # t -> W(t) tensor, least recently used first
        self._coarse_pts: dict = {}  # tree mode: coarse-grid index -> W tensor, least recently used first

    # W(t) tensors kept: a sampler step ends where the next begins and DPM++ SDE asks (t, s) and (t, t') per step, so the two times a
    # new one is bridged between are the previous step's and the interval's end -- ONE fresh normal per element per call
    CACHE_POINTS = 4
    # Tree mode, round 6: W(t) is evaluated in TWO stages, always -- W at the two ends a, b of the cell of a coarse dyadic grid (2**4 cells)
    # that holds t, each an fp32 tensor from its own expansion over the tree's top levels (at most five normals), then
    # W(t) = fa W(a) + fb W(b) + sum over the nodes below that level.  By the bridge construction the top levels' share of W(t) IS that
    # interpolation, so the value is the expansion's up to rounding -- and because the rule never depends on what happens to be kept, W(t)
    # is still a function of (seed, t) alone, bit for bit.  What it buys: the coarse tensors are the same for every time in the cell
    # (a sampler walks through a cell in several steps and into the next one across a shared end), so a call draws D - 4 normals per
    # element instead of D + 1 (cfg5's shard: 250 -> 205 us per call); a sampler made half way through a run pays the two coarse ends once.
    TREE_COARSE_LEVEL = 4
    COARSE_POINTS = 3

    def _remember(self, t: float, w: Tensor) -> None:
        self._points.pop(t, None)
        self._points[t] = w
        while len(self._points) > self.CACHE_POINTS:
            self._points.pop(next(iter(self._points)))

    def _cached(self, t: float, *, cheap: bool = False) -> Optional[Tensor]:
        """The kept W(t); with ``cheap`` the interval's end, whose expansion is a single normal, is evaluated and kept when it is not."""
        w = self._points.get(t)
        if w is not None:
            self._remember(t, w)
        elif cheap and t == self.path.t_hi:
            _, w = hip_lib.brownian_bridge(self.shape, self.device, [self.path.ROOT], [math.sqrt(self.path.t_hi - self.path.t_lo)], self.seed,
                                           self.elem_offset, self.latent_seeds, want_out=False)
            self._remember(t, w)
        return w

    def _emit(self, ids, coefs, *, base_a=None, fa=0.0, base_b=None, fb=0.0, prev=None, scale=1.0, want_out=True, want_w=True, fold=None,
              partials=None):
        """One evaluation W = fa base_a + fb base_b + sum coef z(node): (scale * (W - prev) or None, W or None).  With ``fold`` = (y, y_mul,
        x_mul, partials) the increment is folded into the chain's running sum y instead of being written out (``accumulate``)."""
        tail = (self.seed, self.elem_offset, self.latent_seeds)
        bases = dict(base_a=base_a, fa=fa, base_b=base_b, fb=fb)
        if fold is None:
            return hip_lib.brownian_bridge(self.shape, self.device, ids, coefs, *tail, **bases, prev=prev, scale=scale, want_out=want_out, want_w=want_w,
                                           partials=partials)
        y, y_mul, x_mul, partials, pre = fold
        w = None
        if len(ids) > hip_lib.BROWNIAN_MAX_TERMS:  # a long expansion: W first (in chunks), then the fold with no terms left
            _, w = hip_lib.brownian_bridge(self.shape, self.device, ids, coefs, *tail, **bases, want_out=False)
            ids, coefs, bases = [], [], dict(base_b=w, fb=1.0)
            want_w = False
        made = hip_lib.brownian_bridge_acc_(y, y_mul, x_mul, ids, coefs, *tail, **bases, prev=prev, scale=scale, partials=partials, want_w=want_w,
                                            pre=pre)
        return y, (made if w is None else w)

    def _coarse(self, g: int) -> Optional[Tensor]:
        """Tree mode: the kept W at index g of the coarse grid (None: W(t_lo) = 0), evaluated from its own expansion when it is not kept."""
        path = self.path
        cells = 1 << path.tree_depth
        if g <= 0:
            return None
        if g >= cells:
            return self._cached(path.t_hi, cheap=True)
        kept = self._coarse_pts
        w = kept.pop(g, None)
        if w is None:
            terms = path.coefficients(path._grid_time(g))
            ids = sorted(terms)
            _, w = hip_lib.brownian_bridge(self.shape, self.device, ids, [terms[k] for k in ids], self.seed, self.elem_offset, self.latent_seeds,
                                           want_out=False)
        kept[g] = w
        while len(kept) > self.COARSE_POINTS:
            kept.pop(next(iter(kept)))
        return w

    def _tree_point(self, t: float, *, prev, scale, want_out, fold, partials):
        """Tree mode's two-stage evaluation of a time inside the range (see TREE_COARSE_LEVEL); None: not applicable (a shallow tree)."""
        path = self.path
        shift = path.tree_depth - self.TREE_COARSE_LEVEL
        if shift < 2 or not path.t_lo < t < path.t_hi:
            return None
        g = path.grid_index(t)
        ga = (g >> shift) << shift
        terms = path.coefficients(t)
        if g == ga:  # a point of the coarse grid itself: its expansion only holds top-level nodes
            return None
        gb = ga + (1 << shift)
        ta, tb = path._grid_time(ga), path._grid_time(gb)
        top = 1 << self.TREE_COARSE_LEVEL  # heap indices below it (and the root's id 0) belong to the coarse grid
        ids = sorted(k for k in terms if k >= top)
        wa, wb = self._coarse(ga), self._coarse(gb)
        fb = (t - ta) / (tb - ta)
        return self._emit(ids, [terms[k] for k in ids], base_a=wa, fa=1.0 - fb, base_b=wb, fb=fb, prev=prev, scale=scale, want_out=want_out, fold=fold,
                          partials=partials)

    def _point(self, t: float, *, prev: Optional[Tensor] = None, scale: float = 1.0, want_out: bool = True, fold=None, partials=None):
        """(scale * (W(t) - prev) or None, W(t)) for a time that is not kept: the bridge between its two kept neighbours, else its expansion."""
        made = self.path.bridge.get(t)
        out = w = None
        if self.path.tree_depth:  # (whatever CACHE_POINTS says: the tree's values are defined by the two-stage rule)
            two_stage = self._tree_point(t, prev=prev, scale=scale, want_out=want_out, fold=fold, partials=partials)
            if two_stage is not None:
                out, w = two_stage
                self._remember(t, w)
                return out, w
        # (tree mode always expands: W(t) is then the same bits whatever happens to be kept, a function of (seed, t) alone)
        if made is not None and self.CACHE_POINTS > 0 and not self.path.tree_depth:
            a, b, fa, fb, sd, node = made
            wa = None if a == self.path.t_lo else self._cached(a, cheap=True)  # W(t_lo) = 0
            wb = None if b is None else self._cached(b, cheap=True)            # b is None: an extension beyond the known times
            if (wb is not None or b is None) and (wa is not None or a == self.path.t_lo):
                out, w = self._emit([node], [sd], base_a=wa, fa=fa, base_b=wb, fb=fb, prev=prev, scale=scale, want_out=want_out, fold=fold,
                                    partials=partials)
        if w is None:
            terms = self.path.coefficients(t)
            ids = sorted(terms)
            out, w = self._emit(ids, [terms[k] for k in ids], prev=prev, scale=scale, want_out=want_out, fold=fold, partials=partials)
        self._remember(t, w)
        return out, w

    def accumulate(self, y: Tensor, y_mul: float, x_mul: float, partials, sigma, sigma_next, pre=None) -> bool:
        """y <- y * y_mul + self(sigma, sigma_next) * x_mul without writing the increment out (a noise chain's running sum).  ``pre``
        (``hip_lib.FoldPrefix``): the previous item's fold, applied to y first by the same launch when it can be (else by its own).
        False: nothing was done to y, ``pre`` included."""
        if tuple(y.shape) != self.shape or y.device != self.device or y.dtype != torch.float32 or not y.is_contiguous():
            return False
        self(sigma, sigma_next, fold=(y, y_mul, x_mul, partials, pre))
        return True

    def __call__(self, sigma, sigma_next, *, fold=None, partials=None) -> Tensor:
        """The increment; ``partials`` (optional workspace) receives its (sum, sumsq) statistics from the same pass."""
        t0, t1 = float(self.transform(torch.as_tensor(sigma))), float(self.transform(torch.as_tensor(sigma_next)))
        sign = self.sign * (1.0 if t0 <= t1 else -1.0)
        ta, tb = (t0, t1) if t0 <= t1 else (t1, t0)
        ta, tb = self.path.resolve(ta), self.path.resolve(tb)  # (tree mode: the grid points the two times stand for)
        if ta == tb and t0 != t1:
            # two distinct times inside one cell of the tree's grid (closer than its tolerance, 6e-8 of the range at depth 24): the path does
            # not move between them -- torchsde's tree returns W(t1) - W(t0) = 0 there as well
            out, _ = self._emit([], [], want_w=False, fold=fold, partials=partials)
            return out
        if (self.CACHE_POINTS <= 0 and not self.path.tree_depth) or ta == tb:  # (CACHE_POINTS = 0: the path of bridges without kept tensors)
            ids, coefs = self.path.increment(t0, t1)
            out, _ = self._emit(ids, [c * sign for c in coefs], want_w=False, fold=fold, partials=partials)
            return out
        # out = (W(tb) - W(ta)) / sqrt(tb - ta); both times are defined here, the smaller first, whatever is kept
        self.path.define(ta)
        self.path.define(tb)
        scale = sign / math.sqrt(tb - ta)
        wa, wb = self._cached(ta), self._cached(tb)
        if wa is not None and wb is not None:
            out, _ = self._emit([], [], base_b=wb, fb=1.0, prev=wa, scale=scale, want_w=False, fold=fold, partials=partials)
            return out
        if wa is None and wb is None:
            if ta == self.path.t_lo:  # W(t_lo) = 0
                out, _ = self._point(tb, scale=scale, fold=fold, partials=partials)
                return out
            _, wa = self._point(ta, want_out=False)
        if wb is None:
            out, _ = self._point(tb, prev=wa, scale=scale, fold=fold, partials=partials)
            return out
        out, _ = self._point(ta, prev=wb, scale=-scale, fold=fold, partials=partials)  # scale * (W(tb) - W(ta))
        return out


class BrownianNoiseGenerator(NoiseGenerator):
    """py/noise_generation.py:262-286."""

    name = "brownian"

    def __init__(self, x, *args, **kwargs):
        super().__init__(x, *args, **kwargs)
        seed, sigma_min, sigma_max = (self.options.get(k) for k in ("seed", "sigma_min", "sigma_max"))
        if sigma_min is None or sigma_max is None:
            raise ValueError("Brownian noise requires sigma_min and sigma_max")
        self.brownian_tree_ns = BrownianTreeNoiseSampler(x, sigma_min, sigma_max, seed=seed, cpu=self.cpu)

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {"normalized": False}

    def generate(self, *args):
        # the increment's statistics come out of the same pass: the normalisation that usually follows (py/noise.py:245) needs no sweep
        partials = hip_lib.new_partials(self.device)
        return attach_stats(self.brownian_tree_ns(*args, partials=partials), partials)

    accepts_prefix = True  # generate_into(..., pre=): the previous chain item's fold rides in this generator's kernel

    def generate_into(self, y, y_mul, x_mul, partials, *args, pre=None):
        if not self._plain_output():
            return False
        return self.brownian_tree_ns.accumulate(y, y_mul, x_mul, partials, *args, pre=pre)


class WaveletNoiseOctave(NamedTuple):
    octave: int
    height: float
    width: float
    amplitude: float
    total_amplitude: float


class WaveletNoiseGenerator(FramesToChannelsNoiseGenerator):
    """py/noise_generation.py:2204-2327: octaves of (noise - lowpass(noise)) at shrinking resolutions, resampled back to the
    latent size and summed with decaying amplitudes.  The resampling steps (adaptive average pooling down, bilinear up) run on
    the HIP resampler; the octave arithmetic on the blend / axpby kernels."""

    name = "wavelet"
    MIN_DIMS = 4
    MAX_DIMS = 5

    @classmethod
    def ng_params(cls):
        return super().ng_params() | {
            "octave_scale_mode": "adaptive_avg_pool2d", "octave_rescale_mode": "bilinear", "post_octave_rescale_mode": "bilinear",
            "initial_amplitude": 1.0, "persistence": 0.5, "octaves": 4, "octave_height_factor": 0.5, "octave_width_factor": 0.5,
            "height_factor": 2.0, "width_factor": 2.0, "min_height": 4, "min_width": 4, "update_blend": 1.0,
            "update_blend_function": utils.BLENDING_MODES["lerp"], "noise_sampler": None,
        }

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.set_octave_data()

    def set_internal_noise_sampler(self, noise_sampler) -> None:
        self.noise_sampler = noise_sampler

    def set_octave_data(self) -> None:
        height, width = self.get_adjusted_shape()[-2:]
        amplitude, total = self.initial_amplitude, 0.0
        ch, cw = height, width
        data = []
        reverse = self.octaves < 0
        for octave in (reversed(range(abs(self.octaves))) if reverse else range(self.octaves)):
            ch /= self.height_factor**octave
            cw /= self.width_factor**octave
            if (amplitude == 0 or ch < self.min_height or cw < self.min_width or ch * self.octave_height_factor < 1
                    or cw * self.octave_width_factor < 1):
                if reverse and not data:
                    ch, cw = height, width
                    continue
                break
            total += abs(amplitude)
            data.append(WaveletNoiseOctave(octave, ch, cw, amplitude, total))
            amplitude *= self.persistence
        if not data or not total:
            raise ValueError("Unworkable parameters for wavelet noise")
        self.octave_data = tuple(data)

    def _generate_octave(self, *args, shape) -> Tensor:
        height, width = shape[-2:]
        if self.noise_sampler:
            noise = self.noise_sampler(*args)
            utils.pop_stats(noise)
            noise = noise[..., :height, :width].reshape(shape).contiguous()
        else:
            noise = self.rand_like(shape=(*shape[:-2], height, width))
            utils.pop_stats(noise)
        sh, sw = int(max(1, height * self.octave_height_factor)), int(max(1, width * self.octave_width_factor))
        low = utils.scale_samples(utils.scale_samples(noise, sw, sh, mode=self.octave_scale_mode), width=width, height=height,
                                  mode=self.octave_rescale_mode)
        detail = hip_lib.blend("subtract_b", noise, low, 1.0)  # noise - lowpass(noise)
        return self.update_blend_function(noise, detail, self.update_blend)

    def generate(self, *args) -> Tensor:
        shape = self.get_adjusted_shape()
        height, width = shape[-2:]
        result = None
        for od in self.octave_data:
            octave = self._generate_octave(*args, shape=(*shape[:-2], int(od.height), int(od.width)))
            if tuple(octave.shape) != tuple(shape):
                octave = utils.scale_samples(octave, width, height, mode=self.post_octave_rescale_mode)
            result = hip_lib.mul_scalar(octave, od.amplitude) if result is None else hip_lib.axpby_(result, 1.0, octave, od.amplitude)
        total = self.octave_data[-1].total_amplitude
        if total != 0:
            result = hip_lib.div_scalar(result, total, out=result)
        return self.fix_output_frames(result)
