"""Sigma-gated latent operations (API of the reference's ``py/latent_ops.py``); the arithmetic (scale, difference, blend,
noise add) runs through the HIP elementwise kernels."""
from __future__ import annotations

import math
import random
from typing import Sequence

import torch

from .. import hip_lib
from . import utils


class SonarLatentOperation:
    """py/latent_ops.py:15-58."""

    EXTENDED_LATENT_OPERATION = True

    def __init__(self, *, start_sigma: float = math.inf, end_sigma: float = 0.0, op=None):
        self.start_sigma = start_sigma if start_sigma >= 0 else math.inf
        self.end_sigma = end_sigma
        self.op = op

    def enabled(self, sigma=None) -> bool:
        if isinstance(sigma, torch.Tensor):
            sigma = sigma.detach().max().cpu().item()
        return sigma is None or self.end_sigma <= sigma <= self.start_sigma

    def call_op(self, t, *args, op=None, **kwargs):
        op = self.op if op is None else op
        if op is None:
            return t
        if not getattr(op, "EXTENDED_LATENT_OPERATION", False):
            return op(latent=t)
        return op(*args, latent=t, **kwargs)

    def __call__(self, latent, *, sigma=None, **kwargs):
        if not self.enabled(sigma=sigma):
            return latent
        return self.call_op(latent, sigma=sigma, **kwargs)


class SonarLatentOperationAdvanced(SonarLatentOperation):
    """py/latent_ops.py:61-106: blend(t, ops(t * in) [* out] - t) * diff_mul, strength).  ``output_multiplier`` is applied
    only when it equals 1.0 (reference :101-103; reproduced, not fixed)."""

    def __init__(self, *, blend_mode: str, blend_strength: float, input_multiplier: float, output_multiplier: float,
                 difference_multiplier: float, ops: Sequence, op_alt=None, **kwargs):
        super().__init__(**kwargs)
        self.blend_function = utils.BLENDING_MODES[blend_mode]
        self.blend_strength = blend_strength
        self.input_multiplier, self.output_multiplier, self.difference_multiplier = input_multiplier, output_multiplier, difference_multiplier
        self.op_alt = op_alt
        self.ops = ops

    def __call__(self, latent, *, sigma=None, **kwargs):
        t = latent
        if not self.enabled(sigma):
            return t if self.op_alt is None else self.call_op(t, sigma=sigma, op=self.op_alt, **kwargs)
        t32 = utils.as_f32(t)
        output = hip_lib.mul_scalar(t32, self.input_multiplier) if self.input_multiplier != 1.0 else t32
        for op in self.ops:
            output = self.call_op(output, sigma=sigma, op=op, **kwargs)
        scaled = hip_lib.mul_scalar(utils.as_f32(output), self.output_multiplier) if self.output_multiplier == 1.0 else utils.as_f32(output)
        diff = hip_lib.blend("subtract_b", scaled, t32, 1.0)
        if self.difference_multiplier != 1.0:
            hip_lib.scale_noise_(diff, self.difference_multiplier, False, None)
        return self.blend_function(t32, diff, self.blend_strength)


class SonarLatentOperationNoise(SonarLatentOperation):
    """py/latent_ops.py:109-186: noise (optionally * sigma) + latent."""

    def __init__(self, *args, custom_noise, scale_to_sigma: bool = False, cpu_noise: bool = False, normalize: bool = True,
                 lazy_noise_sampler: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.custom_noise = custom_noise
        self.normalize, self.scale_to_sigma, self.cpu_noise = normalize, scale_to_sigma, cpu_noise
        self.lazy_noise_sampler = lazy_noise_sampler
        self.noise_sampler = None
        self.cache_id = None

    def __call__(self, latent, *, sigma=None, **kwargs):
        t = latent
        if not self.enabled(sigma):
            return t
        if isinstance(sigma, float):
            sigma = torch.full((1,), sigma)
        make_ns = not self.lazy_noise_sampler or self.noise_sampler is None
        sigma_min = sigma_max = sigma_next = None
        sample_sigmas = kwargs.get("raw_args", {}).get("model_options", {}).get("transformer_options", {}).get("sample_sigmas")
        if sample_sigmas is not None and sigma is not None:
            sig_host = sigma.detach().max().cpu()
            ss = sample_sigmas.detach().cpu()
            step = int((ss - sig_host).abs().argmin())
            if ss[step].max().item() == sig_host.item() and step + 1 < len(ss):
                sigma_next = ss[step + 1]
        if self.lazy_noise_sampler and not make_ns:
            cache_id = id(sample_sigmas) if isinstance(sample_sigmas, torch.Tensor) else None
            make_ns = cache_id is None or cache_id != self.cache_id
            self.cache_id = cache_id
            if make_ns and sample_sigmas is not None:
                pos = sample_sigmas[sample_sigmas > 0]
                sigma_min = pos.min().item() if pos.numel() else 0.0
                sigma_max = sample_sigmas.max().item()
        t32 = utils.as_f32(t)
        if make_ns:
            ns = self.custom_noise.make_noise_sampler(t32, sigma_min=sigma_min, sigma_max=sigma_max, normalized=self.normalize,
                                                      seed=torch.randint(1, 1 << 31, (), device="cpu").item(), cpu=self.cpu_noise)
        else:
            ns = self.noise_sampler
        if make_ns and self.lazy_noise_sampler:
            self.noise_sampler = ns
        noise = ns(sigma, sigma if sigma_next is None else sigma_next)
        utils.pop_stats(noise)
        scale = float(sigma.detach().max()) if (self.scale_to_sigma and sigma is not None) else 1.0
        return hip_lib.axpby_(noise, scale, t32, 1.0)  # noise * sigma + t in one kernel


class SonarLatentOperationSetSeed(SonarLatentOperation):
    """py/latent_ops.py:189-209."""

    def __init__(self, *args, seed: int, restore_rng_state: bool, **kwargs):
        super().__init__(*args, **kwargs)
        self.seed = seed
        self.restore_rng_state = restore_rng_state

    def __call__(self, *args, **kwargs):
        saved = (random.getstate(), torch.random.get_rng_state()) if self.restore_rng_state else None
        try:
            torch.manual_seed(self.seed)
            random.seed(self.seed)
            return super().__call__(*args, **kwargs)
        finally:
            if saved is not None:
                random.setstate(saved[0])
                torch.random.set_rng_state(saved[1])
