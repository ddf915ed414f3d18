"""Power-law / band-pass spectral noise (API of the reference's ``py/nodes/powernoise.py``).

Setup (host, once per sampler): the rfft-space gain ``filter[H, W/2+1]`` is built with the reference's
exact recipe (it is *not* a closed-form 1/f: oversampled grid + bilinear resample), and the C x C
channel mixer is factorised.  Per call (device): one fused kernel draws (or replays) the complex
half-spectrum, multiplies by the filter, runs the LDS-resident C2R inverse FFT and accumulates the
normaliser's statistics (``sonar_power_irfft2_f32``); mixing and normalisation are HIP kernels too.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from ... import hip_lib
from ..noise import CustomNoiseItemBase
from ..noise_generation import BrownianTreeNoiseSampler, DeviceRNG, current_batch_offset
from .. import utils
from ..utils import attach_stats, pop_stats, scale_noise


class ChannelMixer:
    """py/nodes/powernoise.py:56-104: row-normalised LDL factor of the channel correlation matrix."""

    def __init__(self, channel_count, common_mode, channel_correlation):
        self.channel_count = channel_count
        self.common_mode = common_mode
        self.channel_correlation = channel_correlation
        self.mixer = self.build() if common_mode is not None else None

    def build(self) -> Tensor:
        c, cm = self.channel_count, self.common_mode
        pairs = c * (c - 1) // 2
        given = self.channel_correlation[:pairs]
        lower = torch.cat((given * cm, torch.full((pairs - given.numel(),), cm)))
        corr = torch.eye(c).index_put_(tuple(torch.tril_indices(c, c, offset=-1)), lower)
        corr += corr.tril(-1).mT
        ld = torch.linalg.ldl_factor(corr).LD
        diag = torch.diagonal_copy(ld)
        torch.diagonal(ld)[:] = 1.0
        ld *= diag.clamp_min(0).sqrt().unsqueeze(0)
        ld /= ld.norm(dim=1, keepdim=True)
        return ld

    @property
    def is_identity(self) -> bool:
        return self.mixer is None or bool(torch.equal(self.mixer.cpu(), torch.eye(self.channel_count)))

    def to(self, *args, **kwargs):
        if self.mixer is not None:
            self.mixer = self.mixer.to(*args, **kwargs)
        return self

    def apply(self, noise, shape, copy=False, partials=None):
        if self.mixer is None:
            return noise if not copy else noise.clone()
        if shape[1] != self.channel_count:
            raise ValueError("Channel count mismatch")
        return hip_lib.channel_mix(noise, self.mixer.to(noise.device, torch.float32).contiguous(), partials)

    __call__ = apply


class PowerFilter:
    """py/nodes/powernoise.py:107-266: band-pass * 1/f**alpha gain in rfft2 layout."""

    _FIELDS = ("min_freq", "max_freq", "stretch", "rotate", "pnorm", "alpha", "scale", "rel_bw", "oversample", "compose_mode")

    def __init__(self, *, min_freq=0.0, max_freq=0.7071, stretch=1.0, rotate=0.0, pnorm=2.0, alpha=0.0, scale=1.0, rel_bw=0.125,
                 oversample=4, compose_with: "PowerFilter | None" = None, compose_mode="max"):
        self.min_freq = min_freq
        self.max_freq = max(max_freq, min_freq)
        self.stretch, self.rotate, self.pnorm, self.alpha = stretch, rotate, pnorm, alpha
        self.scale, self.rel_bw, self.oversample = scale, rel_bw, oversample
        self.compose_with, self.compose_mode = compose_with, compose_mode

    def clone(self):
        args = {k: getattr(self, k) for k in self._FIELDS}
        args["compose_with"] = None if self.compose_with is None else self.compose_with.clone()
        return self.__class__(**args)

    @classmethod
    def compose(cls, a, b, compose_mode="max"):
        if a.shape != b.shape:
            raise ValueError("Filter compose size mismatch!")
        op = {"max": torch.max, "min": torch.min, "add": torch.add, "sub": torch.sub, "mul": torch.mul}.get(compose_mode, torch.max)
        return op(a, b).clamp_(min=0.0)

    @classmethod
    def normalize(cls, op, shape, mix=1.0, normalization_factor=1.0):
        """Unit-RMS gain (lerped by normalization_factor), then lerp with a flat response by ``mix``."""
        height, width = shape[-2:]
        if mix < 1.0:
            flat = torch.ones(1, 1, height, width // 2 + 1)
            if mix <= 0.0:
                return flat
        if normalization_factor != 0:
            op *= torch.lerp(torch.scalar_tensor(1.0), 1.0 / op.square().mean().sqrt(), normalization_factor)
        if mix < 1.0:
            op = torch.lerp(flat, op, mix, out=op)
        return op

    def _radial_distance(self, height, bins, oversample):
        # oversampled fftshift(rfft2freq) grid held as complex numbers purely for 2-D rotate/stretch
        cols = torch.linspace(0, 0.5, oversample * bins)
        rows = torch.linspace(-(height // 2) / height, ((height - 1) // 2) / height, oversample * height).unsqueeze(1)
        grid = torch.complex(cols, rows)
        if abs(self.rotate) >= 1e-3:
            grid *= torch.exp(1.0j * torch.deg2rad(torch.scalar_tensor(self.rotate)))
        if self.stretch > 1.0:
            grid.real *= self.stretch
        else:
            grid.imag *= 1.0 / self.stretch
        if abs(self.pnorm - 2.0) < 1e-3:
            return grid.abs()
        return torch.view_as_real(grid).abs().pow(self.pnorm).sum(-1).pow(1.0 / self.pnorm)

    def build(self, shape, override_oversample=None, composed=True):
        oversample = self.oversample if override_oversample is None else override_oversample
        height, width = shape[-2:]
        bins = width // 2 + 1
        dist = self._radial_distance(height, bins, oversample)
        gain = torch.empty_like(dist)
        ge_min, lt_max = dist >= self.min_freq, dist < self.max_freq
        inside = ge_min & lt_max
        gain[inside] = dist[inside].pow(-self.alpha)
        # Gaussian roll-off outside the band edges
        hi = ~lt_max
        gain[hi] = math.pow(self.max_freq, -self.alpha) * torch.exp(-(dist[hi] - self.max_freq).square() / (self.rel_bw * self.max_freq) ** 2)
        if self.min_freq > 0.0:
            lo = ~ge_min
            gain[lo] = math.pow(self.min_freq, -self.alpha) * torch.exp(-(dist[lo] - self.min_freq).square() / (self.rel_bw * self.min_freq) ** 2)
        gain = torch.nn.functional.interpolate(gain[None, None, ...], (height, bins), mode="bilinear", align_corners=True)
        gain = gain.roll(-(height // 2), -2)  # undo the fftshift along rows
        if self.alpha > 0:
            gain[..., 0, 0] = 0  # the DC gain would be infinite
        if self.scale != 1.0:
            gain *= self.scale
        if composed and self.compose_with is not None:
            return self.compose(gain, self.compose_with.build(shape, override_oversample=override_oversample), self.compose_mode)
        return gain


def _device_irfft2_fallback_error(h, w):
    return hip_lib.SonarHipError(
        f"power noise: plane {h}x{w} is beyond the spectral kernels (LDS-resident FFTs, direct passes up to 2048 x 2048); "
        "there is no CPU fallback"
    )


class PowerNoiseItem(CustomNoiseItemBase):
    """py/nodes/powernoise.py:297-408."""

    stats_lookahead = True  # device-drawn, normalised calls: see hip_lib.power_noise(lookahead=...)

    def __init__(self, factor, *, channel_correlation, power_filter=None, **kwargs):
        if isinstance(channel_correlation, str):
            vals = tuple(float(v) for v in (v.strip() for v in channel_correlation.split(",")) if v)
            channel_correlation = torch.tensor(vals, device="cpu", dtype=torch.float)
        if power_filter is None:
            fargs = {k: kwargs.pop(k) for k in ("min_freq", "max_freq", "stretch", "rotate", "pnorm", "alpha") if k in kwargs}
            power_filter = PowerFilter(**fargs)
        super().__init__(factor, power_filter=power_filter, channel_correlation=channel_correlation, **kwargs)

    def make_filter(self, shape, oversample=None):
        with utils.host_setup_threads():  # a few CPU threads: see there
            return PowerFilter.normalize(self.power_filter.build(shape, override_oversample=oversample), shape, mix=self.mix,
                                         normalization_factor=getattr(self, "filter_norm_factor", 1.0))

    def make_noise_sampler_internal(self, x: Tensor, noise_sampler, filter_rfft, normalized=True):
        """``noise_sampler`` returns a complex64 half-spectrum (replay) or None (draw on device)."""
        shape = tuple(x.shape)
        h, w = shape[-2:]
        if x.ndim < 4:
            raise hip_lib.SonarHipError("power noise: [B, C, ..., H, W] latents (4 or more dimensions)")
        if not hip_lib.power_supported(h, w):
            raise _device_irfft2_fallback_error(h, w)
        device = x.device
        filt = filter_rfft.to(device, torch.float32).reshape(h, w // 2 + 1).contiguous()
        with utils.host_setup_threads():  # the 4 x 4 LDL factorisation goes through LAPACK: its thread pool too (see there)
            mixer = ChannelMixer(shape[1], self.common_mode, self.channel_correlation)
            identity = mixer.is_identity
        mixer.to(device)
        planes_per_latent = math.prod(shape[1:-2])  # 5-D (video) latents: every [H, W] slice is a plane, as in the reference's irfft2
        # a sampler's calls take consecutive stream ids: each call leaves the next one's statistics (`stats_lookahead = False` on the item
        # or the class: every call launches its own statistics pass)
        lookahead = hip_lib.PowerLookahead() if self.stats_lookahead else None

        def sampler(sigma, sigma_next):
            z = noise_sampler(sigma, sigma_next)
            partials = hip_lib.new_partials(device)
            if self.time_brownian:
                # real noise in: rfft2 -> x filter -> irfft2, forward and inverse both LDS-resident (py/nodes/powernoise.py:356-366)
                pop_stats(z)
                noise = hip_lib.spectral_filter(z.to(device).contiguous(), filt, partials if (identity and normalized) else None)
            elif z is None:
                seed, stream = DeviceRNG.take()
                offs = current_batch_offset() * planes_per_latent
                if identity and normalized:
                    # draw + filter + FFT + normalise with a single write of the tensor
                    return hip_lib.power_noise(filt, shape, seed=seed, stream_id=stream, plane_offset=offs, factor=self.factor,
                                               lookahead=lookahead)
                noise = hip_lib.power_irfft2(None, filt, shape, seed=seed, stream_id=stream, plane_offset=offs,
                                             partials=partials if (identity and normalized) else None)
            else:
                noise = hip_lib.power_irfft2(z.to(device).contiguous(), filt, shape, partials=partials if (identity and normalized) else None)
            if not identity:
                noise = mixer(noise, shape, partials=partials)
            if defer_factor:
                return noise
            # the statistics were written by the FFT kernel (identity mixer, normalising call) or by the mixer; otherwise the
            # workspace is untouched memory and must not ride along
            have_stats = (identity and normalized) or not identity
            return scale_noise(attach_stats(noise, partials if have_stats else None), self.factor, normalized=normalized)

        defer_factor = False
        if not normalized:
            def unscaled(sigma, sigma_next):  # a chain folds the factor into its accumulation kernel (CustomNoiseChain)
                nonlocal defer_factor
                defer_factor = True
                try:
                    return sampler(sigma, sigma_next), float(self.factor)
                finally:
                    defer_factor = False

            sampler.unscaled = unscaled
        if getattr(noise_sampler, "draws_on_device", False) and not self.time_brownian:
            # a device-drawn call depends on nothing but the RNG position: traced once, then one foreign call per step (hip_lib.Planned)
            planned = hip_lib.Planned(sampler, take=DeviceRNG.take, rewind=DeviceRNG.rewind,
                                      guards=(current_batch_offset, lambda: (self.factor, self.time_brownian)))
            planned.plan_static = True
            return planned
        return sampler

    def make_noise_sampler(self, x: Tensor, sigma_min, sigma_max, *, seed, cpu: bool = True, normalized=True):
        shape = x.shape
        filter_rfft = self.make_filter(shape)
        if self.time_brownian:
            # time-correlated mode (py/nodes/powernoise.py:383-393): a real Brownian-interval field -> rfft2 -> filter -> irfft2
            if sigma_min is None:
                raise ValueError("time correlated brownian mode is valid only for stochastic samplers")
            brownian = BrownianTreeNoiseSampler(x, sigma_min, sigma_max, seed=seed, cpu=cpu)
            return self.make_noise_sampler_internal(x, brownian, filter_rfft, normalized=normalized)
        if cpu:
            def draw(_s, _sn):  # py/nodes/powernoise.py:396-402
                return torch.randn((*shape[:-1], shape[-1] // 2 + 1), dtype=torch.complex64, device="cpu")
        else:
            def draw(_s, _sn):
                return None

            draw.draws_on_device = True
        return self.make_noise_sampler_internal(x, draw, filter_rfft, normalized=normalized)


class PowerFilterNoiseItem(PowerNoiseItem):
    """py/nodes/powernoise.py:462-504: a power filter applied to ANOTHER noise chain (node passes time_brownian=True)."""

    def __init__(self, factor, *, noise, normalize_noise, normalize_result, **kwargs):
        super().__init__(factor, noise=noise.clone(), normalize_noise=normalize_noise, normalize_result=normalize_result, **kwargs)

    def clone_key(self, k):
        if k == "noise":
            return self.noise.clone()
        return super().clone_key(k)

    def make_noise_sampler(self, x: Tensor, sigma_min, sigma_max, *, seed, cpu: bool = True, normalized=True):
        normalize_noise = self.get_normalize("normalize_noise", False)
        normalize_result = self.get_normalize("normalize_result", normalized)
        filter_rfft = self.make_filter(x.shape)
        noise_sampler = self.noise.make_noise_sampler(x, sigma_min, sigma_max, seed, cpu, normalized=normalize_noise)
        return self.make_noise_sampler_internal(x, noise_sampler, filter_rfft, normalized=normalize_result)
