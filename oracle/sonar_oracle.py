"""CPU oracle for the Sonar hot path — TEST INFRASTRUCTURE ONLY.

A restatement, in plain PyTorch-CPU tensor ops, of the reference's algorithms on the path named by
BASELINE.json (procedural noise, whole-tensor normalisation, noise-chain composition, momentum
sampler steps).  Every function cites the reference `file:line` it follows (paths relative to
the reference root).  The 2-D DWT restatement lives in ``oracle/dwt_oracle.py``.

Parity pin: ``tests/golden/*.npz`` hold inputs (including the captured base random draws) and outputs
produced by the *real* reference imported in the build container (``tests/golden/make_golden.py``,
via ``oracle/ref_import.py``); ``tests/test_oracle_golden.py`` checks this file against them bit-for-bit
where the op sequence is identical.  DWT arithmetic is third-party in the reference
(pytorch_wavelets, unpinned, absent) and is pinned to PyWavelets 1.1.1 outputs instead.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``comfyui-sonar_amd/``) never does.
"""
from __future__ import annotations

import math
from typing import Callable, NamedTuple, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------------------
# blends and normalisation


def blend(mode: str, a: Tensor, b: Tensor, t) -> Tensor:
    """py/utils.py:17-21 (BLENDING_MODES)."""
    if mode == "lerp":
        return torch.lerp(a, b, t)
    if mode == "inject":
        return (b * t).add_(a)
    if mode == "subtract_b":
        return a - b * t
    raise KeyError(mode)


def scale_noise(noise: Tensor, factor: float = 1.0, *, normalized: bool = True, threshold_std_devs: float = 2.5,
                normalize_dims: Optional[tuple] = None, decisions: Optional[dict] = None) -> Tensor:
    """py/utils.py:85-106.  In place where the reference is in place.  ``decisions`` (optional dict) receives
    the two data-dependent branch outcomes."""
    n = noise.numel()
    if not normalized or n == 0:
        return noise.mul_(factor) if factor != 1 else noise
    if normalize_dims is not None:
        sd = noise.std(dim=normalize_dims, keepdim=True)
        noise = noise / sd
        return noise.sub_(noise.mean(dim=normalize_dims, keepdim=True)).mul_(factor)
    mean = noise.mean().item()
    sd = noise.std().item()
    thr = threshold_std_devs / math.sqrt(n)
    sub, div = abs(mean) > thr, abs(1.0 - sd) > thr
    if decisions is not None:
        decisions.update(mean=mean, std=sd, sub=sub, div=div, threshold=thr)
    if sub:
        noise -= mean
    if div:
        noise /= sd
    return noise.mul_(factor) if factor != 1 else noise


def normalize_to_scale(t: Tensor, lo: float, hi: float, *, dim=(-3, -2, -1), eps: float = 1e-07) -> Tensor:
    """py/utils.py:452-470."""
    mn, mx = t.amin(dim=dim, keepdim=True), t.amax(dim=dim, keepdim=True)
    out = t - mn
    out /= (mx - mn).add_(eps)
    return out.mul_(hi - lo).add_(lo).clamp_(lo, hi)


def resize(t: Tensor, width: int, height: int, mode: str) -> Tensor:
    """py/utils.py:58-67 (scale_samples) with ComfyUI's common_upscale restated as F.interpolate."""
    if mode == "adaptive_avg_pool2d":
        return F.adaptive_avg_pool2d(t, (height, width))
    return F.interpolate(t, size=(height, width), mode=mode)


# --------------------------------------------------------------------------------------------------
# base generators: random draws in the reference's order (global CPU generator)


def draw_gaussian(shape) -> Tensor:
    """py/noise_generation.py:133-155, 252-260."""
    return torch.randn(*shape, dtype=torch.float32)


def uniform_noise(u: Tensor, sub_fac=0.5, mul_fac=3.46, mean_fac=0.0) -> Tensor:
    """py/noise_generation.py:496-514 applied to a U[0,1) draw."""
    return u.clone().sub_(sub_fac).mul_(mul_fac).add_(mean_fac)


# --------------------------------------------------------------------------------------------------
# Perlin (row P)


class PerlinDraws(NamedTuple):
    base: Tensor          # U[0,1)   [B,C,H,W]
    angles: tuple         # iters x U[0,2pi) [C,H+1,W+1]


def draw_perlin(shape, iterations: int = 2) -> PerlinDraws:
    """Draw order of py/noise_generation.py:478-493 (base first, then one lattice per iteration, :465-469)."""
    b, c, h, w = shape
    base = torch.rand(b, c, h, w, dtype=torch.float32)
    angles = tuple(torch.empty(c, h + 1, w + 1, dtype=torch.float32).uniform_(to=2.0 * math.pi) for _ in range(iterations))
    return PerlinDraws(base, angles)


def perlin_term(angle: Tensor, blend_mode: str = "lerp") -> Tensor:
    """py/noise_generation.py:352-421,465-476 for grid == output size: every cell is one pixel sampled at
    (0.5, 0.5), smooth_step(0.5) = 0.5; corners in unfold(2,2) order TL, TR, BL, BR (:320-337)."""
    c, gh, gw = angle.shape
    h, w = gh - 1, gw - 1
    gx, gy = torch.cos(angle), torch.sin(angle)

    def corner(dy, dx):
        return gx[:, dy:dy + h, dx:dx + w], gy[:, dy:dy + h, dx:dx + w]

    def dot(g, px, py):
        return g[0] * px + g[1] * py

    half = torch.full((1, 1, 1), 0.5, dtype=angle.dtype)
    step = half * half * (3.0 - 2.0 * half)  # smooth_step, :339-350
    d_tl = dot(corner(0, 0), 0.5, 0.5)
    d_tr = dot(corner(0, 1), 0.5 - 1.0, 0.5)
    d_bl = dot(corner(1, 0), 0.5, 0.5 - 1.0)
    d_br = dot(corner(1, 1), 0.5 - 1.0, 0.5 - 1.0)
    row0 = blend(blend_mode, d_tl, d_tr, step)
    row1 = blend(blend_mode, d_bl, d_br, step)
    return blend(blend_mode, row0, row1, step)


def perlin_noise(draws: PerlinDraws, div_fac: float = 2.0, blend_mode: str = "lerp") -> Tensor:
    """py/noise_generation.py:478-493: U/div_fac plus the lattice terms, each broadcast over the batch."""
    noise = draws.base.clone().div_(div_fac)
    for angle in draws.angles:
        noise += perlin_term(angle, blend_mode)
    return noise


# --------------------------------------------------------------------------------------------------
# Pyramid (row Y)


class PyramidDraws(NamedTuple):
    base: Tensor      # N(0,1) [B,C,H,W]
    rs: tuple         # the r scalars, drawn before each level
    levels: tuple     # N(0,1) [B,C,h_i,w_i]


def pyramid_sizes(h: int, w: int, rs: Sequence[float]) -> list:
    """py/noise_generation.py:626-648: cumulative shrink w,h = max(1, int(w / r**i)); stops after the
    first level with w == 1 or h == 1."""
    out = []
    for i, r in enumerate(rs):
        w, h = max(1, int(w / (r**i))), max(1, int(h / (r**i)))
        out.append((h, w))
        if w == 1 or h == 1:
            break
    return out


def draw_pyramid(shape, iterations: int = 10) -> PyramidDraws:
    """RNG order of py/noise_generation.py:620-648: base, then per level rand(1) followed by randn(level)."""
    b, c, h, w = shape
    base = torch.randn(b, c, h, w, dtype=torch.float32)
    rs, levels = [], []
    cw, ch = w, h
    for i in range(iterations):
        r = torch.rand(1).item() * 2 + 2
        cw, ch = max(1, int(cw / (r**i))), max(1, int(ch / (r**i)))
        rs.append(r)
        levels.append(torch.randn(b, c, ch, cw, dtype=torch.float32))
        if cw == 1 or ch == 1:
            break
    return PyramidDraws(base, tuple(rs), tuple(levels))


def pyramid_noise(draws: PyramidDraws, discount: float = 0.7, upscale_mode: str = "bilinear") -> Tensor:
    """py/noise_generation.py:620-649."""
    noise = draws.base.clone()
    h, w = noise.shape[-2:]
    for i, lvl in enumerate(draws.levels):
        noise += resize(lvl, w, h, upscale_mode).mul_(discount**i)
    return noise


# --------------------------------------------------------------------------------------------------
# Power-law rFFT noise (row PW)


def power_filter_build(shape, *, min_freq=0.0, max_freq=0.7071, stretch=1.0, rotate=0.0, pnorm=2.0, alpha=0.0,
                       scale=1.0, rel_bw=0.125, oversample=4) -> Tensor:
    """py/nodes/powernoise.py:189-266 (PowerFilter.build, without compose): band-pass * 1/f^alpha gain on an
    oversampled half-plane frequency grid, bilinear-resampled (align_corners=True) to H x (W/2+1)."""
    max_freq = max(max_freq, min_freq)  # :124
    height, width = shape[-2:]
    bins = width // 2 + 1
    col = torch.linspace(0, 0.5, oversample * bins)
    row = torch.linspace(-(height // 2) / height, ((height - 1) // 2) / height, oversample * height).unsqueeze(1)
    grid = torch.complex(col, row)  # complex only as a 2-D vector, :200-211
    if abs(rotate) >= 1e-3:
        grid *= torch.exp(1.0j * torch.deg2rad(torch.scalar_tensor(rotate)))
    if stretch > 1.0:
        grid.real *= stretch
    else:
        grid.imag *= 1.0 / stretch
    if abs(pnorm - 2.0) < 1e-3:
        dist = grid.abs()
    else:
        dist = torch.view_as_real(grid).abs().pow(pnorm).sum(-1).pow(1.0 / pnorm)
    gain = torch.empty_like(dist)
    above_min = dist >= min_freq
    below_max = dist < max_freq
    band = above_min & below_max
    gain[band] = dist[band].pow(-alpha)
    over = ~below_max
    gain[over] = math.pow(max_freq, -alpha) * torch.exp(-(dist[over] - max_freq).square() / (rel_bw * max_freq) ** 2)
    if min_freq > 0.0:
        under = ~above_min
        gain[under] = math.pow(min_freq, -alpha) * torch.exp(-(dist[under] - min_freq).square() / (rel_bw * min_freq) ** 2)
    gain = F.interpolate(gain[None, None, ...], (height, bins), mode="bilinear", align_corners=True)
    gain = gain.roll(-(height // 2), -2)  # ifftshift along rows
    if alpha > 0:
        gain[..., 0, 0] = 0
    if scale != 1.0:
        gain *= scale
    return gain


def power_filter_compose(a: Tensor, b: Tensor, mode: str = "max") -> Tensor:
    """py/nodes/powernoise.py:156-167."""
    fn = {"max": torch.max, "min": torch.min, "add": torch.add, "sub": torch.sub, "mul": torch.mul}.get(mode, torch.max)
    return fn(a, b).clamp_(min=0.0)


def power_filter_normalize(op: Tensor, shape, mix: float = 1.0, normalization_factor: float = 1.0) -> Tensor:
    """py/nodes/powernoise.py:169-187."""
    height, width = shape[-2:]
    bins = width // 2 + 1
    if mix < 1.0:
        flat = torch.ones(1, 1, height, bins)
        if mix <= 0.0:
            return flat
    if normalization_factor != 0:
        op *= torch.lerp(torch.scalar_tensor(1.0), 1.0 / op.square().mean().sqrt(), normalization_factor)
    if mix < 1.0:
        op = torch.lerp(flat, op, mix, out=op)
    return op


def channel_mixer(channels: int, common_mode: Optional[float], correlation: Tensor) -> Optional[Tensor]:
    """py/nodes/powernoise.py:63-90 (ChannelMixer.build): LDL factor of the channel correlation matrix,
    row-normalised."""
    if common_mode is None:
        return None
    c = channels
    count = c * (c - 1) // 2
    corr = correlation[:count]
    corr = torch.cat((corr * common_mode, torch.full((count - corr.numel(),), common_mode)))
    m = torch.eye(c).index_put_(tuple(torch.tril_indices(c, c, offset=-1)), corr)
    m += m.tril(-1).mT
    m = torch.linalg.ldl_factor(m).LD
    d = torch.diagonal_copy(m)
    torch.diagonal(m)[:] = 1.0
    m *= d.clamp_min(0).sqrt().unsqueeze(0)
    m /= m.norm(dim=1, keepdim=True)
    return m


def draw_power(shape) -> Tensor:
    """py/nodes/powernoise.py:396-402: complex64 normal draw of the half spectrum."""
    return torch.randn((*shape[:-1], shape[-1] // 2 + 1), dtype=torch.complex64)


def power_noise(z: Tensor, filt: Tensor, shape, mixer: Optional[Tensor] = None, factor: float = 1.0,
                normalized: bool = True, decisions: Optional[dict] = None, pre_norm: Optional[list] = None) -> Tensor:
    """py/nodes/powernoise.py:366-377 (sampler of make_noise_sampler_internal, time_brownian=False)."""
    noise = torch.fft.irfft2(z.clone().mul_(filt), s=tuple(shape[-2:]), norm="ortho")
    if mixer is not None:
        b, c, h, w = shape
        noise = (mixer @ noise.swapaxes(0, 1).reshape(c, -1)).reshape(c, b, h, w).swapaxes(1, 0)
    if pre_norm is not None:
        pre_norm.append(noise.clone())
    return scale_noise(noise, factor, normalized=normalized, decisions=decisions)


def spectral_filter(noise: Tensor, filt: Tensor) -> Tensor:
    """py/nodes/powernoise.py:368-375 with time_brownian / PowerFilterNoiseItem: rfft2 -> *filter -> irfft2."""
    spec = torch.fft.rfft2(noise, norm="ortho")
    return torch.fft.irfft2(spec.mul_(filt), s=tuple(noise.shape[-2:]), norm="ortho")


# --------------------------------------------------------------------------------------------------
# composition (rows C, S)


def chain_noise(raw_items: Sequence[Tensor], factors: Sequence[float], normalized: bool = True) -> Tensor:
    """py/noise.py:137-196 + :249-257: every item is produced un-normalised and scaled by its own factor
    (NoiseSampler.__call__), the sum is normalised once with the chain factor sum(|f_i|)."""
    total = None
    for raw, f in zip(raw_items, factors):
        item = scale_noise(raw.clone(), f, normalized=False)
        total = item if total is None else total.add_(item)
    return scale_noise(total, sum(abs(f) for f in factors), normalized=normalized)


def composite_noise(dst: Tensor, src: Tensor, mask: Tensor, factor: float, normalize_result: bool) -> Tensor:
    """py/noise.py:507-531; dst/src already carry their own normalisation; mask is [B,1,H,W]."""
    inv = torch.ones_like(mask) - mask
    a = dst.clone().mul_(inv)
    b = src.clone().mul_(mask)
    return scale_noise(a.add_(b), factor, normalized=normalize_result)


def blended_noise(n1: Tensor, n2: Tensor, weight, blend_mode: str, factor: float, normalize: bool) -> Tensor:
    """py/noise.py:1389-1405; weight is a (1,) tensor or a mask-derived tensor."""
    return scale_noise(blend(blend_mode, n1, n2, weight), factor, normalized=normalize)


def blend_mask_weight(mask_noise: Tensor, pct: float) -> Tensor:
    """py/noise.py:1395-1398."""
    return (normalize_to_scale(mask_noise, 0.0, 1.0) + pct).clamp_(0.0, 1.0)


# --------------------------------------------------------------------------------------------------
# momentum samplers (rows M, M2)


class MomentumCfg(NamedTuple):
    """The SonarConfig fields the recurrence uses, py/sonar.py:46-62."""
    momentum: float = 0.95
    momentum_hist: float = 0.75
    direction: float = 1.0
    momentum_start_step: int = 0
    momentum_end_step: int = 9999
    always_update_history: bool = True
    mode: str = "NEW"            # CLASSIC | NEW | DENOISED
    init: str = "ZERO"           # ZERO | RAND | SAMPLE | SAMPLE_NORM
    rand_init_noise_multiplier: float = 1.0
    blend_mode: str = "lerp"
    momentum_blend_mode: Optional[str] = None
    history_blend_mode: Optional[str] = None


def ancestral_step(sigma_from, sigma_to, eta: float = 1.0):
    """k-diffusion get_ancestral_step (ComfyUI, un-vendored; SURVEY.md §8c restates the formula)."""
    if not eta:
        return sigma_to, 0.0
    up = min(sigma_to, eta * (sigma_to**2 * (sigma_from**2 - sigma_to**2) / sigma_from**2) ** 0.5)
    down = (sigma_to**2 - up**2) ** 0.5
    return down, up


class MomentumState:
    """History-carrying recurrence of py/sonar.py:169-320, one tensor op per reference tensor op."""

    def __init__(self, cfg: MomentumCfg, rand_init: Optional[Callable[[], Tensor]] = None):
        self.cfg = cfg
        self.h: Optional[Tensor] = None
        self.rand_init = rand_init
        self.mblend = cfg.momentum_blend_mode or cfg.blend_mode
        self.hblend = cfg.history_blend_mode or cfg.blend_mode
        d, mh = cfg.direction, cfg.momentum_hist
        # py/sonar.py:208-219
        self.ratios = (mh, 1.0 + abs(d) * (1 - mh) if d < 0 else 2.0 - d, d)

    def check_step(self, step: int, *, is_history: bool = False) -> bool:
        c = self.cfg
        if is_history and c.always_update_history:
            return True
        return c.momentum_start_step <= step <= c.momentum_end_step

    def init_hist(self, x, denoised, sigma, step):
        """py/sonar.py:169-206."""
        c = self.cfg
        if self.h is not None or not self.check_step(step, is_history=True):
            return
        src = x if c.mode != "DENOISED" else denoised
        if c.init == "SAMPLE":
            self.h = src
        elif c.init == "SAMPLE_NORM":
            self.h = src / sigma
        elif c.init == "RAND":
            self.h = self.rand_init()
            if c.rand_init_noise_multiplier != 1:
                self.h *= c.rand_init_noise_multiplier

    def update_hist(self, v, step):
        """py/sonar.py:227-236."""
        if self.cfg.momentum_hist == 1 or not self.check_step(step, is_history=True):
            return
        hr, hs, ms = self.ratios
        self.h = v if self.h is None else blend(self.hblend, v * ms, self.h * hs, hr)

    def mix(self, history, item, sigma, *, is_denoised=False):
        """py/sonar.py:238-260 (blend weight is always cfg.momentum for the d-path, see M2 note)."""
        c = self.cfg
        if c.momentum == 1 or history is None or (c.mode == "DENOISED") != is_denoised:
            return item
        return blend(self.mblend, history * sigma if is_denoised else history, item, c.momentum)

    def momentum_denoised(self, x, denoised, sigma, step):
        """py/sonar.py:262-283."""
        out = self.mix(self.h, denoised, sigma, is_denoised=True)
        self.init_hist(x, denoised, sigma, step)
        self.update_hist(denoised / sigma, step)
        return out if self.check_step(step) else denoised

    def momentum_d(self, x, denoised, sigma, step, *, gate_momentum=None, d=None):
        """py/sonar.py:285-307; ``gate_momentum`` only feeds the ``== 1`` early-out (:298-303)."""
        c = self.cfg
        gate = c.momentum if gate_momentum is None else gate_momentum
        d = (x - denoised) / sigma if d is None else d
        if gate == 1 or c.mode == "DENOISED":
            return d
        md = self.mix(self.h, d, sigma)
        self.init_hist(x, denoised, sigma, step)
        self.update_hist(d if c.mode == "NEW" else md, step)
        return md if self.check_step(step) else d

    def euler_step(self, step, x, denoised, sigma, sigma_down):
        """py/sonar.py:309-320."""
        dt = sigma_down - sigma
        den_m = self.momentum_denoised(x, denoised, sigma, step)
        md = self.momentum_d(x, den_m, sigma, step)
        return (md * dt).add_(x)


def sonar_euler(model, x, sigmas, cfg: MomentumCfg, *, ancestral=False, eta=1.0, s_noise=1.0, noise_fn=None,
                rand_init=None, trace: Optional[list] = None):
    """py/sonar.py:460-526 (Euler) and :541-623 (ancestral)."""
    st = MomentumState(cfg, rand_init)
    s_in = x.new_ones((x.shape[0],))
    for i in range(len(sigmas) - 1):
        sigma, sigma_next = sigmas[i], sigmas[i + 1]
        down, up = ancestral_step(sigma, sigma_next, eta) if ancestral else (sigma_next, 0.0)
        den = model(x, sigma * s_in)
        x = st.euler_step(i, x, den, sigma, down)
        if ancestral and sigma_next > 0:
            x = x + noise_fn(sigma, sigma_next) * (s_noise * up)
        if trace is not None:
            trace.append((x.clone(), None if st.h is None else st.h.clone()))
    return x


def sonar_dpmpp_sde(model, x, sigmas, cfg: MomentumCfg, *, eta=1.0, s_noise=1.0, noise_fn=None, rand_init=None,
                    trace: Optional[list] = None):
    """py/sonar.py:649-770 (DPM-Solver++ SDE, r = 1/2)."""
    st = MomentumState(cfg, rand_init)

    def sig(t):
        return t.neg().exp()

    def tf(s):
        return s.log().neg()

    s_in = x.new_ones((x.shape[0],))
    for i in range(len(sigmas) - 1):
        sigma, sigma_next = sigmas[i], sigmas[i + 1]
        den = model(x, sigma * s_in)
        if sigma_next == 0:
            down, _ = ancestral_step(sigma, sigma_next, eta)
            x = st.euler_step(i, x, den, sigma, down)
        else:
            adj = cfg.momentum + (1 - cfg.momentum) / 2 if st.h is not None else cfg.momentum
            r = 1 / 2
            t, t_next = tf(sigma), tf(sigma_next)
            h = t_next - t
            s = t + h * r
            fac = 1 / (2 * r)
            s_t, s_s = sig(t), sig(s)
            sd, su = ancestral_step(s_t, s_s, eta)
            s_ = tf(sd)
            md1 = st.momentum_denoised(x, den, sigma, i)
            diff_2 = (t - s_).expm1() * md1
            m_d = st.momentum_d(x, md1, sigma, i, gate_momentum=adj, d=diff_2)
            x_2 = ((sig(s_) / s_t) * x).sub_(m_d)
            x_2 += noise_fn(s_t, s_s).mul_(s_noise * su)
            den2 = model(x_2, s_s * s_in)
            md2 = st.momentum_denoised(x, den2, s_s, i)
            s_t_next = sig(t_next)
            sd, su = ancestral_step(s_t, s_t_next, eta)
            t_down = tf(sd)
            dd = (1 - fac) * md1 + fac * md2
            diff_1 = (t - t_down).expm1() * dd
            m_d = st.momentum_d(x, md2, s_s, i, gate_momentum=adj, d=diff_1)
            x = ((sig(t_down) / s_t) * x).sub_(m_d)
            x += noise_fn(s_t, s_t_next).mul_(s_noise * su)
        if trace is not None:
            trace.append((x.clone(), None if st.h is None else st.h.clone()))
    return x


# ------------------------------------------------------------------------------------------------ spatial power law, latent ops
def powerlaw_noise(draw: Tensor, alpha: float = 2.0, use_sign: bool = False, div_max_dims=None, use_div_max_abs: bool = True) -> Tensor:
    """py/noise_generation.py:775-786 (white / grey / velvet / violet and AdvancedPowerLawNoise)."""
    noise = draw.clone()
    modulation = torch.abs(noise) ** alpha
    noise = (torch.sign(noise) if use_sign else noise).mul_(modulation)
    if div_max_dims is not None:
        noise /= torch.amax(torch.abs(noise) if use_div_max_abs else noise, keepdim=True, dim=div_max_dims)
    return noise


def latent_op_advanced(t: Tensor, ops: Sequence[Callable], *, blend_mode: str, blend_strength: float, input_multiplier: float = 1.0,
                       output_multiplier: float = 1.0, difference_multiplier: float = 1.0) -> Tensor:
    """py/latent_ops.py:84-106 with every op enabled (the ==1.0 test on output_multiplier is the reference's)."""
    output = t * input_multiplier if input_multiplier != 1.0 else t
    for op in ops:
        output = op(output)
    diff = (output * output_multiplier if output_multiplier == 1.0 else output) - t
    if difference_multiplier != 1.0:
        diff *= difference_multiplier
    return blend(blend_mode, t, diff, blend_strength)


def latent_op_noise(t: Tensor, noise: Tensor, sigma: Optional[Tensor], scale_to_sigma: bool) -> Tensor:
    """py/latent_ops.py:180-186."""
    noise = noise.clone()
    if scale_to_sigma and sigma is not None:
        noise *= sigma
    noise += t
    return noise


# ------------------------------------------------------------------------------------------------ spectral-gain generators (F1, F2)
def onef_noise(draw: Tensor, alpha: float = 2.0, k: float = 1.0, hfac: float = 1.0, wfac: float = 1.0, base_power: float = 1.0,
               use_sqrt: bool = True) -> Tensor:
    """py/noise_generation.py:735-759 (4-D input): fftn over ALL dims / sqrt(power[h, w]) -> ifftn -> real."""
    batch, _c, height, width = draw.shape
    fx, fy = torch.meshgrid(torch.fft.fftfreq(height, hfac), torch.fft.fftfreq(width, wfac), indexing="ij")
    power = (fx**2 + fy**2) ** (-alpha / 2.0)
    if k != 0:
        power = k / power
    power[0, 0] = base_power
    power = power.unsqueeze(0).expand(batch, 1, height, width)
    noise_fft = torch.fft.fftn(draw)
    noise_fft /= torch.sqrt(power.to(noise_fft.dtype)) if use_sqrt else power.to(noise_fft.dtype)
    return torch.fft.ifftn(noise_fft).real


def green_test_noise(draw: Tensor, scale_fac: float = 1.0, x_pow=2, y_pow=2, power_base=1) -> Tensor:
    """py/noise_generation.py:693-704."""
    height, width = draw.shape[-2:]
    scale = scale_fac / (width * height)
    fy = torch.fft.fftfreq(height)[:, None] ** y_pow
    fx = torch.fft.fftfreq(width) ** x_pow
    power = torch.sqrt(fy + fx)
    power[0, 0] = power_base
    noise = torch.fft.ifft2(torch.fft.fft2(draw) / torch.sqrt(power))
    noise *= scale / noise.std()
    return torch.real(noise)


# ---- ModulatedNoise (py/noise.py:762-1019) ------------------------------------------------------------------
MODULATION_DIMS = (-3, (-2, -1), (-3, -2, -1))  # py/noise.py:763


def _modulation_gain(ref: Tensor, strength: float, dims) -> Tensor:
    """1 / (std * |strength| + 1) with std over `dims` of the (globally centred) reference, py/noise.py:795-803 and :821-826."""
    sd = torch.std(ref - ref.mean(), dim=dims, keepdim=True)
    return 1.0 / (sd * abs(strength) + 1.0)


def modulated_intensity(ref: Tensor, noise: Tensor, sigma_up, strength: float, dims, s_noise: float = 1.0) -> Tensor:
    """py/noise.py:784-810: noise scaled down where the reference is busy, renormalised to the plain noise's L2 norm, then mixed."""
    plain = noise * s_noise * sigma_up
    shaped = plain * _modulation_gain(ref, strength, dims) + plain
    shaped = shaped * (torch.norm(plain) / torch.norm(shaped))
    return shaped * strength + plain * (1 - strength)


def modulated_frequency(ref: Tensor, noise: Tensor, sigma_up, strength: float, dims, s_noise: float = 1.0) -> Tensor:
    """py/noise.py:812-866: the shaped noise goes through fft2, magnitudes are multiplied by 1 + (1 - exp(-(ky^2/h^2 + kx^2/w^2) b^2))
    (b = |strength|; unshifted index grid, as the reference writes it), ifft2, real part, same renormalise + mix."""
    plain = noise * s_noise * sigma_up
    spec = torch.fft.fft2(plain * _modulation_gain(ref, strength, dims) + plain)
    h, w = ref.shape[-2:]
    boost = 2.0 - torch.exp(-((torch.arange(h)[:, None] / h) ** 2 + (torch.arange(w)[None, :] / w) ** 2) * strength**2)
    shaped = torch.fft.ifft2(spec * boost).real
    shaped = shaped * (torch.norm(plain) / torch.norm(shaped))
    return shaped * strength + plain * (1 - strength)


def modulated_noise(x_or_ref: Tensor, noise: Tensor, s, sn, *, modulation_type: str, strength: float, modulation_dims: int, factor: float,
                    normalize_ref: bool, normalize_result: bool) -> Tensor:
    """ModulatedNoise.make_noise_sampler's closure (py/noise.py:1002-1017): sigma_up from get_ancestral_step(s, sn, eta=1); the
    reference tensor is passed through scale_noise first (IN PLACE -- it is the sampler's own x when no ref latent is given)."""
    _, sigma_up = ancestral_step(s, sn, 1.0)
    fn = {"intensity": modulated_intensity, "frequency": modulated_frequency}[modulation_type]
    ref = scale_noise(x_or_ref, normalized=normalize_ref)
    return scale_noise(fn(ref, noise, sigma_up, strength, MODULATION_DIMS[modulation_dims - 1]), factor, normalized=normalize_result)


# ---- two more registry types (py/noise_generation.py:789-802, 1259-1287) ------------------------------------------------------
def laplacian_noise(normal_draw: Tensor, uniform_draw: Tensor, loc: float = 0.0, scale: float = 1.0, div_fac: float = 4.0) -> Tensor:
    """LaplacianNoiseGenerator.generate: randn / div_fac + Laplace(loc, scale).rsample(); `uniform_draw` is the rsample's
    uniform_(eps - 1, 1) (torch.distributions.Laplace: loc - scale * sign(u) * log1p(-|u|.clamp(min=tiny)))."""
    tiny = torch.finfo(uniform_draw.dtype).tiny
    lap = loc - scale * uniform_draw.sign() * torch.log1p(-uniform_draw.abs().clamp(min=tiny))
    return normal_draw / div_fac + lap


def power_old_noise(uniform_draw: Tensor, alpha: float = 2, k: float = 1) -> Tensor:
    """PowerOldNoiseGenerator.generate: rand * k / (batch index + 1)^alpha, then every [H, W] plane standardised."""
    b = uniform_draw.shape[0]
    freq = torch.arange(1, b + 1, dtype=uniform_draw.dtype).reshape((b,) + (1,) * (uniform_draw.dim() - 1))
    noise = uniform_draw * (k / freq**alpha)
    return (noise - noise.mean(dim=(-2, -1), keepdim=True)) / noise.std(dim=(-2, -1), keepdim=True)


def studentt_noise(normal_draw: Tensor, gamma_draw: Tensor, loc: float = 0.0, scale: float = 0.2, df: float = 1.0, quantile_fac: float = 0.75,
                   pow_fac: float = 0.5, nq_fac: float = 1.0) -> Tensor:
    """StudentTNoiseGenerator.generate (py/noise_generation.py:652-677).  torch.distributions.StudentT.rsample draws X = empty.normal_()
    and the Chi2 through torch._standard_gamma(df / 2) (`gamma_draw`), Z = (gamma / 0.5).clamp(min=tiny), Y = X * rsqrt(Z / df); then the
    per-latent quantile of |noise| clamps the tails and sign(x) |x|^pow_fac compresses them."""
    z = (gamma_draw / 0.5).clamp(min=torch.finfo(gamma_draw.dtype).tiny)
    noise = loc + scale * (normal_draw * torch.rsqrt(z / df))
    nq = torch.quantile(noise.flatten(start_dim=1).abs(), quantile_fac, dim=-1) * nq_fac
    nq = nq.reshape(tuple(nq.shape) + (1,) * (noise.ndim - nq.ndim))
    noise = noise.clamp(-nq, nq)
    return torch.copysign(torch.pow(torch.abs(noise), pow_fac), noise)
