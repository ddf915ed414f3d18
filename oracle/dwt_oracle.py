"""CPU oracle for the 2-D DWT / IDWT — TEST INFRASTRUCTURE ONLY (numpy, fp64 or fp32).

The reference never implements wavelet arithmetic itself: ``py/wavelet_functions.py:56-105`` wraps the
third-party ``pytorch_wavelets`` (``DWTForward(J, wave, mode)`` / ``DWTInverse(wave, mode)``), which takes its
filter taps from ``pywt``.  Neither package is vendored, pinned (no requirements file) or installed here;
the reference has no tests at that boundary.  This file therefore restates the *published* algorithm —
PyWavelets' ``dwt``/``idwt`` (downsampling / upsampling convolution with signal extension) applied
separably, in pytorch_wavelets' output layout — and is pinned against golden vectors produced by
PyWavelets 1.1.1 (``tests/golden/make_dwt_golden.py`` -> ``tests/golden/dwt.npz``).
Parity with the reference for rows W / WC / WF is anchored on those vectors and on the reference's call sites
(py/wavelet_functions.py:81-105,193-238; py/wavelet_cfg.py:750-791), not on reference-run outputs.

Layout (pytorch_wavelets): ``yl [B,C,h_J,w_J]``, ``yh[j] [B,C,3,h_j,w_j]`` with j = 0 the finest level and the
orientation axis ordered (LH, HL, HH) == pywt (cH, cV, cD): cH = high-pass along H / low-pass along W.
"""
from __future__ import annotations

import json
import os

import numpy as np

MODES = ("zero", "symmetric", "reflect", "periodization", "periodic", "constant")
_TAPS = None


def taps(name: str):
    """(dec_lo, dec_hi, rec_lo, rec_hi) from the PyWavelets-generated table shipped with the package data."""
    global _TAPS
    if _TAPS is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "comfyui-sonar_amd", "wavelet_taps.json")
        _TAPS = json.load(open(path))["wavelets"]
    w = _TAPS[name]
    return tuple(np.asarray(w[k], dtype=np.float64) for k in ("dec_lo", "dec_hi", "rec_lo", "rec_hi"))


def out_len(n: int, flen: int, mode: str) -> int:
    """pywt.dwt_coeff_len."""
    return (n + 1) // 2 if mode == "periodization" else (n + flen - 1) // 2


def _ext_index(idx: np.ndarray, n: int, mode: str):
    """Map extended-signal indices to (source index, validity mask) for the non-periodization modes."""
    valid = np.ones(idx.shape, dtype=bool)
    if mode == "zero":
        valid = (idx >= 0) & (idx < n)
        src = np.clip(idx, 0, n - 1)
    elif mode == "constant":
        src = np.clip(idx, 0, n - 1)
    elif mode == "periodic":
        src = np.mod(idx, n)
    elif mode == "symmetric":  # ... x1 x0 | x0 x1 ... x_{n-1} | x_{n-1} ...
        p = np.mod(idx, 2 * n)
        src = np.where(p < n, p, 2 * n - 1 - p)
    elif mode == "reflect":    # ... x2 x1 | x0 x1 ... x_{n-1} | x_{n-2} ...
        if n == 1:
            src = np.zeros_like(idx)
        else:
            p = np.mod(idx, 2 * n - 2)
            src = np.where(p < n, p, 2 * n - 2 - p)
    else:
        raise ValueError(mode)
    return src, valid


def dwt_axis(x: np.ndarray, lo: np.ndarray, hi: np.ndarray, mode: str, axis: int):
    """One analysis step along ``axis``: out[i] = sum_j f[j] * ext(x)[2i + 1 - j]  (pywt downsampling_convolution);
    periodization: out[i] = sum_j f[j] * xe[(2i - j + F/2) mod Ne], xe = x with its last sample repeated if n is odd."""
    x = np.moveaxis(x, axis, -1)
    n, flen = x.shape[-1], len(lo)
    m = out_len(n, flen, mode)
    i = np.arange(m)[:, None]
    j = np.arange(flen)[None, :]
    if mode == "periodization":
        if n % 2:
            x = np.concatenate([x, x[..., -1:]], axis=-1)
        ne = x.shape[-1]
        src = np.mod(2 * i - j + flen // 2, ne)
        g = x[..., src]  # [..., m, F]
    else:
        src, valid = _ext_index(2 * i + 1 - j, n, mode)
        g = x[..., src] * valid
    a = (g * lo.astype(x.dtype)).sum(-1)
    d = (g * hi.astype(x.dtype)).sum(-1)
    return np.moveaxis(a, -1, axis), np.moveaxis(d, -1, axis)


def idwt_axis(a: np.ndarray, d: np.ndarray, lo: np.ndarray, hi: np.ndarray, mode: str, axis: int):
    """One synthesis step: x[o] = sum_{2i + j = o + F - 2} a[i] lo[j] + d[i] hi[j]  (length 2n - F + 2);
    periodization: indices modulo 2n with 2i + j == o + F/2 - 1."""
    a = np.moveaxis(a, axis, -1)
    d = np.moveaxis(d, axis, -1)
    n, flen = a.shape[-1], len(lo)
    if mode == "periodization":
        nout = 2 * n
        o = np.arange(nout)[:, None]
        i = np.arange(n)[None, :]
        j = np.mod(o + flen // 2 - 1 - 2 * i, nout)          # [nout, n]
        jj = j[..., None] + nout * np.arange((flen + nout - 1) // nout + 1)  # taps wrap more than once when F > 2n
        ok = jj < flen
        wl = np.where(ok, lo[np.minimum(jj, flen - 1)], 0.0).sum(-1)
        wh = np.where(ok, hi[np.minimum(jj, flen - 1)], 0.0).sum(-1)
    else:
        nout = 2 * n - flen + 2
        o = np.arange(nout)[:, None]
        i = np.arange(n)[None, :]
        j = o + flen - 2 - 2 * i
        ok = (j >= 0) & (j < flen)
        wl = np.where(ok, lo[np.clip(j, 0, flen - 1)], 0.0)
        wh = np.where(ok, hi[np.clip(j, 0, flen - 1)], 0.0)
    x = a @ wl.T.astype(a.dtype) + d @ wh.T.astype(a.dtype)
    return np.moveaxis(x, -1, axis)


def dwt2(x: np.ndarray, wave: str, mode: str):
    """One 2-D level on [..., H, W] -> (ll, hi[..., 3, h, w]) in (cH, cV, cD) order."""
    dec_lo, dec_hi, _, _ = taps(wave)
    lo_w, hi_w = dwt_axis(x, dec_lo, dec_hi, mode, -1)
    ll, ch = dwt_axis(lo_w, dec_lo, dec_hi, mode, -2)   # cH: high along H, low along W
    cv, cd = dwt_axis(hi_w, dec_lo, dec_hi, mode, -2)   # cV: low along H, high along W
    return ll, np.stack([ch, cv, cd], axis=-3)


def idwt2(ll: np.ndarray, hi: np.ndarray, wave: str, mode: str):
    _, _, rec_lo, rec_hi = taps(wave)
    ch, cv, cd = hi[..., 0, :, :], hi[..., 1, :, :], hi[..., 2, :, :]
    lo_w = idwt_axis(ll, ch, rec_lo, rec_hi, mode, -2)
    hi_w = idwt_axis(cv, cd, rec_lo, rec_hi, mode, -2)
    return idwt_axis(lo_w, hi_w, rec_lo, rec_hi, mode, -1)


def wavedec2(x: np.ndarray, wave: str, mode: str, level: int):
    """pytorch_wavelets DWTForward(J=level): (yl, [yh_finest, ..., yh_coarsest])."""
    yh = []
    ll = x
    for _ in range(level):
        ll, hi = dwt2(ll, wave, mode)
        yh.append(hi)
    return ll, yh


def waverec2(yl: np.ndarray, yh, wave: str, mode: str):
    """pytorch_wavelets DWTInverse: coarse to fine; drop the extra row/col when ll is one larger than the band."""
    ll = yl
    for hi in reversed(yh):
        if ll.shape[-2] > hi.shape[-2]:
            ll = ll[..., :-1, :]
        if ll.shape[-1] > hi.shape[-1]:
            ll = ll[..., :-1]
        ll = idwt2(ll, hi, wave, mode)
    return ll


def wavedec1(x: np.ndarray, wave: str, mode: str, level: int):
    """pytorch_wavelets DWT1DForward(J=level) on [..., L] (Wavelet(use_1d_dwt=True), py/wavelet_functions.py:56-57):
    (yl, [yh_finest, ..., yh_coarsest]); == pywt.wavedec reordered."""
    dec_lo, dec_hi, _, _ = taps(wave)
    yh = []
    lo = x
    for _ in range(level):
        lo, hi = dwt_axis(lo, dec_lo, dec_hi, mode, -1)
        yh.append(hi)
    return lo, yh


def waverec1(yl: np.ndarray, yh, wave: str, mode: str):
    """pytorch_wavelets DWT1DInverse: coarse to fine; drop the extra sample when the approximation is one longer than the band."""
    _, _, rec_lo, rec_hi = taps(wave)
    lo = yl
    for hi in reversed(yh):
        if lo.shape[-1] > hi.shape[-1]:
            lo = lo[..., :-1]
        lo = idwt_axis(lo, hi, rec_lo, rec_hi, mode, -1)
    return lo


# ------------------------------------------------------------------------------------------------ WaveletCFG arithmetic
def expand_yh_scales(nbands: int, norient: int, yh_scales):
    """py/wavelet_functions.py:148-190 ("fill" repeats the previous entry up to the band count)."""
    if isinstance(yh_scales, (float, int)):
        return ((float(yh_scales),) * norient,) * nbands
    template = (1.0,) * norient
    out = []
    for band in yh_scales:
        if isinstance(band, (float, int)):
            out.append((float(band),) * norient)
        elif isinstance(band, (tuple, list)):
            vals = tuple(float(v) for v in band[:norient])
            out.append(vals + template[: norient - len(vals)])
        else:
            out.append(band)
    out = tuple(out)
    if "fill" in out:
        k = out.index("fill")
        if "fill" in out[k + 1:]:
            raise ValueError("Only one fill allowed.")
        if k == 0 or len(out) < 2:
            raise ValueError("Invalid fill value, cannot be in the first position or the only item.")
        if len(out) - 1 < nbands:
            out = (*out[:k], *((out[k - 1],) * (nbands - (len(out) - 1))), *out[k + 1:])
        else:
            out = (*out[:k], *out[k + 1:])
    return out[:nbands]


def wavelet_scaling(yl, yh, yl_scale, yh_scales):
    """py/wavelet_functions.py:193-216 (out of place).  The reference indexes the scaled slice as ``ht[:, :, lidx]`` for
    ``lidx < min(ht.shape[2], len(scales))``: axis 2 is the orientation axis of a 2-D band ``[B, C, 3, h, w]``, but for the 1-D
    transform's bands ``[B, C, l]`` it is the COEFFICIENT axis and the table has one entry per band -- only coefficient 0 of every
    row is scaled there (pinned by tests/golden/wavelet_scaling.npz, cases ``oned_*``)."""
    yl = yl * yl_scale if yl_scale != 1.0 else yl.copy()
    norient = yh[0].shape[2] if yh[0].ndim > 3 else 1
    scales = expand_yh_scales(len(yh), norient, 1.0 if yh_scales is None else yh_scales)
    out = [band.copy() for band in yh]
    for sc, band in zip(scales, out):
        for o in range(min(band.shape[2], len(sc))):
            band[:, :, o] *= sc[o]
    return yl, out


def _lerp(a, b, t):
    """torch.lerp: a + t (b - a) for |t| < 0.5, else b - (b - a)(1 - t)."""
    return a + t * (b - a) if abs(t) < 0.5 else b - (b - a) * (1 - t)


BLENDS = {"lerp": _lerp, "inject": lambda a, b, t: b * t + a, "subtract_b": lambda a, b, t: a - b * t}


def wavelet_blend(a, b, *, yl_factor, blend, yh_factor=None, yh_blend=None):
    """py/wavelet_functions.py:219-238 with named blends (py/utils.py:17-21)."""
    yh_factor = yl_factor if yh_factor is None else yh_factor
    fl, fh = BLENDS[blend], BLENDS[blend if yh_blend is None else yh_blend]
    return fl(a[0], b[0], yl_factor), [fh(x, y, yh_factor) for x, y in zip(a[1], b[1])]


def _transform(one_d):
    return (wavedec1, waverec1) if one_d else (wavedec2, waverec2)


def wavelet_cfg(cond, uncond, wave, mode, level, *, cond_scales=None, uncond_scales=None, diff_scales=None, final_scales=None, strength=1.0,
                blend="inject", inv_wave=None, inv_mode=None, one_d=False):
    """py/wavelet_cfg.py:750-791: ``IDWT(scale_f(blend(U, scale_d(C - U), s)))`` with ``C = scale_c(DWT cond)``, ``U = scale_u(DWT uncond)``.
    ``*_scales`` are (yl_scale, yh_scales) pairs or None (rule section absent); returns the reconstruction at its own size."""
    dec, rec = _transform(one_d)
    cw, uw = dec(cond, wave, mode, level), dec(uncond, wave, mode, level)
    if cond_scales is not None:
        cw = wavelet_scaling(*cw, *cond_scales)
    if uncond_scales is not None:
        uw = wavelet_scaling(*uw, *uncond_scales)
    dw = (cw[0] - uw[0], [a - b for a, b in zip(cw[1], uw[1])])
    if diff_scales is not None:
        dw = wavelet_scaling(*dw, *diff_scales)
    rw = wavelet_blend(uw, dw, yl_factor=strength, blend=blend)
    if final_scales is not None:
        rw = wavelet_scaling(*rw, *final_scales)
    return rec(rw[0], rw[1], inv_wave or wave, inv_mode or mode)


def wavelet_cfg_call(args, *, target="denoised", high_precision=True, use_1d=False, wcfg_blend=1.0, blend_mode="lerp", fallback=None, ops=None,
                     **transform_kw):
    """py/wavelet_cfg.py:677-748,793-842 for a rule that matched: context (target selection, NOISE_NORM division, operation hooks,
    flattening), the transform-domain step in fp64 / fp32, the optional blend with the fallback CFG result, crop, ``x - result`` /
    ``* sigma``.  ``args``: numpy fp32 arrays input / cond / uncond / cond_denoised / uncond_denoised, sigma [B], cond_scale;
    ``fallback(args)`` / ``ops[...]`` are numpy callables (None: plain CFG / no hook)."""
    ops = ops or {}
    x = args["input"]
    sigma = args["sigma"].reshape(x.shape[0], *((1,) * (x.ndim - 1))).astype(np.float32)
    if x.ndim == 3 and not use_1d:
        raise RuntimeError("Enable use_1d_dwt mode for 3D latents.")
    if x.ndim < 3:
        raise RuntimeError("Wavelet CFG can't handle latents with 2 or less dimensions.")
    if target == "denoised":
        cond, uncond = args["cond_denoised"], args["uncond_denoised"]
    else:
        cond, uncond = args["cond"], args["uncond"]
        if target == "noise_norm":
            cond, uncond = cond / sigma, uncond / sigma
    op_kw = dict(sigma=args["sigma"], cond=cond, uncond=uncond, cond_scale=args["cond_scale"])

    def hook(name, t):
        return ops[name](t, **op_kw) if ops.get(name) is not None else t

    def plain_cfg(a):
        return a["input"] - ((a["cond_denoised"] - a["uncond_denoised"]) * np.float32(a["cond_scale"]) + a["uncond_denoised"])

    fallback = plain_cfg if fallback is None else fallback
    cond, uncond = hook("operation_cond", cond), hook("operation_uncond", uncond)
    if use_1d:
        cond, uncond = cond.reshape(*cond.shape[:2], -1), uncond.reshape(*uncond.shape[:2], -1)
    elif x.ndim > 4:
        cond, uncond = cond.reshape(cond.shape[0], -1, *cond.shape[-2:]), uncond.reshape(uncond.shape[0], -1, *uncond.shape[-2:])
    dt = np.float64 if high_precision else np.float32
    result = wavelet_cfg(cond.astype(dt), uncond.astype(dt), one_d=use_1d, **transform_kw).astype(np.float32)
    if blend_mode != "lerp" or wcfg_blend != 1.0:
        normal = hook("operation_fallback_cfg", fallback(args))
        if target == "denoised":
            normal = x - normal
        elif target == "noise_norm":
            normal = normal / sigma
        # blended before the crop (:825-836): torch refuses shapes that do not broadcast, first mismatch counted from the trailing dimension
        n = max(normal.ndim, result.ndim)
        pa, pb = (1,) * (n - normal.ndim) + normal.shape, (1,) * (n - result.ndim) + result.shape
        for d in range(n - 1, -1, -1):
            if pa[d] != pb[d] and pa[d] != 1 and pb[d] != 1:
                raise RuntimeError(f"The size of tensor a ({pa[d]}) must match the size of tensor b ({pb[d]}) at non-singleton dimension {d}")
        result = BLENDS[blend_mode](normal, result, np.float32(wcfg_blend)).astype(np.float32)
    if use_1d:
        result = result[..., : cond.shape[2]].reshape(x.shape)
    elif x.ndim > 4:
        result = result[..., : x.shape[-2], : x.shape[-1]].reshape(x.shape)
    else:
        result = result[tuple(slice(None, n) for n in x.shape)]
    if target == "denoised":
        result = x - result
    elif target == "noise_norm":
        result = result * sigma
    return hook("operation_result", hook("operation_wavelet_cfg", result))


def wavelet_filtered_noise(noise, *, wave="haar", mode="periodization", level=3, yl_scale=1.0, yh_scales=1.0, noise_high=None,
                           yl_blend_high=0.0, yh_blend_high=1.0, yl_blend="lerp", yh_blend="lerp", two_step_inverse=False,
                           preblend_low=None, preblend_high=None, inv_wave=None, inv_mode=None, use_1d_dwt=False):
    """py/noise_generation.py:1968-2032 (WaveletFilteredNoiseGenerator.generate).  ``noise`` / ``noise_high``: the base draws,
    [B, C, H, W] or [B, C, T, H, W] (frames fold into channels, :186-200); preblend_* = (yl_scale, yh_scales) pairs applied to the
    low / high decompositions before blending (either member None -> 1.0, whole pair None -> step skipped)."""
    shape = noise.shape
    fold = (lambda t: t.reshape(t.shape[0], -1, *t.shape[-2:])) if noise.ndim == 5 else (lambda t: t)
    noise = fold(noise)
    work_shape = noise.shape
    flat = (lambda t: t.reshape(*t.shape[:2], -1)) if use_1d_dwt else (lambda t: t)
    dec, rec = _transform(use_1d_dwt)
    inv_wave, inv_mode = inv_wave or wave, inv_mode or mode
    yl, yh = dec(flat(noise), wave, mode, level)
    if noise_high is not None:
        hl, hh = dec(flat(fold(noise_high)), wave, mode, level)
        if preblend_high is not None:
            hl, hh = wavelet_scaling(hl, hh, *(1.0 if v is None else v for v in preblend_high))
        if preblend_low is not None:
            yl, yh = wavelet_scaling(yl, yh, *(1.0 if v is None else v for v in preblend_low))
        yl, yh = wavelet_blend((yl, yh), (hl, hh), yl_factor=yl_blend_high, yh_factor=yh_blend_high, blend=yl_blend, yh_blend=yh_blend)
    yl, yh = wavelet_scaling(yl, yh, yl_scale, yh_scales)
    if two_step_inverse:
        out = rec(np.zeros_like(yl), yh, inv_wave, inv_mode) + rec(yl, [np.zeros_like(b) for b in yh], inv_wave, inv_mode)
    else:
        out = rec(yl, yh, inv_wave, inv_mode)
    if use_1d_dwt:
        out = out.reshape(work_shape)
    out = out[tuple(slice(0, d) for d in work_shape)]
    return out.reshape(shape)
