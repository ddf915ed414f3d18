"""Dual-tree complex wavelet transform on MI355X: what the reference gets from ``pytorch_wavelets.DTCWTForward / DTCWTInverse``
(py/wavelet_functions.py:56-73), i.e. N. G. Kingsbury's dtwavexfm2 / dtwaveifm2.

pytorch_wavelets and its filter-bank data files are not part of the reference (un-vendored, absent): **parity unpinned**.  The
algorithm and the filter banks are the published ones -- `near_sym_a` (5 / 7 taps), `legall` (5 / 3) and `antonini` (9 / 7) in closed form, `qshift_a`
(10 taps) from the published coefficients -- and are tested by their defining properties: perfect reconstruction, orthonormal shifts /
half-band products of the filters, orientation selectivity of the six subbands, agreement with a plain numpy restatement kept with
the tests.  Layout as pytorch_wavelets documents it: ``yl [B, C, H / 2**(J-1), W / 2**(J-1)]`` and
``yh[j] [B, C, 6, h_j, w_j, 2]`` (orientations 15, 45, 75, 105, 135, 165 degrees; last axis real / imaginary), finest level first.

Every stage is a sparse linear map along one axis; the host builds its (source index, coefficient) table for a given length once (numpy,
below) and ``sonar_axis_taps_*`` applies it on the device.  The quad <-> complex-pair shuffles are ``sonar_dtcwt_q2c_* / c2q_*``.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from .. import hip_lib

_BIORT = {
    # analysis / synthesis low-pass (odd lengths, unit DC gain); the high-pass pair is their alternating-sign mirror
    "near_sym_a": ([-1 / 20, 5 / 20, 12 / 20, 5 / 20, -1 / 20], [-3 / 280, -15 / 280, 73 / 280, 170 / 280, 73 / 280, -15 / 280, -3 / 280]),
    "legall": ([-1 / 8, 2 / 8, 6 / 8, 2 / 8, -1 / 8], [1 / 4, 2 / 4, 1 / 4]),
    # CDF 9 / 7: the factors of the eighth-order maximally flat half-band filter (four zeros at -1 each; h0o takes the complex root pair
    # of 1 + 4y + 10y^2 + 20y^3, g0o the real root), unit DC gain; agrees with the published 12-digit table
    "antonini": ([0.02674875741081003, -0.01686411844287496, -0.07822326652899003, 0.2668641184428749, 0.60294901823636,
                  0.2668641184428749, -0.07822326652899003, -0.01686411844287496, 0.02674875741081003],
                 [-0.04563588155712507, -0.02877176311425014, 0.29563588155712506, 0.5575435262285002, 0.29563588155712506,
                  -0.02877176311425014, -0.04563588155712507]),
}
_QSHIFT = {
    "qshift_a": [0.0511304052838317, -0.0139753702468888, -0.109836051665971, 0.263839561058938, 0.766628467793037, 0.563655710127052,
                 0.000873622695217097, -0.100231219507476, -0.00168968127252815, -0.00618188189211644],
}


def biort_filters(name: str):
    """(h0o, g0o, h1o, g1o): h1o[n] = (-1)^n g0o[n], g1o[n] = -(-1)^n h0o[n], so that h0o * g0o + h1o * g1o = delta."""
    if name not in _BIORT:
        raise NotImplementedError(f"DTCWT level-1 filter bank {name!r}: its coefficients are data files of the absent pytorch_wavelets / dtcwt "
                                  f"packages; built in: {', '.join(_BIORT)}")
    h0o, g0o = (np.asarray(v, dtype=np.float64) for v in _BIORT[name])
    sign = lambda v: np.array([(-1.0) ** i for i in range(len(v))])  # noqa: E731
    return h0o, g0o, g0o * sign(g0o), -h0o * sign(h0o)


def qshift_filters(name: str):
    """(h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b): tree b is tree a reversed; each tree is synthesised with the other's analysis filter."""
    if name not in _QSHIFT:
        raise NotImplementedError(f"DTCWT q-shift filter bank {name!r}: its coefficients are data files of the absent pytorch_wavelets / dtcwt "
                                  f"packages; built in: {', '.join(_QSHIFT)}")
    h0a = np.asarray(_QSHIFT[name], dtype=np.float64)
    h0b = h0a[::-1].copy()
    h1a = h0b * np.array([(-1.0) ** i for i in range(len(h0a))])
    h1b = h1a[::-1].copy()
    return h0a, h0b, h0b, h0a, h1a, h1b, h1b, h1a


def _fold(i: np.ndarray, n: int) -> np.ndarray:
    """Half-sample symmetric extension of 0 .. n-1 (the toolbox's reflect(x, -0.5, n - 0.5))."""
    t = np.mod(i, 2 * n)
    return np.where(t >= n, 2 * n - 1 - t, t)


# ---- the three kinds of stage as (idx [n_out, taps], coef [n_out, taps]) tables ---------------------------------------------------------
def table_odd(n: int, h: np.ndarray):
    """colfilter: y[j] = sum_k h[k] x_ext[j + m - k], m = (len(h) - 1) / 2, same length out."""
    L = len(h)
    m = L // 2
    j = np.arange(n)[:, None]
    k = np.arange(L)[None, :]
    return _fold(j + m - k, n).astype(np.int32), np.broadcast_to(h[None, :], (n, L)).copy()


def table_decimate(n: int, ha: np.ndarray, hb: np.ndarray):
    """coldfilt: n -> n / 2.  Output pair q = 0 .. n/4 - 1: the `a` row takes the odd taps of ha on the samples 4q + 2s + 4 - m (extended
    index; s = tap pair, latest first) and the even taps two samples earlier; the `b` row the same one sample later.  The `a` rows are the
    even outputs when sum(ha hb) > 0 (the low-pass pair), the odd ones otherwise."""
    if n % 4:
        raise ValueError("DTCWT: a decimated length must be a multiple of 4")
    m = len(ha)
    half = m // 2
    q = np.arange(n // 4)[:, None]
    s = np.arange(half)[None, :]
    # toolbox (1-based): t = 6:4:(r + 2m - 2); Y(s1) = conv(X(xe(t - 1)), hao, 'valid') + conv(X(xe(t - 3)), hae, 'valid'); xe(i) <-> i - m - 1
    # conv 'valid': out[q] = sum_s f[s] u[q + half - 1 - s]; u[p] = X(xe(t_p + off)), t_p = 5 + 4 p (0-based position) -> extended index t_p + off - m
    pos = 5 + 4 * (q + half - 1 - s) - m
    ia_o, ia_e = _fold(pos - 1, n), _fold(pos - 3, n)
    ib_o, ib_e = _fold(pos, n), _fold(pos - 2, n)
    rows_a = (np.concatenate((ia_o, ia_e), axis=1), np.concatenate((np.broadcast_to(ha[0::2], ia_o.shape), np.broadcast_to(ha[1::2], ia_e.shape)), axis=1))
    rows_b = (np.concatenate((ib_o, ib_e), axis=1), np.concatenate((np.broadcast_to(hb[0::2], ib_o.shape), np.broadcast_to(hb[1::2], ib_e.shape)), axis=1))
    idx = np.empty((n // 2, m), dtype=np.int32)
    coef = np.empty((n // 2, m), dtype=np.float64)
    first, second = (rows_a, rows_b) if float(np.sum(ha * hb)) > 0 else (rows_b, rows_a)
    idx[0::2], coef[0::2] = first
    idx[1::2], coef[1::2] = second
    return idx, coef


def table_interpolate(n: int, ha: np.ndarray, hb: np.ndarray):
    """colifilt: n -> 2 n (toolbox colifilt.m, both parities of len(ha) / 2)."""
    if n % 2:
        raise ValueError("DTCWT: an interpolated length must be even")
    m = len(ha)
    m2 = m // 2
    hao, hae, hbo, hbe = ha[0::2], ha[1::2], hb[0::2], hb[1::2]
    positive = float(np.sum(ha * hb)) > 0
    idx = np.empty((2 * n, m2), dtype=np.int32)
    coef = np.empty((2 * n, m2), dtype=np.float64)
    q = np.arange(n // 2)[:, None]
    s = np.arange(m2)[None, :]

    def rows(t0, off, filt):
        # u[p] = X(xe(t_p + off)), t_p = t0 + 2 p (0-based position in xe), xe position i <-> extended index i - m2; valid conv with `filt`
        pos = t0 + 2 * (q + m2 - 1 - s) + off - m2
        return _fold(pos, n).astype(np.int32), np.broadcast_to(filt[None, :], pos.shape)

    if m2 % 2 == 0:
        t0 = 3  # toolbox t = 4:2:(r + m)
        ta, tb = (0, -1) if positive else (-1, 0)
        plan = ((tb - 2, hae), (ta - 2, hbe), (tb, hao), (ta, hbo))
    else:
        t0 = 2  # toolbox t = 3:2:(r + m - 1)
        ta, tb = (0, -1) if positive else (-1, 0)
        plan = ((tb, hao), (ta, hbo), (tb, hae), (ta, hbe))
    for r, (off, filt) in enumerate(plan):
        idx[r::4], coef[r::4] = rows(t0, off, filt)
    return idx, coef


_TABLES: dict = {}


def _device_table(kind: str, n: int, filters: Sequence[np.ndarray], dtype: torch.dtype, device) -> tuple:
    key = (kind, n, tuple(tuple(np.round(f, 15)) for f in filters), dtype, str(device))
    hit = _TABLES.get(key)
    if hit is None:
        if len(_TABLES) > 256:
            _TABLES.clear()
        build = {"odd": table_odd, "dec": table_decimate, "int": table_interpolate}[kind]
        idx, coef = build(n, *filters)
        hit = _TABLES[key] = (torch.from_numpy(np.ascontiguousarray(idx)).to(device), torch.from_numpy(np.ascontiguousarray(coef)).to(device=device, dtype=dtype))
    return hit


def _apply(kind: str, x: torch.Tensor, axis: int, filters, out=None, accumulate=False) -> torch.Tensor:
    idx, coef = _device_table(kind, x.shape[axis], filters, x.dtype, x.device)
    return hip_lib.axis_taps(x, idx, coef, axis, out=out, accumulate=accumulate)


class DTCWT:
    """J-level forward / inverse pair with pytorch_wavelets' defaults (``biort='near_sym_a'``, ``qshift='qshift_a'``, symmetric extension)."""

    def __init__(self, *, level: int, biort: str = "near_sym_a", qshift: str = "qshift_a", inv_biort=None, inv_qshift=None):
        self.level = int(level)
        self.h0o, _g, self.h1o, _g1 = biort_filters(biort)
        _h, self.g0o, _h1, self.g1o = biort_filters(inv_biort or biort)
        self.h0a, self.h0b, _, _, self.h1a, self.h1b, _, _ = qshift_filters(qshift)
        _, _, self.g0a, self.g0b, _, _, self.g1a, self.g1b = qshift_filters(inv_qshift or qshift)

    def forward(self, x: torch.Tensor):
        if x.ndim != 4 or x.dtype not in (torch.float32, torch.float64) or not x.is_cuda:
            raise hip_lib.SonarHipError("DTCWT.forward expects a float32 / float64 [B, C, H, W] tensor on a ROCm device")
        if self.level == 0:
            return x, ()
        if x.shape[-2] % 2:
            x = torch.cat((x, x[..., -1:, :]), dim=-2)
        if x.shape[-1] % 2:
            x = torch.cat((x, x[..., :, -1:]), dim=-1)
        x = x.contiguous()
        yh = []
        lo, hi = _apply("odd", x, -1, (self.h0o,)), _apply("odd", x, -1, (self.h1o,))
        ll = _apply("odd", lo, -2, (self.h0o,))
        yh.append(hip_lib.dtcwt_q2c(_apply("odd", lo, -2, (self.h1o,)), _apply("odd", hi, -2, (self.h1o,)), _apply("odd", hi, -2, (self.h0o,))))
        for _ in range(1, self.level):
            if ll.shape[-2] % 4:
                ll = torch.cat((ll[..., :1, :], ll, ll[..., -1:, :]), dim=-2)
            if ll.shape[-1] % 4:
                ll = torch.cat((ll[..., :, :1], ll, ll[..., :, -1:]), dim=-1)
            ll = ll.contiguous()
            low, high = (self.h0b, self.h0a), (self.h1b, self.h1a)
            lo, hi = _apply("dec", ll, -1, low), _apply("dec", ll, -1, high)
            ll = _apply("dec", lo, -2, low)
            yh.append(hip_lib.dtcwt_q2c(_apply("dec", lo, -2, high), _apply("dec", hi, -2, high), _apply("dec", hi, -2, low)))
        return ll, tuple(yh)

    def inverse(self, coeffs) -> torch.Tensor:
        yl, yh = coeffs
        ll = yl.contiguous()
        for j in range(len(yh) - 1, -1, -1):
            if yh[j] is None or yh[j].numel() == 0:
                raise hip_lib.SonarHipError("DTCWT.inverse: every level's bands are required")
            lh, hh, hl = hip_lib.dtcwt_c2q(yh[j])
            if tuple(ll.shape[-2:]) != tuple(lh.shape[-2:]):
                raise hip_lib.SonarHipError(f"DTCWT.inverse: level {j + 1} low-pass {tuple(ll.shape[-2:])} does not match its bands {tuple(lh.shape[-2:])}")
            if j > 0:
                low, high = (self.g0b, self.g0a), (self.g1b, self.g1a)
                kind = "int"
            else:
                low, high = (self.g0o,), (self.g1o,)
                kind = "odd"
            y1 = _apply(kind, ll, -2, low)
            _apply(kind, lh, -2, high, out=y1, accumulate=True)
            y2 = _apply(kind, hl, -2, low)
            _apply(kind, hh, -2, high, out=y2, accumulate=True)
            ll = _apply(kind, y1, -1, low)
            _apply(kind, y2, -1, high, out=ll, accumulate=True)
            if j > 0:
                want_h, want_w = 2 * yh[j - 1].shape[3], 2 * yh[j - 1].shape[4]
                if ll.shape[-2] != want_h:
                    ll = ll[..., 1:-1, :]
                if ll.shape[-1] != want_w:
                    ll = ll[..., :, 1:-1]
                ll = ll.contiguous()
        return ll
