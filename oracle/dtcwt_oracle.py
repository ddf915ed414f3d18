"""CPU oracle for the dual-tree complex wavelet transform (test infrastructure: only tests/ may import this).

The reference reaches the DTCWT through ``pytorch_wavelets.DTCWTForward / DTCWTInverse`` (py/wavelet_functions.py:56-73), an
un-vendored, absent dependency whose filter banks are data files: **parity unpinned**.  This file restates the published algorithm --
N. G. Kingsbury's dtwavexfm2 / dtwaveifm2 (DTCWT toolbox 4.3; "Complex wavelets for shift invariant analysis and filtering of
signals", ACHA 2001; "Design of Q-shift complex wavelets for image processing using frequency domain energy minimisation", ICIP
2003), which pytorch_wavelets implements -- in plain numpy, line by line in the toolbox's own structure (colfilter / coldfilt /
colifilt / q2c / c2q), with pytorch_wavelets' tensor layout: ``yl [B, C, H / 2**(J-1), W / 2**(J-1)]``, ``yh[j] [B, C, 6, h_j, w_j, 2]``
(orientations 15, 45, 75, 105, 135, 165 degrees; last axis real / imaginary).

Filter banks: `near_sym_a` (5 / 7 taps), `legall` (5 / 3) and `antonini` (9 / 7) in closed form, `qshift_a` (10 taps) from the published coefficients; they
are checked by their defining properties (tests/test_dtcwt_cpu.py: half-band product, orthonormal shifts), and the transform by perfect
reconstruction, by its 4:1 redundancy layout and by the orientation selectivity of its six subbands.
"""
from __future__ import annotations

import numpy as np

BIORT = {
    # (h0o, g0o): analysis / synthesis low-pass, odd lengths, unit DC gain; the high-pass pair is their alternating-sign mirror
    "near_sym_a": (np.array([-1, 5, 12, 5, -1]) / 20.0, np.array([-3, -15, 73, 170, 73, -15, -3]) / 280.0),
    "legall": (np.array([-1, 2, 6, 2, -1]) / 8.0, np.array([1, 2, 1]) / 4.0),
    # CDF 9 / 7 (factors of the 8th-order maximally flat half-band filter; the published 12-digit table to full precision)
    "antonini": (np.array([0.02674875741081003, -0.01686411844287496, -0.07822326652899003, 0.2668641184428749, 0.60294901823636,
                           0.2668641184428749, -0.07822326652899003, -0.01686411844287496, 0.02674875741081003]),
                 np.array([-0.04563588155712507, -0.02877176311425014, 0.29563588155712506, 0.5575435262285002, 0.29563588155712506,
                           -0.02877176311425014, -0.04563588155712507])),
}
QSHIFT = {
    "qshift_a": np.array([0.0511304052838317, -0.0139753702468888, -0.109836051665971, 0.263839561058938, 0.766628467793037,
                          0.563655710127052, 0.000873622695217097, -0.100231219507476, -0.00168968127252815, -0.00618188189211644]),
}


def biort(name: str):
    """h0o, g0o, h1o, g1o: h1o[n] = (-1)^n g0o[n] and g1o[n] = -(-1)^n h0o[n] (n from 0: both centre taps negative), so that
    h0o * g0o + h1o * g1o = delta."""
    h0o, g0o = BIORT[name]

    def alt(v):
        c = (len(v) - 1) // 2
        return v * np.array([(-1.0) ** (i - c) for i in range(len(v))])

    return h0o, g0o, -alt(g0o), -alt(h0o)


def qshift(name: str):
    """h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b: tree b is tree a reversed, synthesis = the other tree's analysis filter."""
    h0a = QSHIFT[name]
    n = len(h0a)
    h0b = h0a[::-1]
    h1a = h0b * np.array([(-1.0) ** i for i in range(n)])
    h1b = h1a[::-1]
    return h0a, h0b, h0b, h0a, h1a, h1b, h1b, h1a


def reflect(x, minx, maxx):
    """The toolbox's reflect(): fold the (integer) positions x into [minx, maxx] with half-sample symmetry."""
    x = np.asarray(x, dtype=np.float64)
    rng = maxx - minx
    y = np.mod(x - minx, 2 * rng)
    y = np.where(y > rng, 2 * rng - y, y) + minx
    return np.rint(y).astype(np.int64)


def _conv_valid(u, h):
    """conv2(u, h, 'valid') along axis 0: out[i] = sum_k h[k] u[i + len(h) - 1 - k]."""
    L = len(h)
    n = u.shape[0] - L + 1
    out = np.zeros((n, *u.shape[1:]), dtype=u.dtype)
    for k in range(L):
        out += h[k] * u[L - 1 - k: L - 1 - k + n]
    return out


def colfilter(X, h):
    """Filter the columns of X (axis 0) with the odd-length filter h, symmetric extension, same size out."""
    r = X.shape[0]
    m2 = len(h) // 2
    xe = reflect(np.arange(-m2, r + m2), -0.5, r - 0.5)
    return _conv_valid(X[xe], h)


def coldfilt(X, ha, hb):
    """Decimating dual-tree column filter (toolbox coldfilt.m): rows r -> r / 2, the two trees' outputs interleaved."""
    r = X.shape[0]
    if r % 4:
        raise ValueError("No. of rows in X must be a multiple of 4!")
    m = len(ha)
    xe = reflect(np.arange(-m, r + m), -0.5, r - 0.5)
    hao, hae, hbo, hbe = ha[0::2], ha[1::2], hb[0::2], hb[1::2]
    t = np.arange(5, r + 2 * m - 2, 4)  # 0-based positions of the toolbox's t = 6:4:(r+2m-2)
    r2 = r // 2
    Y = np.zeros((r2, *X.shape[1:]), dtype=X.dtype)
    if np.sum(ha * hb) > 0:
        s1, s2 = slice(0, r2, 2), slice(1, r2, 2)
    else:
        s2, s1 = slice(0, r2, 2), slice(1, r2, 2)
    Y[s1] = _conv_valid(X[xe[t - 1]], hao) + _conv_valid(X[xe[t - 3]], hae)
    Y[s2] = _conv_valid(X[xe[t]], hbo) + _conv_valid(X[xe[t - 2]], hbe)
    return Y


def colifilt(X, ha, hb):
    """Interpolating dual-tree column filter (toolbox colifilt.m): rows r -> 2 r."""
    r = X.shape[0]
    if r % 2:
        raise ValueError("No. of rows in X must be a multiple of 2!")
    m = len(ha)
    m2 = m // 2
    Y = np.zeros((2 * r, *X.shape[1:]), dtype=X.dtype)
    xe = reflect(np.arange(-m2, r + m2), -0.5, r - 0.5)
    hao, hae, hbo, hbe = ha[0::2], ha[1::2], hb[0::2], hb[1::2]
    if m2 % 2 == 0:
        t = np.arange(3, r + m, 2)  # toolbox t = 4:2:(r+m)
        ta, tb = (t, t - 1) if np.sum(ha * hb) > 0 else (t - 1, t)
        Y[0::4] = _conv_valid(X[xe[tb - 2]], hae)
        Y[1::4] = _conv_valid(X[xe[ta - 2]], hbe)
        Y[2::4] = _conv_valid(X[xe[tb]], hao)
        Y[3::4] = _conv_valid(X[xe[ta]], hbo)
    else:
        t = np.arange(2, r + m - 1, 2)  # toolbox t = 3:2:(r+m-1)
        ta, tb = (t, t - 1) if np.sum(ha * hb) > 0 else (t - 1, t)
        Y[0::4] = _conv_valid(X[xe[tb]], hao)
        Y[1::4] = _conv_valid(X[xe[ta]], hbo)
        Y[2::4] = _conv_valid(X[xe[tb]], hae)
        Y[3::4] = _conv_valid(X[xe[ta]], hbe)
    return Y


def _cols(fn, X, *f):
    """fn along the rows axis (-2) of [..., H, W]."""
    return np.moveaxis(fn(np.moveaxis(X, -2, 0), *f), 0, -2)


def _rows(fn, X, *f):
    """fn along the columns axis (-1)."""
    return np.moveaxis(fn(np.moveaxis(X, -1, 0), *f), 0, -1)


def q2c(y):
    """Quads (a b / c d) -> two complex subbands ((a - d) + i (b + c)) / sqrt 2, ((a + d) + i (b - c)) / sqrt 2."""
    a, b, c, d = y[..., 0::2, 0::2], y[..., 0::2, 1::2], y[..., 1::2, 0::2], y[..., 1::2, 1::2]
    s = np.sqrt(0.5)
    return np.stack(((a - d) * s, (b + c) * s), axis=-1), np.stack(((a + d) * s, (b - c) * s), axis=-1)


def c2q(z1, z2):
    """Inverse of q2c."""
    s = np.sqrt(0.5)
    p, q = (z1 + z2) * s, (z1 - z2) * s  # p = a + i b, q = -d + i c
    out = np.zeros((*z1.shape[:-3], 2 * z1.shape[-3], 2 * z1.shape[-2]), dtype=z1.dtype)
    out[..., 0::2, 0::2] = p[..., 0]
    out[..., 0::2, 1::2] = p[..., 1]
    out[..., 1::2, 0::2] = q[..., 1]
    out[..., 1::2, 1::2] = -q[..., 0]
    return out


def forward(x, J: int, biort_name: str = "near_sym_a", qshift_name: str = "qshift_a"):
    """x [B, C, H, W] -> (yl, [yh_1 (finest) .. yh_J]); orientation order 15, 45, 75, 105, 135, 165 degrees."""
    x = np.asarray(x)
    if J == 0:
        return x, []
    h0o, _g0o, h1o, _g1o = biort(biort_name)
    h0a, h0b, _, _, h1a, h1b, _, _ = qshift(qshift_name)
    if x.shape[-2] % 2:
        x = np.concatenate((x, x[..., -1:, :]), axis=-2)
    if x.shape[-1] % 2:
        x = np.concatenate((x, x[..., :, -1:]), axis=-1)
    yh = []
    # level 1: odd-length filters, no decimation; the quads give the 2:1 subsampling
    lo, hi = _rows(colfilter, x, h0o), _rows(colfilter, x, h1o)      # along W
    ll = _cols(colfilter, lo, h0o)
    lh, hl, hh = _cols(colfilter, lo, h1o), _cols(colfilter, hi, h0o), _cols(colfilter, hi, h1o)
    yh.append(_bands(lh, hh, hl))
    for _ in range(1, J):
        if ll.shape[-2] % 4:
            ll = np.concatenate((ll[..., :1, :], ll, ll[..., -1:, :]), axis=-2)
        if ll.shape[-1] % 4:
            ll = np.concatenate((ll[..., :, :1], ll, ll[..., :, -1:]), axis=-1)
        lo, hi = _rows(coldfilt, ll, h0b, h0a), _rows(coldfilt, ll, h1b, h1a)
        ll = _cols(coldfilt, lo, h0b, h0a)
        lh, hl, hh = _cols(coldfilt, lo, h1b, h1a), _cols(coldfilt, hi, h0b, h0a), _cols(coldfilt, hi, h1b, h1a)
        yh.append(_bands(lh, hh, hl))
    return ll, yh


def _bands(lh, hh, hl):
    """lh = low along W / high along H (the +-15 degree pair), hh the diagonals (45, 135), hl the +-75 pair."""
    d15, d165 = q2c(lh)
    d45, d135 = q2c(hh)
    d75, d105 = q2c(hl)
    return np.stack((d15, d45, d75, d105, d135, d165), axis=2)


def inverse(yl, yh, biort_name: str = "near_sym_a", qshift_name: str = "qshift_a"):
    _h0o, g0o, _h1o, g1o = biort(biort_name)
    _, _, g0a, g0b, _, _, g1a, g1b = qshift(qshift_name)
    ll = np.asarray(yl)
    for j in range(len(yh) - 1, -1, -1):
        h = yh[j]
        lh, hh, hl = c2q(h[:, :, 0], h[:, :, 5]), c2q(h[:, :, 1], h[:, :, 4]), c2q(h[:, :, 2], h[:, :, 3])
        if j > 0:
            y1 = _cols(colifilt, ll, g0b, g0a) + _cols(colifilt, lh, g1b, g1a)
            y2 = _cols(colifilt, hl, g0b, g0a) + _cols(colifilt, hh, g1b, g1a)
            ll = _rows(colifilt, y1, g0b, g0a) + _rows(colifilt, y2, g1b, g1a)
            want_h, want_w = 2 * yh[j - 1].shape[3], 2 * yh[j - 1].shape[4]
            if ll.shape[-2] != want_h:
                ll = ll[..., 1:-1, :]
            if ll.shape[-1] != want_w:
                ll = ll[..., :, 1:-1]
        else:
            y1 = _cols(colfilter, ll, g0o) + _cols(colfilter, lh, g1o)
            y2 = _cols(colfilter, hl, g0o) + _cols(colfilter, hh, g1o)
            ll = _rows(colfilter, y1, g0o) + _rows(colfilter, y2, g1o)
    return ll
