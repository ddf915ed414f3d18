"""CPU: the DTCWT oracle (numpy restatement of Kingsbury's dtwavexfm2 / dtwaveifm2; pytorch_wavelets is absent -> parity unpinned) is
checked by what defines the transform, and the product's per-axis tap tables (py/dtcwt.py) against the oracle's routines."""
import importlib

import numpy as np
import pytest

from oracle import dtcwt_oracle as dto


@pytest.fixture(scope="module")
def dt(pkg):
    return importlib.import_module("comfyui_sonar_amd.py.dtcwt")


def test_filter_banks_have_their_defining_properties(dt):
    for name in ("near_sym_a", "legall", "antonini"):
        h0o, g0o, h1o, g1o = dto.biort(name)
        assert abs(h0o.sum() - 1) < 1e-15 and abs(g0o.sum() - 1) < 1e-15 and abs(h1o.sum()) < 2e-16 * len(h1o) and abs(g1o.sum()) < 2e-16 * len(g1o)
        p = np.convolve(h0o, g0o) + np.convolve(h1o, g1o)  # undecimated analysis + synthesis = identity
        want = np.zeros_like(p)
        want[len(p) // 2] = 1.0
        assert np.abs(p - want).max() < 4e-16
        for mine, ref in zip(dt.biort_filters(name), (h0o, g0o, h1o, g1o)):
            assert np.array_equal(mine, ref)
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = dto.qshift("qshift_a")
    assert abs(h0a.sum() - np.sqrt(2)) < 1e-14
    for k in range(5):  # orthonormal under even shifts: the published coefficients, not a transcription slip
        assert abs(np.dot(h0a[: 10 - 2 * k], h0a[2 * k:]) - (k == 0)) < 1e-14
        assert abs(np.dot(h0a[: 10 - 2 * k], h1a[2 * k:])) < 1e-14 and abs(np.dot(h1a[: 10 - 2 * k], h0a[2 * k:])) < 1e-14
    assert np.array_equal(h0b, h0a[::-1]) and np.array_equal(g0a, h0b) and np.array_equal(g1b, h1a)
    for mine, ref in zip(dt.qshift_filters("qshift_a"), (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b)):
        assert np.array_equal(mine, ref)
    with pytest.raises(NotImplementedError):
        dt.biort_filters("near_sym_b")
    with pytest.raises(NotImplementedError):
        dt.qshift_filters("qshift_06")


@pytest.mark.parametrize("shape,levels", [((1, 2, 64, 64), 3), ((2, 1, 32, 48), 2), ((1, 1, 128, 128), 4), ((1, 1, 36, 52), 3), ((1, 1, 33, 47), 2),
                                          ((1, 1, 64, 64), 1), ((1, 1, 20, 28), 3)])
def test_oracle_reconstructs_perfectly(shape, levels):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(shape)
    for bi in ("near_sym_a", "legall", "antonini"):
        yl, yh = dto.forward(x, levels, bi)
        assert len(yh) == levels and all(h.shape[2] == 6 and h.shape[-1] == 2 for h in yh)
        h1, w1 = (shape[2] + 1) // 2, (shape[3] + 1) // 2
        assert yh[0].shape[3:5] == (h1, w1)
        assert yl.shape[-2:] == tuple(2 * v for v in yh[-1].shape[3:5])  # the low-pass sits at twice the resolution of the last bands
        back = dto.inverse(yl, yh, bi)
        assert np.abs(back[..., : shape[2], : shape[3]] - x).max() < 1e-13


def test_oracle_subbands_are_oriented_and_analytic():
    """A grating at 15, 45, ... 165 degrees lights up subband 0, 1, ... 5 (pytorch_wavelets' order), and hardly its mirror image
    (the complex wavelets are one-sided in frequency)."""
    yy, xx = np.mgrid[0:128, 0:128].astype(float)
    for o, ang in enumerate((15, 45, 75, 105, 135, 165)):
        a = np.deg2rad(ang)
        img = np.cos(2 * np.pi * 0.18 * (np.sin(a) * xx + np.cos(a) * yy))
        _, yh = dto.forward(img[None, None], 3)
        e = np.array([(yh[1][0, 0, k] ** 2).sum() for k in range(6)])
        assert int(np.argmax(e)) == o
        assert e[5 - o] < 0.3 * e[o]


def test_tap_tables_reproduce_the_oracle_stages(dt):
    rng = np.random.default_rng(1)
    h0o, g0o, h1o, g1o = dto.biort("near_sym_a")
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = dto.qshift("qshift_a")

    def dense(table, x):
        idx, coef = table
        return np.einsum("jk,jk...->j...", coef, x[idx])

    for n in (8, 12, 20, 64):
        x = rng.standard_normal((n, 3))
        for h in (h0o, h1o, g0o, g1o, dto.biort("legall")[1]):
            assert np.abs(dense(dt.table_odd(n, h), x) - dto.colfilter(x, h)).max() < 1e-14
        if n % 4 == 0:
            for ha, hb in ((h0b, h0a), (h1b, h1a)):
                assert np.abs(dense(dt.table_decimate(n, ha, hb), x) - dto.coldfilt(x, ha, hb)).max() < 1e-14
        for ha, hb in ((g0b, g0a), (g1b, g1a)):
            assert np.abs(dense(dt.table_interpolate(n, ha, hb), x) - dto.colifilt(x, ha, hb)).max() < 1e-14
    with pytest.raises(ValueError):
        dt.table_decimate(10, h0b, h0a)
