"""CPU: oracle/dwt_oracle.py against PyWavelets 1.1.1 golden vectors (tests/golden/dwt.npz) and
size-independent properties (perfect reconstruction, linearity)."""
import numpy as np
import pytest

from oracle import dwt_oracle as dwo
from tests.conftest import GOLDEN

G = np.load(f"{GOLDEN}/dwt.npz", allow_pickle=False)
TAGS = sorted({k.split("__")[0] for k in G.files})


@pytest.mark.parametrize("tag", TAGS)
def test_wavedec2_and_waverec2_match_pywt(tag):
    wave, mode, level = G[f"{tag}__meta"]
    level = int(level)
    x = G[f"{tag}__x"]
    dec, rec_fn = (dwo.wavedec1, dwo.waverec1) if tag.startswith("d1_") else (dwo.wavedec2, dwo.waverec2)  # d1_*: pywt.wavedec / waverec
    yl, yh = dec(x, str(wave), str(mode), level)
    np.testing.assert_allclose(yl, G[f"{tag}__yl"], rtol=1e-12, atol=1e-12)
    for j in range(level):
        assert yh[j].shape == G[f"{tag}__yh{j}"].shape
        np.testing.assert_allclose(yh[j], G[f"{tag}__yh{j}"], rtol=1e-12, atol=1e-12)
    rec = rec_fn(G[f"{tag}__yl"], [G[f"{tag}__yh{j}"] for j in range(level)], str(wave), str(mode))
    np.testing.assert_allclose(rec, G[f"{tag}__rec"], rtol=1e-11, atol=1e-12)


def test_subband_sizes_of_the_configs():
    """db4/symmetric/J=5 on 128 -> 67,37,22,14,10 (yl 10); haar/periodization/J=3 -> 64,32,16 (yl 16)."""
    x = np.zeros((1, 1, 128, 128))
    yl, yh = dwo.wavedec2(x, "db4", "symmetric", 5)
    assert [b.shape[-1] for b in yh] == [67, 37, 22, 14, 10] and yl.shape[-1] == 10
    yl, yh = dwo.wavedec2(x, "haar", "periodization", 3)
    assert [b.shape[-1] for b in yh] == [64, 32, 16] and yl.shape[-1] == 16
    yl, yh = dwo.wavedec2(np.zeros((1, 1, 64, 64)), "db4", "symmetric", 5)
    assert [b.shape[-1] for b in yh] == [35, 21, 14, 10, 8]


@pytest.mark.parametrize("wave", ["haar", "db4", "sym5", "bior2.2", "coif2", "db10"])
@pytest.mark.parametrize("mode", dwo.MODES)
def test_perfect_reconstruction(wave, mode):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 2, 45, 38))
    yl, yh = dwo.wavedec2(x, wave, mode, 3)
    rec = dwo.waverec2(yl, yh, wave, mode)
    np.testing.assert_allclose(rec[..., :45, :38], x, rtol=0, atol=1e-9)


def test_linearity_and_fill_scales():
    rng = np.random.default_rng(8)
    a, b = rng.standard_normal((1, 1, 32, 32)), rng.standard_normal((1, 1, 32, 32))
    la, ha = dwo.wavedec2(a, "db4", "symmetric", 2)
    lb, hb = dwo.wavedec2(b, "db4", "symmetric", 2)
    lc, hc = dwo.wavedec2(2 * a - 3 * b, "db4", "symmetric", 2)
    np.testing.assert_allclose(lc, 2 * la - 3 * lb, atol=1e-12)
    np.testing.assert_allclose(hc[1], 2 * ha[1] - 3 * hb[1], atol=1e-12)
    assert dwo.expand_yh_scales(5, 3, 3) == ((3.0, 3.0, 3.0),) * 5
    assert dwo.expand_yh_scales(4, 3, [1.5, "fill", 0.5]) == ((1.5,) * 3, (1.5,) * 3, (1.5,) * 3, (0.5,) * 3)
    assert dwo.expand_yh_scales(3, 3, [[1, 2], 4.0]) == ((1.0, 2.0, 1.0), (4.0, 4.0, 4.0))
    with pytest.raises(ValueError):
        dwo.expand_yh_scales(3, 3, ["fill", 1.0])


def test_wavelet_cfg_identity():
    """diff scales 1, inject strength s: IDWT(u + s (c - u)) == u + s (c - u) (linearity + PR)."""
    rng = np.random.default_rng(9)
    c, u = rng.standard_normal((1, 2, 32, 32)), rng.standard_normal((1, 2, 32, 32))
    out = dwo.wavelet_cfg(c, u, "db4", "symmetric", 3, strength=0.7)
    np.testing.assert_allclose(out[..., :32, :32], u + 0.7 * (c - u), atol=1e-9)
