#!/usr/bin/env python3
"""Golden vectors for the 2-D DWT / IDWT from PyWavelets 1.1.1 (the arithmetic the reference reaches
through pytorch_wavelets, which is un-vendored and unpinned — SURVEY.md §8c).  Run with the interpreter
that has pywt:   /opt/conda/bin/python3.9 tests/golden/make_dwt_golden.py
Writes tests/golden/dwt.npz: inputs, per-level coefficients (pytorch_wavelets layout: yl, yh[j][:, :, (cH,cV,cD)]
finest first) and reconstructions."""
import os

import numpy as np
import pywt

rng = np.random.default_rng(1234)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dwt.npz")
cases = {}
specs = [
    # tag, shape, wavelet, mode, level
    ("db4_sym_l5_128", (1, 2, 128, 128), "db4", "symmetric", 5),      # WaveletCFG default (py/wavelet_cfg.py:468-479)
    ("haar_per_l3_128", (1, 2, 128, 128), "haar", "periodization", 3),  # WaveletFilteredNoise default
    ("db4_sym_l3_odd", (2, 3, 37, 50), "db4", "symmetric", 3),
    ("db4_zero_l2_odd", (2, 3, 37, 50), "db4", "zero", 2),
    ("sym5_reflect_l2", (1, 2, 40, 33), "sym5", "reflect", 2),
    ("db2_per_l3_odd", (1, 2, 37, 50), "db2", "periodization", 3),
    ("bior22_periodic_l2", (1, 2, 24, 30), "bior2.2", "periodic", 2),
    ("coif1_constant_l2", (1, 2, 21, 18), "coif1", "constant", 2),
    ("haar_sym_l1_tiny", (1, 1, 5, 3), "haar", "symmetric", 1),
    ("db4_sym_l1_short", (1, 1, 6, 4), "db4", "symmetric", 1),        # filter longer than the signal
    ("db4_sym_l5_64", (1, 1, 64, 64), "db4", "symmetric", 5),
]
for tag, shape, wave, mode, level in specs:
    x = rng.standard_normal(shape)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        coeffs = pywt.wavedec2(x, wave, mode=mode, level=level, axes=(-2, -1))
        rec = pywt.waverec2(coeffs, wave, mode=mode, axes=(-2, -1))
    cases[f"{tag}__x"] = x
    cases[f"{tag}__yl"] = coeffs[0]
    for j in range(level):  # finest first, like pytorch_wavelets
        ch, cv, cd = coeffs[level - j]
        cases[f"{tag}__yh{j}"] = np.stack([ch, cv, cd], axis=2)
    cases[f"{tag}__rec"] = rec
    cases[f"{tag}__meta"] = np.array([wave, mode, str(level)])
# 1-D transform (Wavelet(use_1d_dwt=True) -> pytorch_wavelets DWT1DForward / DWT1DInverse == pywt.wavedec / waverec);
# drawn after the 2-D cases so those keep their values
specs1d = [
    ("d1_db4_sym_l5_1024", (1, 2, 1024), "db4", "symmetric", 5),
    ("d1_haar_per_l3_256", (2, 2, 256), "haar", "periodization", 3),
    ("d1_db2_per_l3_odd", (1, 3, 37), "db2", "periodization", 3),
    ("d1_sym5_reflect_l2", (1, 2, 101), "sym5", "reflect", 2),
    ("d1_db4_zero_l2", (2, 1, 50), "db4", "zero", 2),
    ("d1_bior22_periodic_l2", (1, 2, 30), "bior2.2", "periodic", 2),
    ("d1_coif1_constant_l2", (1, 2, 21), "coif1", "constant", 2),
    ("d1_db4_sym_l1_short", (1, 1, 5), "db4", "symmetric", 1),
]
for tag, shape, wave, mode, level in specs1d:
    x = rng.standard_normal(shape)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        coeffs = pywt.wavedec(x, wave, mode=mode, level=level, axis=-1)
        rec = pywt.waverec(coeffs, wave, mode=mode, axis=-1)
    cases[f"{tag}__x"] = x
    cases[f"{tag}__yl"] = coeffs[0]
    for j in range(level):
        cases[f"{tag}__yh{j}"] = coeffs[level - j]
    cases[f"{tag}__rec"] = rec
    cases[f"{tag}__meta"] = np.array([wave, mode, str(level)])
np.savez_compressed(OUT, **cases)
print(OUT, os.path.getsize(OUT) // 1024, "KiB")
