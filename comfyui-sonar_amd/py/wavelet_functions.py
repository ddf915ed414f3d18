"""2-D wavelet split on MI355X (API of the reference's ``py/wavelet_functions.py``).

The reference wraps ``pytorch_wavelets`` (DWTForward / DWTInverse) and takes filter taps from ``pywt``;
here the transform itself is a set of HIP kernels (``sonar_dwt2_{fwd,inv}_{f32,f64}``) with PyWavelets
semantics, and the taps come from a table generated once with PyWavelets 1.1.1
(``wavelet_taps.json``, tools/make_wavelet_taps.py).  Output layout is pytorch_wavelets':
``yl [B,C,h_J,w_J]`` and ``yh[j] [B,C,3,h_j,w_j]`` (finest level first; orientations LH, HL, HH ==
horizontal, vertical, diagonal == pywt cH, cV, cD).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Optional, Sequence

import torch

from .. import hip_lib
from .utils import fallback

HAVE_WAVELETS = True  # the transform is built in; no optional dependency
_TABLE = None


def _taps(name: str) -> dict:
    global _TABLE
    if _TABLE is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavelet_taps.json")
        with open(path) as fh:
            _TABLE = json.load(fh)["wavelets"]
    try:
        return _TABLE[name]
    except KeyError:
        raise ValueError(f"Unknown wavelet name '{name}', check wavelist() for the list of available builtin wavelets") from None


class Wavelet:
    """py/wavelet_functions.py:23-145: the 2-D DWT, the 1-D DWT (``use_1d_dwt``, [B, C, L] inputs; yh[j] is then [B, C, l_j]) and the
    dual-tree complex transform (``use_dtcwt``: ``py/dtcwt.py``, yh[j] [B, C, 6, h_j, w_j, 2]; the published algorithm with the
    `near_sym_a` / `legall` and `qshift_a` filter banks -- the other banks are data files of the absent ``pytorch_wavelets`` / ``dtcwt``
    packages and raise; parity unpinned)."""

    DEFAULT_MODE = "symmetric"
    DEFAULT_LEVEL = 3
    DEFAULT_WAVE = "db4"
    DEFAULT_USE_1D_DWT = False
    DEFAULT_USE_DTCWT = False
    DEFAULT_QSHIFT = "qshift_a"
    DEFAULT_BIORT = "near_sym_a"

    def __init__(self, *, wave: str = DEFAULT_WAVE, level: int = DEFAULT_LEVEL, mode: str = DEFAULT_MODE,
                 use_1d_dwt: bool = DEFAULT_USE_1D_DWT, use_dtcwt: bool = DEFAULT_USE_DTCWT, biort: str = DEFAULT_BIORT,
                 qshift: str = DEFAULT_QSHIFT, inv_wave: Optional[str] = None, inv_mode: Optional[str] = None,
                 inv_biort: Optional[str] = None, inv_qshift=None, device=None):
        self.use_dtcwt = bool(use_dtcwt)
        self.use_1d_dwt = bool(use_1d_dwt) and not self.use_dtcwt  # the reference tests use_dtcwt first (:56-61)
        if self.use_dtcwt:
            from .dtcwt import DTCWT

            if mode != "symmetric" or fallback(inv_mode, mode) != "symmetric":
                raise NotImplementedError("DTCWT: symmetric extension only (pytorch_wavelets' default)")
            self.level, self.mode, self.inv_mode = int(level), mode, fallback(inv_mode, mode)
            self.wave = self.inv_wave = None  # no (dec, rec) tap pair: the DWT fast paths of WaveletCFG do not apply
            self._dt = DTCWT(level=self.level, biort=biort, qshift=qshift, inv_biort=inv_biort, inv_qshift=inv_qshift)
            self.device, self.dtype = device, None
            return
        if mode not in hip_lib.DWT_MODE_IDS or fallback(inv_mode, mode) not in hip_lib.DWT_MODE_IDS:
            raise ValueError(f"Unknown padding mode {mode!r}; valid: {', '.join(self.modelist())}")
        self.level = int(level)
        self.mode = mode
        self.inv_mode = fallback(inv_mode, mode)
        self.wave, self.inv_wave = wave, fallback(inv_wave, wave)
        fwd, inv = _taps(wave), _taps(fallback(inv_wave, wave))
        self.dec_lo, self.dec_hi = fwd["dec_lo"], fwd["dec_hi"]
        self.rec_lo, self.rec_hi = inv["rec_lo"], inv["rec_hi"]
        self.device, self.dtype = device, None

    # ---- transform
    def forward(self, t: torch.Tensor, *, forward_function: Optional[Callable] = None):
        """DWTForward(J=level): returns (yl, [yh_0 (finest), ..., yh_{J-1}])."""
        if forward_function is not None:
            return forward_function(t)
        if getattr(self, "use_dtcwt", False):
            return self._dt.forward(t)
        one_d = getattr(self, "use_1d_dwt", False)
        if t.ndim != (3 if one_d else 4):
            raise hip_lib.SonarHipError("Wavelet.forward expects a [B, C, L] tensor in 1-D mode" if one_d else "Wavelet.forward expects a [B, C, H, W] tensor")
        step = hip_lib.dwt1_forward if one_d else hip_lib.dwt2_forward
        ll = t.contiguous()
        yh = []
        for _ in range(self.level):
            ll, hi = step(ll, self.dec_lo, self.dec_hi, self.mode)
            yh.append(hi)
        return ll, yh

    def _inverse(self, yl: torch.Tensor, yh: Sequence) -> torch.Tensor:
        if getattr(self, "use_dtcwt", False):
            return self._dt.inverse((yl, tuple(yh)))
        step = hip_lib.dwt1_inverse if getattr(self, "use_1d_dwt", False) else hip_lib.dwt2_inverse
        ll = yl.contiguous()
        for hi in reversed(tuple(yh)):
            # a coarser ll may be one row/column larger than the band; the kernel reads only its leading block
            ll = step(ll, hi.contiguous(), self.rec_lo, self.rec_hi, self.inv_mode)
        return ll

    def inverse(self, yl: torch.Tensor, yh: Sequence, *, inverse_function: Optional[Callable] = None,
                two_step_inverse: bool = False) -> torch.Tensor:
        inv = fallback(inverse_function, lambda pair: self._inverse(*pair))
        if not two_step_inverse:
            return inv((yl, yh))
        result = inv((torch.zeros_like(yl), yh))
        result += inv((yl, tuple(torch.zeros_like(band) for band in yh)))
        return result

    def to(self, *args, copy: bool = False, **kwargs) -> "Wavelet":
        """Taps are host constants and the kernels follow the input dtype: nothing to move."""
        o = Wavelet.__new__(Wavelet) if copy else self
        if copy:
            o.__dict__.update(self.__dict__)
        o.device = kwargs.get("device", o.device)
        o.dtype = kwargs.get("dtype", o.dtype)
        return o

    # ---- catalogues
    @staticmethod
    def wavelist() -> tuple:
        _taps("haar")
        return tuple(_TABLE.keys())

    @staticmethod
    def biortlist() -> tuple:
        return ("near_sym_a", "near_sym_b", "antonini", "legall")

    @staticmethod
    def qshiftlist() -> tuple:
        return ("qshift_a", "qshift_b", "qshift_c", "qshift_d", "qshift_06")

    @staticmethod
    def modelist() -> tuple:
        return ("symmetric", "zero", "reflect", "replicate", "periodization", "periodic", "constant")


def expand_yh_scales(yh: Sequence, *, yh_scales=1.0):
    """py/wavelet_functions.py:148-190: per-band, per-orientation scale table; one ``"fill"`` entry repeats
    the previous band's scales up to the number of bands."""
    nbands = len(yh)
    shape = yh[0].shape
    norient = shape[2] if len(shape) > 3 else 1
    if isinstance(yh_scales, (float, int)):
        return ((float(yh_scales),) * norient,) * nbands
    ones = (1.0,) * norient
    rows = []
    for band in yh_scales:
        if isinstance(band, (float, int)):
            rows.append((float(band),) * norient)
        elif isinstance(band, (tuple, list)):
            vals = tuple(float(v) for v in band[:norient])
            rows.append(vals + ones[: norient - len(vals)])
        else:
            rows.append(band)
    rows = tuple(rows)
    if "fill" in rows:
        k = rows.index("fill")
        if "fill" in rows[k + 1:]:
            raise ValueError("Only one fill allowed.")
        if k == 0 or len(rows) < 2:
            raise ValueError("Invalid fill value, cannot be in the first position or the only item.")
        missing = nbands - (len(rows) - 1)
        rows = (*rows[:k], *((rows[k - 1],) * max(missing, 0)), *rows[k + 1:])
    return rows[:nbands]


def wavelet_scaling(yl: torch.Tensor, yh: Sequence, yl_scale, yh_scales, *, in_place: bool = False) -> tuple:
    """py/wavelet_functions.py:193-216."""
    if not in_place:
        yl = yl.clone()
        yh = tuple(band.clone() for band in yh)
    if yl_scale != 1.0:
        hip_lib.band_scale_(yl, [float(yl_scale)])
    table = expand_yh_scales(yh, yh_scales=1.0 if yh_scales is None else yh_scales)
    for scales, band in zip(table, yh):
        if isinstance(scales, (int, float)):
            scales = (float(scales),)
        if band.ndim <= 3:
            # 1-D transform, bands [B, C, l]: the reference's ``ht[:, :, lidx] *= scales[lidx]`` walks axis 2 -- the coefficient axis
            # here -- for lidx < len(scales) == 1, so only the first coefficient of every row is scaled (:212-215; golden `oned_*`)
            hip_lib.band_scale_head_(band, scales[0])
            continue
        norient = band.shape[2]
        full = tuple(scales) + (1.0,) * (norient - len(scales))
        hip_lib.band_scale_(band, full[:norient])
    return (yl, yh)


def wavelet_blend(a: tuple, b: tuple, *, yl_factor, blend_function: Callable, yh_factor=None,
                  yh_blend_function: Optional[Callable] = None) -> tuple:
    """py/wavelet_functions.py:219-238."""
    if yh_factor is None:
        yh_factor = yl_factor
    yh_blend_function = fallback(yh_blend_function, blend_function)
    return (blend_function(a[0], b[0], yl_factor), tuple(yh_blend_function(ta, tb, yh_factor) for ta, tb in zip(a[1], b[1])))
