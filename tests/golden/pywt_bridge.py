"""Container-only tool: a stand-in ``pytorch_wavelets`` module whose arithmetic is the REAL PyWavelets 1.1.1.

The reference's wavelet rows call ``pytorch_wavelets.DWTForward / DWTInverse / DWT1DForward / DWT1DInverse``
(``py/wavelet_functions.py:56-80``); that package is absent from the image, and the interpreter that holds
PyWavelets (``/opt/conda/bin/python3.9``) has no torch.  This file bridges the two so that
``tests/golden/make_golden.py`` can run the reference's OWN ``Wavelet``, ``wavelet_scaling``, ``WaveletCFG`` and
``WaveletFilteredNoiseGenerator`` code end to end with nothing of this repository's oracle or product in the loop:

* run as ``python3.9 pywt_bridge.py --serve`` it is a tiny request loop around ``pywt.wavedec2 / waverec2 /
  wavedec / waverec`` (length-prefixed JSON header + raw array bytes on stdin / stdout);
* imported from the torch interpreter it provides ``install(ref_wavelet_functions_module)``, which plugs classes
  with pytorch_wavelets' constructor signatures and output layout (``yl``, ``yh[j] [B, C, 3, h, w]`` finest first,
  orientations = pywt's (cH, cV, cD); 1-D: ``yh[j] [B, C, l]``) into the reference module and flips its
  ``HAVE_WAVELETS``.

pytorch_wavelets documents its DWT as matching ``pywt.wavedec2`` coefficient for coefficient in that layout; the
reference has no tests at this boundary, so PyWavelets is the arithmetic the fixtures pin.
"""
from __future__ import annotations

import json
import os
import struct
import subprocess
import sys

import numpy as np

PYWT_PYTHON = os.environ.get("SONAR_PYWT_PYTHON", "/opt/conda/bin/python3.9")


def _send(stream, header: dict, arrays) -> None:
    header = dict(header, arrays=[[str(a.dtype), list(a.shape)] for a in arrays])
    blob = json.dumps(header).encode()
    stream.write(struct.pack("<I", len(blob)))
    stream.write(blob)
    for a in arrays:
        stream.write(np.ascontiguousarray(a).tobytes())
    stream.flush()


def _recv(stream):
    raw = stream.read(4)
    if len(raw) < 4:
        return None, None
    header = json.loads(stream.read(struct.unpack("<I", raw)[0]).decode())
    arrays = []
    for dtype, shape in header["arrays"]:
        n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        arrays.append(np.frombuffer(stream.read(n), dtype=dtype).reshape(shape).copy())
    return header, arrays


def serve() -> None:  # runs under the PyWavelets interpreter
    import warnings

    import pywt

    warnings.simplefilter("ignore")
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        header, arrays = _recv(inp)
        if header is None:
            return
        op, wave, mode = header["op"], header["wave"], header["mode"]
        if op == "version":
            _send(out, {"version": pywt.__version__}, [])
        elif op == "wavedec2":
            coeffs = pywt.wavedec2(arrays[0], wave, mode=mode, level=header["level"], axes=(-2, -1))
            res = [coeffs[0]]
            for j in range(header["level"]):  # finest first
                res.append(np.stack(coeffs[header["level"] - j], axis=-3))
            _send(out, {}, res)
        elif op == "waverec2":
            yl, yh = arrays[0], arrays[1:]
            ll = yl
            # pytorch_wavelets DWTInverse walks coarse -> fine and drops the extra row / column of the running
            # approximation when it is one larger than the band; pywt.idwt2 per level keeps exactly that order.
            for band in reversed(yh):
                if ll.shape[-2] > band.shape[-2]:
                    ll = ll[..., :-1, :]
                if ll.shape[-1] > band.shape[-1]:
                    ll = ll[..., :-1]
                ll = pywt.idwt2((ll, (band[..., 0, :, :], band[..., 1, :, :], band[..., 2, :, :])), wave, mode=mode, axes=(-2, -1))
            _send(out, {}, [ll])
        elif op == "wavedec":
            coeffs = pywt.wavedec(arrays[0], wave, mode=mode, level=header["level"], axis=-1)
            _send(out, {}, [coeffs[0], *[coeffs[header["level"] - j] for j in range(header["level"])]])
        elif op == "waverec":
            lo = arrays[0]
            for band in reversed(arrays[1:]):
                if lo.shape[-1] > band.shape[-1]:
                    lo = lo[..., :-1]
                lo = pywt.idwt(lo, band, wave, mode=mode, axis=-1)
            _send(out, {}, [lo])
        else:
            raise SystemExit(f"unknown op {op}")


class _Client:
    proc = None

    @classmethod
    def call(cls, header: dict, arrays):
        if cls.proc is None:
            cls.proc = subprocess.Popen([PYWT_PYTHON, os.path.abspath(__file__), "--serve"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                        stderr=subprocess.DEVNULL)
        _send(cls.proc.stdin, header, arrays)
        hdr, out = _recv(cls.proc.stdout)
        if hdr is None:
            raise RuntimeError("the PyWavelets bridge process died")
        return hdr, out


def pywt_version() -> str:
    return _Client.call({"op": "version", "wave": "", "mode": ""}, [])[0]["version"]


def _modules():
    import torch

    class _Base:
        def to(self, *_a, **_k):  # the reference moves the transform objects around; these are stateless
            return self

    def t2n(t):
        return t.detach().cpu().numpy()

    def n2t(a, like):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dtype=like.dtype, device=like.device)

    class DWTForward(_Base):
        op = "wavedec2"

        def __init__(self, J=1, wave="db1", mode="zero"):
            self.J, self.wave, self.mode = J, wave, mode

        def __call__(self, x):
            _, out = _Client.call({"op": self.op, "wave": self.wave, "mode": self.mode, "level": self.J}, [t2n(x)])
            return n2t(out[0], x), [n2t(a, x) for a in out[1:]]

    class DWTInverse(_Base):
        op = "waverec2"

        def __init__(self, wave="db1", mode="zero"):
            self.wave, self.mode = wave, mode

        def __call__(self, coeffs):
            yl, yh = coeffs
            _, out = _Client.call({"op": self.op, "wave": self.wave, "mode": self.mode}, [t2n(yl), *[t2n(b) for b in yh]])
            return n2t(out[0], yl)

    class DWT1DForward(DWTForward):
        op = "wavedec"

    class DWT1DInverse(DWTInverse):
        op = "waverec"

    class _NoDTCWT:
        def __init__(self, *a, **k):
            raise NotImplementedError("DTCWT filter banks are not available in this container")

    return dict(DWTForward=DWTForward, DWTInverse=DWTInverse, DWT1DForward=DWT1DForward, DWT1DInverse=DWT1DInverse,
                DTCWTForward=_NoDTCWT, DTCWTInverse=_NoDTCWT)


def install(ref_wavelet_functions, wavelist=()) -> None:
    """Plug the bridge into the imported reference module (``sonar_ref.wavelet_functions``)."""
    import types

    ref_wavelet_functions.ptwav = types.SimpleNamespace(**_modules())
    ref_wavelet_functions.pywt = types.SimpleNamespace(wavelist=lambda: list(wavelist))
    ref_wavelet_functions.HAVE_WAVELETS = True


if __name__ == "__main__" and "--serve" in sys.argv:
    serve()
