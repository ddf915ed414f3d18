#!/usr/bin/env python3
"""Dumps the decomposition / reconstruction taps of every discrete PyWavelets wavelet to
comfyui-sonar_amd/wavelet_taps.json.  The reference takes its taps from pywt at run time
(py/wavelet_functions.py:13-14 via pytorch_wavelets); pywt is not importable by the product's
interpreter, so the table (mathematical constants) is generated once with PyWavelets 1.1.1:

    /opt/conda/bin/python3.9 tools/make_wavelet_taps.py
"""
import json
import os

import pywt

out = {"pywt_version": pywt.__version__, "wavelets": {}}
for name in pywt.wavelist(kind="discrete"):
    w = pywt.Wavelet(name)
    out["wavelets"][name] = {"dec_lo": list(w.dec_lo), "dec_hi": list(w.dec_hi), "rec_lo": list(w.rec_lo), "rec_hi": list(w.rec_hi)}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "comfyui-sonar_amd", "wavelet_taps.json")
with open(path, "w") as fh:
    json.dump(out, fh, separators=(",", ":"))
print(path, len(out["wavelets"]), "wavelets", os.path.getsize(path), "bytes")
