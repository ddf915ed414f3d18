"""Randomised differential runs of two kernel families against independent checkers:
  * the resampler (scale_samples, every mode) against torch.nn.functional.interpolate on the same device,
  * the 2-D / 1-D wavelet transform (Wavelet.forward / inverse, every wavelet family and extension mode) against oracle/dwt_oracle.py (numpy,
    pinned to PyWavelets vectors).
python scratch/fuzz_kernels.py [iterations] [seed]"""
import importlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sonar_pkg
import torch.nn.functional as F
from oracle import dwt_oracle as dwo
pkg = sonar_pkg.load(); pkg.hip_lib.load()
utils = importlib.import_module("comfyui_sonar_amd.py.utils"); wf = importlib.import_module("comfyui_sonar_amd.py.wavelet_functions")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
MODES = {"bilinear": dict(mode="bilinear", align_corners=False), "nearest": dict(mode="nearest"), "nearest-exact": dict(mode="nearest-exact"),
         "area": dict(mode="area"), "bicubic": dict(mode="bicubic", align_corners=False)}
for it in range(iters):
    b, c = rnd.randint(1, 3), rnd.choice([1, 3, 4])
    h, w, H, W = (rnd.randint(1, 70) for _ in range(4))
    mode = rnd.choice(list(MODES))
    x = torch.randn(b, c, h, w, device="cuda")
    try:
        got = utils.scale_samples(x, W, H, mode=mode)
        want = F.interpolate(x, size=(H, W), **MODES[mode])
        err = float((got - want).abs().max())
        if err > 3e-5 * max(1.0, float(want.abs().max())):
            print(f"[resample {it}] {mode} {(b, c, h, w)} -> {(H, W)}: max diff {err:.3e}", flush=True); bad += 1
    except Exception as exc:  # noqa: BLE001
        print(f"[resample {it}] {mode} {(b, c, h, w)} -> {(H, W)}: {type(exc).__name__}: {str(exc)[:120]}", flush=True); bad += 1
WAVES = ["haar", "db2", "db3", "db4", "db6", "db8", "db10", "sym2", "sym4", "sym5", "sym8", "coif1", "coif2", "coif3", "bior1.3", "bior2.2", "bior3.5", "bior4.4", "rbio2.2", "dmey"]
DMODES = ["zero", "symmetric", "reflect", "periodization", "periodic", "constant"]
for it in range(iters):
    wave, mode = rnd.choice(WAVES), rnd.choice(DMODES)
    one_d = rnd.random() < 0.25
    level = rnd.randint(1, 4)
    b, c = rnd.randint(1, 2), rnd.choice([1, 3, 4])
    h, w = rnd.randint(2, 70), rnd.randint(2, 70)
    dtype = rnd.choice([torch.float64, torch.float32])
    shape = (b, c, h * w // 4 + 2) if one_d else (b, c, h, w)
    try:
        wv = wf.Wavelet(wave=wave, level=level, mode=mode, use_1d_dwt=one_d)
    except Exception as exc:  # noqa: BLE001
        skipped = globals().get('skipped', 0) + 1
        continue
    x = torch.randn(shape, dtype=torch.float64)
    try:
        yl, yh = wv.forward(x.to(dtype).cuda())
        if one_d:
            oyl, oyh = dwo.wavedec1(x.numpy(), wave, mode, level)
        else:
            oyl, oyh = dwo.wavedec2(x.numpy(), wave, mode, level)
        tol = 1e-10 if dtype == torch.float64 else 5e-5
        peak = max(1.0, float(np.abs(oyl).max()))
        errs = [float(np.abs(yl.cpu().double().numpy() - oyl).max())] + [float(np.abs(a.cpu().double().numpy() - o).max()) for a, o in zip(yh, oyh)]
        rec = wv.inverse(yl, yh)
        orec = dwo.waverec1(oyl, oyh, wave, mode) if one_d else dwo.waverec2(oyl, oyh, wave, mode)
        errs.append(float(np.abs(rec.cpu().double().numpy() - orec).max()))
        if max(errs) > tol * peak * 4:
            print(f"[dwt {it}] {wave} {mode} level {level} {shape} {dtype}: max diff {max(errs):.3e}", flush=True); bad += 1
    except Exception as exc:  # noqa: BLE001
        msg = str(exc)[:140]
        print(f"[dwt {it}] {wave} {mode} level {level} {shape} {dtype}: {type(exc).__name__}: {msg}", flush=True); bad += 1
print(f"{2 * iters} runs ({globals().get('skipped', 0)} wavelet configurations refused by the constructor), {bad} problems")
