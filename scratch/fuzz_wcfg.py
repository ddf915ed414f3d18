"""Randomised WaveletCFG calls (random latent shapes incl. odd sizes and video latents, wavelets, extension modes, levels, scale tables,
blends, targets, precisions, rule windows) on the device against oracle/dwt_oracle.py's wavelet_cfg_call (numpy; pinned to the reference's
own outputs by tests/test_wavelet_oracle_cpu.py).  python scratch/fuzz_wcfg.py [iterations] [seed]"""
import importlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sonar_pkg
from oracle import dwt_oracle as dwo
from tests import wavelet_helpers as wh
from tests.golden import wavelet_cases as wc
pkg = sonar_pkg.load(); pkg.hip_lib.load()
mod = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WAVES = ["haar", "db2", "db4", "db6", "sym4", "sym5", "coif2", "bior2.2", "bior4.4"]
MODES = ["zero", "symmetric", "reflect", "periodization", "periodic", "constant"]


def scales(level):
    kind = rnd.choice(["scalar", "list", "nested", "fill"])
    if kind == "scalar":
        return rnd.choice([3.0, 0.5, 1.0, 2.0])
    if kind == "list":
        return [rnd.choice([3.0, 0.5, 1.5, 1.0]) for _ in range(level)]
    if kind == "nested":
        return [[rnd.choice([3.0, 1.0, 0.5]) for _ in range(3)] if rnd.random() < 0.5 else rnd.choice([2.0, 0.25]) for _ in range(level)]
    return [rnd.choice([2.0, 0.5]), "fill"][: max(level, 2)] if level >= 2 else 2.0


bad = 0
for it in range(iters):
    level = rnd.randint(1, 4)
    video = rnd.random() < 0.15
    one_d = (not video) and rnd.random() < 0.12
    h, w = rnd.choice([(32, 32), (24, 40), (17, 23), (33, 31), (64, 48), (16, 16), (50, 38), (128, 128)])
    shape = (1, 4, 3, h, w) if video else (rnd.randint(1, 3), rnd.choice([3, 4]), h, w)
    params = dict(wave=rnd.choice(WAVES), level=level, padding_mode=rnd.choice(MODES), high_precision_mode=rnd.random() < 0.5,
                  difference=dict(yl_scale=rnd.choice([5.0, 1.0, 0.5]), yh_scales=scales(level)),
                  difference_blend_mode=rnd.choice(["inject", "lerp", "subtract_b"]), difference_blend_strength=rnd.choice([1.0, 0.35, 0.8]),
                  target_mode=rnd.choice(["denoised", "denoised", "noise", "noise_norm"]), blend_mode=rnd.choice(["lerp", "lerp", "inject"]),
                  blend_strength=rnd.choice([1.0, 1.0, 0.5]))
    if one_d:
        params["use_1d_dwt"] = True
        params["difference"]["yh_scales"] = rnd.choice([3.0, [2.0, 0.5, "fill"]]) if level >= 2 else 3.0
    for name in ("cond", "uncond", "final"):
        if rnd.random() < 0.25 and not one_d:
            params[name] = dict(yl_scale=rnd.choice([1.0, 1.2]), yh_scales=scales(level))
    if rnd.random() < 0.2:
        # (periodization and the other modes produce bands of different lengths: mixing them between analysis and synthesis has no defined
        # result -- the numpy checker refuses, pytorch_wavelets is not here to say what the reference does)
        per = params["padding_mode"] == "periodization"
        params["inv_padding_mode"] = rnd.choice([m for m in MODES if (m == "periodization") == per])
    if rnd.random() < 0.2:
        params.update(start_sigma=rnd.choice([14.0, 8.0]), end_sigma=rnd.choice([5.0, 1.0]))
    sigma = rnd.choice([7.0, 3.0, 9.5, [6.0, 2.0, 4.0][: shape[0]]])
    case = dict(shape=shape, sigma=sigma, params=params)
    name = f"fuzz{it}"
    try:
        args_cpu = wh.wcfg_args(case, name, wc.FakeModel())
        kw = wh.resolve_for_oracle(mod, case, args_cpu)
        np_args = wh.numpy_args(args_cpu)
        if kw is None:
            want = np_args["input"] - ((np_args["cond_denoised"] - np_args["uncond_denoised"]) * np.float32(np_args["cond_scale"]) + np_args["uncond_denoised"])
        else:
            want = dwo.wavelet_cfg_call(np_args, **kw)
    except Exception as exc:  # noqa: BLE001 -- the oracle refuses: the product must refuse too
        try:
            wh.build_wcfg(mod, case)(wh.wcfg_args(case, name, wc.FakeModel(), device="cuda"))
            print(f"[{it}] oracle raised {type(exc).__name__} ({str(exc)[:80]}), the product did not: {params} {shape}", flush=True); bad += 1
        except Exception:  # noqa: BLE001
            pass
        continue
    try:
        got = wh.build_wcfg(mod, case)(wh.wcfg_args(case, name, wc.FakeModel(), device="cuda")).cpu().numpy()
    except Exception as exc:  # noqa: BLE001
        print(f"[{it}] {type(exc).__name__}: {str(exc)[:140]}  {params} {shape} sigma {sigma}", flush=True); bad += 1
        continue
    tol = (2e-6 if (kw is None or kw["high_precision"]) else 5e-5) * max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) if got.shape == want.shape else float("inf")
    if not err <= tol:
        print(f"[{it}] max diff {err:.3e} > {tol:.1e}  {params} {shape} sigma {sigma}", flush=True); bad += 1
print(f"{iters} calls, {bad} problems")
