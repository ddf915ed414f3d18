import os, sys, time, math, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
import sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
ms = types.SimpleNamespace(sigma_min=torch.tensor(0.03), sigma_max=torch.tensor(14.6), timestep=lambda sg: (999 * (1 - (sg.log() - math.log(0.03)) / (math.log(14.6) - math.log(0.03)))).clamp(0, 999))
b4 = 256
cond, uncond, xin = (torch.randn(b4, 4, 128, 128, device="cuda") for _ in range(3))
wargs = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": xin - cond, "uncond": xin - uncond, "input": xin, "cond_scale": 7.0,
         "sigma": torch.full((b4,), 7.0, device="cuda"), "model": types.SimpleNamespace(model_sampling=ms), "model_options": {}}
for hp in (True, False):
    fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=3.0), high_precision_mode=hp))
    for _ in range(3): fn(wargs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn(wargs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("fp64" if hp else "fp32", f"{dt*1e6:.1f} us/call", f"{16*65536*b4/dt/1e9:.0f} GB/s at 16N")
if os.environ.get("WCFG_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50): fn(wargs)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
