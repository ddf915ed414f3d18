"""kernel-trace workload: WaveletCFG with a per-orientation difference rule (band path), 256 latents, fp32 and fp64, 20 calls each"""
import importlib, math, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
b4, C, H, W = int(os.environ.get("WCFG_BATCH", "256")), 4, 128, 128
ms = types.SimpleNamespace(sigma_min=torch.tensor(0.03), sigma_max=torch.tensor(14.6), timestep=lambda sg: (
    999 * (1 - (sg.log() - math.log(0.03)) / (math.log(14.6) - math.log(0.03)))).clamp(0, 999))
cond, uncond, xin = (torch.randn(b4, C, H, W, device="cuda") for _ in range(3))
wargs = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": xin - cond, "uncond": xin - uncond, "input": xin, "cond_scale": 7.0,
         "sigma": torch.full((b4,), 7.0, device="cuda"), "model": types.SimpleNamespace(model_sampling=ms),
         "model_options": {"transformer_options": {"sample_sigmas": torch.cat([torch.linspace(14.6, 0.03, 20), torch.zeros(1)])}}}
for hp in (False, True):
    fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=[[3.0, 2.5, 2.0]] * 5), high_precision_mode=hp))
    for _ in range(20): fn(wargs)
torch.cuda.synchronize()
