"""WaveletCFG band route (per-orientation difference scales, cfg4 size) for a rocprofv3 kernel trace: argv[1] = fp64|fp32, argv[2] = diff|pair."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
hp, route = sys.argv[1] == "fp64", sys.argv[2]
if route == "pair":
    wc._reconstructs = lambda w: False
b = 256
cond, uncond, x = (torch.randn(b, 4, 128, 128, device="cuda") for _ in range(3))
args = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": x - cond, "uncond": x - uncond, "input": x, "cond_scale": 7.0,
        "sigma": torch.full((b,), 7.0, device="cuda"), "model": FakeModel(), "model_options": {"transformer_options": {"sample_sigmas": SAMPLE_SIGMAS["karras12"]}}}
fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=[[3.0, 2.0, 1.0], "fill"] and [[3.0, 2.0, 1.0], [3.0, 2.0, 1.0], 2.0, 1.5, 1.0]), high_precision_mode=hp))
for _ in range(30):
    fn(args)
torch.cuda.synchronize()
