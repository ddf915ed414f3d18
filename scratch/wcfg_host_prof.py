"""cProfile of the WaveletCFG call (cfg4 size, placeholder rule): where the host time between the sigma read and the launch goes."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
b = 256
cond, uncond, x = (torch.randn(b, 4, 128, 128, device="cuda") for _ in range(3))
args = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": x - cond, "uncond": x - uncond, "input": x, "cond_scale": 7.0,
        "sigma": torch.full((b,), 7.0, device="cuda"), "model": FakeModel(), "model_options": {"transformer_options": {"sample_sigmas": SAMPLE_SIGMAS["karras12"]}}}
fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=3.0)))
for _ in range(10): fn(args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): fn(args)
torch.cuda.synchronize()
print("per call us", (time.perf_counter() - t0) / 200 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fn(args)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
