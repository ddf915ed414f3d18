"""WaveletCFG rules that need the bands (256 x 4 x 128 x 128, db4, level 5): the LDS-resident band kernel (sonar_wcfg_bands_*) against the
three tile kernels (sonar_wcfg_fused_*), end to end and kernel only, fp64 / fp32."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
b = int(os.environ.get("WCFG_BATCH", "256"))
cond, uncond, x = (torch.randn(b, 4, 128, 128, device="cuda") for _ in range(3))
args = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": x - cond, "uncond": x - uncond, "input": x, "cond_scale": 7.0,
        "sigma": torch.full((b,), 7.0, device="cuda"), "model": FakeModel(), "model_options": {"transformer_options": {"sample_sigmas": SAMPLE_SIGMAS["karras12"]}}}

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3

rules = {"bands_difference": dict(difference=dict(yl_scale=5.0, yh_scales=[[3.0, 2.5, 2.0]] * 5)),
         "bands_difference_hv": dict(difference=dict(yl_scale=5.0, yh_scales=[[3.0, 2.0, 2.0]] * 5)),
         "bands_pair": dict(cond=dict(yl_scale=1.1, yh_scales=1.0), uncond=dict(yl_scale=1.0, yh_scales=0.9), difference=dict(yl_scale=5.0, yh_scales=3.0)),
         "placeholder": dict(difference=dict(yl_scale=5.0, yh_scales=3.0))}
real_bands, real_low = wc.WaveletCFG.wavelet_cfg_bands, wc.WaveletCFG._lowpass_launch
wc.WaveletCFG._lowpass_launch = classmethod(lambda cls, **_k: None)
for tag, params in rules.items():
    for hp in (True, False):
        fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(**params, high_precision_mode=hp))
        row = []
        outs = []
        for path in ("lds", "tiles", "tiles, fp64 detail storage"):
            wc.WaveletCFG.single_launch_bands = path == "lds"
            hl.load().sonar_wcfg_hi_storage(0 if "fp64 detail" in path else 1)
            if "fp64 detail" in path and not hp:
                continue
            outs.append(fn(args).clone())
            wall, ev = timed(lambda: fn(args))
            row.append(f"{path} {ev:6.1f} us (wall {wall:6.1f})")
        wc.WaveletCFG.single_launch_bands = None
        hl.load().sonar_wcfg_hi_storage(1)
        err = (outs[0] - outs[1]).abs().max().item() / outs[1].abs().max().item()
        print(f"{tag:22s} {'fp64' if hp else 'fp32'}: " + " | ".join(row) + f" | max rel diff {err:.1e}", flush=True)
