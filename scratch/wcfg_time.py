"""WaveletCFG (cfg4: 256 x 4 x 128 x 128, placeholder rule) end to end and kernel only, low-pass path vs band path, fp64 / fp32."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
b = 256
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

real_low = wc.WaveletCFG._lowpass_launch
for hp in (True, False):
    fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=3.0), high_precision_mode=hp))
    real_pr = wc._reconstructs
    for path in ("lowpass", "diff", "pair"):
        wc.WaveletCFG._lowpass_launch = real_low if path == "lowpass" else classmethod(lambda cls, **_k: None)
        wc._reconstructs = (lambda w: False) if path == "pair" else real_pr
        wall, ev = timed(lambda: fn(args))
        print(f"{'fp64' if hp else 'fp32'} {path:8s} wall {wall:7.1f} us  events {ev:7.1f} us  -> {16 * 4 * 128 * 128 * b / ev / 1e3:7.1f} GB/s at 16N", flush=True)
    wc.WaveletCFG._lowpass_launch = real_low
    wc._reconstructs = real_pr
    w = fn.rules[0].make_wavelet()
    g = [3.0, 0, 0, 0, 0, 2.0]
    wall, ev = timed(lambda: hl.wcfg_lowpass(cond, uncond, x, levels=5, dec_lo=w.dec_lo, rec_lo=w.rec_lo, mode="symmetric", inv_mode="symmetric", g=g, ku=1.0, kt=1.0,
                                             subtract_from_x=True, high_precision=hp))
    print(f"{'fp64' if hp else 'fp32'} kernel   wall {wall:7.1f} us  events {ev:7.1f} us", flush=True)

