"""WaveletCFG call (cfg4 size, placeholder rule), fp32 and fp64: sigma read, host logic after it, kernel -- where the end-to-end time goes."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
b = 256
cond, uncond, x = (torch.randn(b, 4, 128, 128, device="cuda") for _ in range(3))
sig = torch.full((b,), 7.0, device="cuda")
args = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": x - cond, "uncond": x - uncond, "input": x, "cond_scale": 7.0,
        "sigma": sig, "model": FakeModel(), "model_options": {"transformer_options": {"sample_sigmas": SAMPLE_SIGMAS["karras12"]}}}
for hp in (False, True):
    fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=3.0), high_precision_mode=hp))
    for _ in range(300): fn(args)
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for _ in range(n): fn(args)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n): hl.max_to_host(sig)
    ms = (time.perf_counter() - t0) / n * 1e6
    # host part alone: the calls queue behind each other when nothing waits for the device
    real = hl.max_to_host_begin, hl.max_to_host_end
    hl.max_to_host_begin, hl.max_to_host_end = (lambda s: None), (lambda t: 7.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn(args)
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    hl.max_to_host_begin, hl.max_to_host_end = real
    print(f"high_precision={hp}: per call {tot:.1f} us; sigma read alone {ms:.1f} us; host logic without the read {host:.1f} us")
    if not hp:
        hl.max_to_host_begin, hl.max_to_host_end = (lambda s: None), (lambda t: 7.0)
        pr = cProfile.Profile(); pr.enable()
        for _ in range(n): fn(args)
        pr.disable(); torch.cuda.synchronize()
        hl.max_to_host_begin, hl.max_to_host_end = real
        pstats.Stats(pr).sort_stats("tottime").print_stats(30)
