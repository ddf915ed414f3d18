"""cfg4's configured rule (wcfg_lowpass_kernel, db4 / level 5 / symmetric, 256 SDXL latents): us per launch, fp64 and fp32, and a check of the
result against the first library named (SONAR_HIP_LIB picks the build).  Usage: python scratch/lowpass_time.py [check.pt]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
wf = importlib.import_module("comfyui_sonar_amd.py.wavelet_functions")
torch.manual_seed(0)
cond, uncond, xin = (torch.randn(256, 4, 128, 128, device="cuda") for _ in range(3))
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
cfg_fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=3.0), high_precision_mode=True))
w = cfg_fn.rules[0].make_wavelet()
ref = sys.argv[1] if len(sys.argv) > 1 else None
saved = torch.load(ref) if ref and os.path.exists(ref) else {}
for hp in (True, False):
    kw = dict(levels=5, dec_lo=w.dec_lo, rec_lo=w.rec_lo, mode="symmetric", inv_mode="symmetric", g=[3.0, 0.0, 0.0, 0.0, 0.0, 2.0], ku=1.0, kt=1.0,
              subtract_from_x=True, high_precision=hp)
    out = hl.wcfg_lowpass(cond, uncond, xin, **kw)
    times = sorted(bench.event_us(lambda: hl.wcfg_lowpass(cond, uncond, xin, **kw), 20, 5) for _ in range(5))
    key = "fp64" if hp else "fp32"
    same = ""
    if key in saved:
        same = f"  equal to reference build: {torch.equal(saved[key].cuda(), out[:8])}  max diff {float((saved[key].cuda() - out[:8]).abs().max()):.2e}"
    else:
        saved[key] = out[:8].cpu()
    print(f"{os.environ.get('SONAR_HIP_LIB', 'product'):40s} {key}: median {times[2]:7.1f} us  min {times[0]:7.1f}{same}", flush=True)
if ref and not os.path.exists(ref):
    torch.save(saved, ref)
