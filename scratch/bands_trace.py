"""Phase stamps of wcfg_bands_kernel (trace build: scratch/bandsv.sh btrace "-DSONAR_BANDS_TRACE"): thread 0's cycle stamps of every
workgroup's first plane, for the tile route's deeper-levels call and the single-launch route, fp32 / fp64."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SONAR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scratch/bin/pwvar/lib_btrace.so"))
import numpy as np, torch, sonar_pkg
from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel
pkg = sonar_pkg.load(); hl = pkg.hip_lib; lib = hl.load()
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
b = 256
cond, uncond, x = (torch.randn(b, 4, 128, 128, device="cuda") for _ in range(3))
args = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": x - cond, "uncond": x - uncond, "input": x, "cond_scale": 7.0,
        "sigma": torch.full((b,), 7.0, device="cuda"), "model": FakeModel(), "model_options": {"transformer_options": {"sample_sigmas": SAMPLE_SIGMAS["karras12"]}}}
raw = C.CDLL(os.environ["SONAR_HIP_LIB"])
names = {0: "tables / previous plane", 1: "level 1 down", 2: "level 2 down", 3: "level 3 down", 4: "level 4 down", 5: "level 5 down", 8: "top",
         9: "up to level 4", 10: "up to level 3", 11: "up to level 2", 12: "up to level 1", 16: "level 1 up + stores"}
wc.WaveletCFG._lowpass_launch = classmethod(lambda cls, **_k: None)
for hp in (False, True):
    for single in (False, True):
        fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=[[3.0, 2.5, 2.0]] * 5), high_precision_mode=hp))
        wc.WaveletCFG.single_launch_bands = single
        for _ in range(5): fn(args)
        torch.cuda.synchronize()
        buf = np.zeros(512 * 32, dtype=np.uint64)
        assert raw.sonar_debug_bands_trace(buf.ctypes.data_as(C.c_void_p)) == 0
        t = buf.reshape(512, 32).astype(np.int64)
        t = t[t[:, 16] > t[:, 0]]
        order = [k for k in sorted(names) if ((t[:, k] >= t[:, 0]) & (t[:, k] <= t[:, 16])).all()]  # (slots of levels this call does not have keep old stamps)
        print(f"{'fp64' if hp else 'fp32'} {'single launch' if single else 'deeper levels of the tile route'}: {len(t)} workgroups, ticks (2000 per us?)")
        for a_, b_ in zip(order[:-1], order[1:]):
            print(f"   {names[b_]:24s} {np.mean(t[:, b_] - t[:, a_]):9.0f}")
        print(f"   {'whole plane':24s} {np.mean(t[:, 16] - t[:, 0]):9.0f}")
        print(f"   last stage by phase, summed over its tiles: stage the tile's rows {np.mean(t[:, 20]):.0f}, along H + lowW {np.mean(t[:, 21]):.0f}, along W + tail + stores {np.mean(t[:, 22]):.0f}")
wc.WaveletCFG.single_launch_bands = None
