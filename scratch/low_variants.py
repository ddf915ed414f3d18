"""Times sonar_wcfg_lowpass_f64 / _f32 (256 x 4 x 128 x 128, db4 level 5 symmetric) for the variant libraries in scratch/bin/dwtvar/."""
import ctypes as C, glob, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
taps = json.load(open(os.path.join(ROOT, "comfyui-sonar_amd/wavelet_taps.json")))["wavelets"]["db4"]
names = sys.argv[1:] or sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(ROOT, "scratch/bin/dwtvar/lib_*.so")))
b = 256
cond, uncond, x = (torch.randn(b, 4, 128, 128, device="cuda") for _ in range(3))
out = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
D8 = C.c_double * 8
dlo, rlo = D8(*taps["dec_lo"]), D8(*taps["rec_lo"])
g = (C.c_double * 6)(3.0, 0, 0, 0, 0, 2.0)
def timed(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    for name in names:
        lib = C.CDLL(os.path.join(ROOT, f"scratch/bin/dwtvar/lib_{name}.so"))
        res = []
        for fn in (lib.sonar_wcfg_lowpass_f64, lib.sonar_wcfg_lowpass_f32):
            fn.argtypes = [C.c_void_p] * 4 + [C.c_int64] * 3 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p]
            res.append(timed(lambda: fn(cond.data_ptr(), uncond.data_ptr(), x.data_ptr(), out.data_ptr(), b * 4, 128, 128, 5, dlo, rlo, 8, 1, 1, g, 1.0, 1.0, 1, stream)))
        print(f"{name:12s} fp64 {res[0]:7.1f} us   fp32 {res[1]:7.1f} us", flush=True)
