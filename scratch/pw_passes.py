"""Times sonar_power_noise_f32 (statistics pass + final pass) and the bare final pass (sonar_power_irfft2_f32, generate mode) for every
variant library in scratch/bin/pwvar/ (see pw_build_variants.sh) on 512 SDXL latents; HIP-event timed over 100 launches.
    python scratch/pw_passes.py [name ...]"""
import ctypes as C, glob, os, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:] or sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(ROOT, "scratch/bin/pwvar/lib_*.so")))
planes, H, W = 2048, 128, 128
dev = torch.device("cuda")
filt = (torch.rand(H, W // 2 + 1, device=dev) + 0.5).contiguous()
out = torch.empty(planes, H, W, device=dev)
partials = torch.empty(2048, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, n=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name in names:
    lib = C.CDLL(os.path.join(ROOT, f"scratch/bin/pwvar/lib_{name}.so"))
    lib.sonar_power_noise_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_float,
                                          C.c_float, C.c_void_p, C.c_void_p]
    lib.sonar_power_irfft2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64,
                                           C.c_int, C.c_void_p, C.c_void_p]
    pair = timed(lambda: lib.sonar_power_noise_f32(filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, 1.0, 2.5, partials.data_ptr(), stream))
    final = timed(lambda: lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, None, stream))
    lib.sonar_spectral_filter_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    xin = torch.randn(planes, H, W, device=dev)
    sf = timed(lambda: lib.sonar_spectral_filter_f32(xin.data_ptr(), filt.data_ptr(), out.data_ptr(), planes, H, W, None, stream), 50)
    print(f"{name:24s} spectral filter {sf:7.1f} us", flush=True)
    print(f"{name:24s} pair {pair:7.1f} us   final pass alone {final:7.1f} us   std {out.std().item():.4f}", flush=True)
