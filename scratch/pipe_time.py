"""Launch time of the generate pass (sonar_power_irfft2_f32, z = NULL, 2048 planes of 128 x 128) for profiling variants of the library:
    python scratch/pipe_time.py scratch/bin/pwvar/lib_a.so scratch/bin/pwvar/lib_b.so ...   (default: the product library)"""
import ctypes as C, os, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["comfyui-sonar_amd/libsonar_hip.so"]
planes, H, W = 2048, 128, 128
dev = torch.device("cuda")
filt = (torch.rand(H, W // 2 + 1, device=dev) + 0.5).contiguous()
out = torch.empty(planes, H, W, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for path in libs:
    lib = C.CDLL(os.path.join(ROOT, path))
    lib.sonar_power_irfft2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    lib.sonar_power_noise_ahead_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
    ws = [torch.zeros(2048, dtype=torch.float64, device=dev) for _ in range(2)]
    step = [0]
    def launch():
        if os.environ.get("MODE") == "ahead":  # steady state of a sampler: every call finds its statistics and leaves the next call's
            k = step[0]
            step[0] += 1
            assert lib.sonar_power_noise_ahead_f32(filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2 + k, 0, 4, 1.0, 2.5, ws[k & 1].data_ptr(), int(k > 0), 3 + k, ws[(k + 1) & 1].data_ptr(), stream) == 0
        else:
            assert lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, None, stream) == 0
    for _ in range(500):
        launch()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            launch()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10)
    print(f"{os.path.basename(path):28s} {best:6.1f} us per launch   (std of output {out.std().item():.4f})", flush=True)
