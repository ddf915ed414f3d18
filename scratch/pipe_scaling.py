"""The pipelined power-law kernel's launch time against planes per CU (batch 128 .. 1536 SDXL latents): the slope is a steady-state plane,
the intercept everything a launch pays once (dispatch, pipeline fill and drain, the look-ahead statistics)."""
import ctypes as C, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("SONAR_HIP_LIB") or os.path.join(ROOT, "comfyui-sonar_amd", "libsonar_hip.so"))
lib.sonar_power_irfft2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
lib.sonar_power_noise_ahead_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
lib.sonar_power_noise_ahead_ok.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int]
dev = torch.device("cuda")
H = W = 128
filt = (torch.rand(H, W // 2 + 1, device=dev) + 0.5).contiguous()
stream = torch.cuda.current_stream().cuda_stream
rows = []
for batch in (128, 256, 384, 512, 768, 1024, 1536):
    planes = batch * 4
    out = torch.empty(planes, H, W, device=dev)
    ws = [torch.zeros(2048, dtype=torch.float64, device=dev) for _ in range(2)]
    res = {}
    for mode in ("final", "ahead"):
        if mode == "ahead" and not lib.sonar_power_noise_ahead_ok(planes, H, W, 4):
            res[mode] = float("nan")
            continue
        k = [0]
        def launch():
            if mode == "ahead":
                i = k[0]; k[0] += 1
                assert lib.sonar_power_noise_ahead_f32(filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2 + i, 0, 4, 1.0, 2.5, ws[i & 1].data_ptr(), int(i > 0), 3 + i, ws[(i + 1) & 1].data_ptr(), stream) == 0
            else:
                assert lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, None, stream) == 0
        for _ in range(100): launch()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(100): launch()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        res[mode] = statistics.median(ts)
    rows.append((batch, planes // 256, res["final"], res["ahead"]))
    print(f"batch {batch:5d} ({planes // 256:2d} planes per CU): final pass alone {res['final']:7.2f} us   look-ahead call {res['ahead']:7.2f} us", flush=True)
import numpy as np
x = np.array([r[1] for r in rows if r[1] >= 4], float); y = np.array([r[2] for r in rows if r[1] >= 4], float)
a, b = np.polyfit(x, y, 1)
print(f"final pass alone: {a:.2f} us per plane and CU + {b:.2f} us per launch (least squares over >= 4 planes per CU)")
