"""Phase timeline of power_pipe_kernel<128,128>: cycle stamps of every wave (trace build scratch/bin/pwvar/lib_trace.so =
-DSONAR_PW_TRACE).  A barrier opens when the last of the 16 waves arrives; per phase: its duration, and how long after the previous
barrier the waves of each team arrived (mean / last)."""
import ctypes as C, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "scratch/bin/pwvar/lib_trace.so"))
planes, H, W = 2048, 128, 128
dev = torch.device("cuda")
filt = (torch.rand(H, W // 2 + 1, device=dev) + 0.5).contiguous()
out = torch.empty(planes, H, W, device=dev)
stream = torch.cuda.current_stream().cuda_stream
lib.sonar_power_irfft2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
lib.sonar_power_noise_ahead_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
AHEAD = os.environ.get("SONAR_TRACE_AHEAD", "0") == "1"  # the normalised call with the look-ahead statistics in the idle corners
ws = [torch.zeros(2048, dtype=torch.float64, device=dev) for _ in range(2)]
def launch():
    if AHEAD:
        assert lib.sonar_power_noise_ahead_f32(filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, 1.0, 2.5, ws[0].data_ptr(), 0, 3, ws[1].data_ptr(), stream) == 0
    else:
        assert lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, None, stream) == 0
for _ in range(300):
    launch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    launch()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
NS = 4
buf = np.zeros(256 * 16 * 10 * NS, dtype=np.uint64)
lib.sonar_debug_pipe_trace.argtypes = [C.c_void_p]
assert lib.sonar_debug_pipe_trace(buf.ctypes.data) == 0
t = buf.reshape(256, 16, 10, NS).astype(np.int64)[:, :, :9]   # blocks x wave (0-7 transform, 8-15 draw) x iteration x stamp
spans = t[:, :, 8, NS - 1].max(axis=1) - t[:, :, 0, 0].min(axis=1)  # per workgroup (the counters of different XCDs have different origins)
TICKS_PER_US = float(os.environ.get("SONAR_TICKS_PER_US", "2000"))
tick_us = 1.0 / TICKS_PER_US
print(f"kernel {us:.1f} us per launch (events, 20 launches); a workgroup's stamps span {spans.mean():.0f} ticks = {spans.mean() * tick_us:.1f} us at {TICKS_PER_US:.0f} ticks/us")
names = ["cols b | draw c0 + edges", "rows a | draw c1 + packed col", "rows b + store | draw c2 + cols a"]
rel = t.max(axis=1)  # blocks x iteration x stamp: the barrier after stamp k opens when the last wave arrives
print("per iteration and phase: duration us (transform team arrival mean/last | drawing team arrival mean/last)")
tot = np.zeros(9)
for j in range(9):
    row = []
    for k in range(NS - 1):
        start = rel[:, j, k] if k > 0 else (rel[:, j - 1, NS - 1] if j > 0 else t[:, :, 0, 0].min(axis=1))
        dur = (rel[:, j, k + 1] - start).mean() * tick_us
        tot[j] += dur
        arr = (t[:, :, j, k + 1] - start[:, None]) * tick_us
        row.append(f"{dur:5.2f} ({arr[:, :8].mean():4.2f}/{arr[:, :8].max(axis=1).mean():4.2f} | {arr[:, 8:].mean():4.2f}/{arr[:, 8:].max(axis=1).mean():4.2f})")
    print(f"  j={j}: " + "  ".join(row) + f"   = {tot[j]:.2f}")
print(f"steady state (j = 1..7): {tot[1:8].mean():.2f} us per plane; prologue {tot[0]:.2f}, epilogue {tot[8]:.2f}; sum {tot.sum():.1f} us")
