"""Phase timeline of power_irfft2_kernel<128,128,GEN,NORM>: s_memtime stamps of wave 0 of every workgroup (trace build
scratch/bin/pwvar/lib_trace.so = -DSONAR_PW_TRACE).  Prints the mean duration of each phase over all workgroups and planes, in
shader-clock cycles and microseconds (the clock is calibrated from the first and last stamps against the HIP-event time)."""
import ctypes as C, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "scratch/bin/pwvar/lib_trace.so"))
planes, H, W = 2048, 128, 128
dev = torch.device("cuda")
filt = (torch.rand(H, W // 2 + 1, device=dev) + 0.5).contiguous()
out = torch.empty(planes, H, W, device=dev)
stream = torch.cuda.current_stream().cuda_stream
lib.sonar_power_irfft2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
for _ in range(5):
    lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, None, stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, None, stream)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
buf = np.zeros(1024 * 8 * 12, dtype=np.uint64)
lib.sonar_debug_pw_trace.argtypes = [C.c_void_p]
assert lib.sonar_debug_pw_trace(buf.ctypes.data) == 0
t = buf.reshape(1024, 8, 12)[:512, :4].astype(np.int64)   # 512 workgroups x 4 planes x 12 stamps
span = t[..., 11].max() - t[..., 0].min()
print(f"kernel {us:.1f} us (one launch, events); stamps span {span} ticks -> {span / us:.0f} ticks/us")
names = ["fill (draw)", "wait fill barrier", "fix-up + barrier", "cols a", "wait barrier", "cols b", "wait barrier", "rows a reads+math", "wait mid barrier",
         "rows a writes", "wait barrier", "rows b + stores"]
d = np.diff(t, axis=-1)
tick_us = us / span
tot = 0.0
for i in range(11):
    m = d[..., i].mean()
    tot += m
    print(f"  {names[i]:22s} {m:8.0f} ticks  {m * tick_us:6.2f} us   (min {d[..., i].min()}, max {d[..., i].max()})")
gap = (t[:, 1:, 0] - t[:, :-1, 11]).mean()
print(f"  {'loop top barrier':22s} {gap:8.0f} ticks  {gap * tick_us:6.2f} us")
print(f"  per plane {tot + gap:.0f} ticks = {(tot + gap) * tick_us:.2f} us; x 4 planes = {(tot + gap) * 4 * tick_us:.1f} us")
start = t[:, 0, 0] - t[..., 0].min()
print(f"  first stamp of a workgroup after kernel start: mean {start.mean() * tick_us:.2f} us, max {start.max() * tick_us:.2f} us; blocks 256.. : {start[256:].mean() * tick_us:.2f} us")
