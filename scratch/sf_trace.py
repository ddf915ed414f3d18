"""Phase timeline of the spectral filter at 128 x 128 (trace build: scratch/pwv.sh sftrace "-DSONAR_PW_TRACE"): wave 0's cycle stamps
behind every workgroup barrier of a plane, averaged over workgroups and the steady-state planes."""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "scratch/bin/pwvar/lib_sftrace.so"))
planes, H, W = 2048, 128, 128
dev = torch.device("cuda")
filt = (torch.rand(H, W // 2 + 1, device=dev) + 0.5).contiguous()
x = torch.randn(planes, H, W, device=dev)
out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
lib.sonar_spectral_filter_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
def launch():
    assert lib.sonar_spectral_filter_f32(x.data_ptr(), filt.data_ptr(), out.data_ptr(), planes, H, W, None, st) == 0
for _ in range(200): launch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): launch()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
buf = np.zeros(1024 * 8 * 12, dtype=np.uint64)
lib.sonar_debug_pw_trace.argtypes = [C.c_void_p]
assert lib.sonar_debug_pw_trace(buf.ctypes.data) == 0
t = buf.reshape(1024, 8, 12).astype(np.int64)[:512]
TICK = 1.0 / float(os.environ.get("SONAR_TICKS_PER_US", "2000"))  # the shader clock counter
slots = [0, 1, 2, 3, 7, 8, 9, 10, 4, 5, 6, 11]
names = ["rows b' (global loads -> LDS)", "rows a' + split", "cols b'", "cols a': loads + forward radix 16", "  lane 0 leaves P, x filter, barrier", "  lane 0 builds Q", "  inverse radix 16 + twiddles", "  stores + barrier", "cols b", "rows a", "rows b + stores"]
nplanes = int((t[:, :, 0] > 0).sum(axis=1).min())
print(f"kernel {us:.1f} us per launch; {nplanes} planes per workgroup")
tot = 0.0
for k in range(len(slots) - 1):
    d = (t[:, 1:nplanes - 1, slots[k + 1]] - t[:, 1:nplanes - 1, slots[k]]) * TICK
    tot += d.mean()
    print(f"  {names[k]:48s} {d.mean():6.2f} us")
gap = (t[:, 2:nplanes, 0] - t[:, 1:nplanes - 1, 11]) * TICK
print(f"  {'to the next plane (top barrier)':48s} {gap.mean():6.2f} us")
print(f"  per plane {tot + gap.mean():.2f} us; span of a workgroup {((t[:, nplanes - 1, 11] - t[:, 0, 0]) * TICK).mean():.1f} us")
