"""Direct DFT passes, per pass, for P planes of 135 x 240 (event-timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; lib = hl.load()
H, W = 135, 240; K = W // 2 + 1
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for P in (16, 256, 336, 512):
    x = torch.randn(P, H, W, device="cuda"); a = torch.empty(P, H, K, dtype=torch.complex64, device="cuda"); b = torch.empty_like(a)
    filt = torch.rand(H, K, device="cuda"); out = torch.empty_like(x); part = hl.new_partials("cuda")
    r = [t(lambda: lib.sonar_dft_rows_r2c_f32(x.data_ptr(), a.data_ptr(), P * H, W, None)),
         t(lambda: lib.sonar_dft_cols_f32(a.data_ptr(), None, b.data_ptr(), P, H, K, 0, None)),
         t(lambda: lib.sonar_dft_cols_f32(b.data_ptr(), filt.data_ptr(), a.data_ptr(), P, H, K, 1, None)),
         t(lambda: lib.sonar_dft_rows_c2r_f32(a.data_ptr(), out.data_ptr(), P * H, W, 1.0, part.data_ptr(), None)),
         t(lambda: hl.philox_normal((P, H, W), "cuda", 1, 2, 0)), t(lambda: hl.scale_noise_(out, 1.0, True, part))]
    print(P, "planes: r2c %.0f  cols fwd %.0f  cols inv+filter %.0f  c2r %.0f  white %.0f  normalise %.0f us" % tuple(r))
