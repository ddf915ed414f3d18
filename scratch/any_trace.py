"""Phase stamps of the general-size plane kernel on an SDXL bucket (104 x 152; trace build: see below): draw, columns, rows, store per plane.
    (cd comfyui-sonar_amd/csrc && hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DSONAR_ANY_TRACE -c power_buckets_a.hip -o /tmp/pba.o)
    hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/bin/pwvar/lib_anytrace.so /tmp/pba.o $(ls comfyui-sonar_amd/build/*.o | grep -v power_buckets_a.o)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SONAR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scratch/bin/pwvar/lib_anytrace.so"))
import numpy as np, torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
raw = C.CDLL(os.environ["SONAR_HIP_LIB"])
H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "104x152").split("x"))
planes = 512 * 4 * 128 * 128 // (H * W) // 4 * 4
filt = (torch.rand(H, W // 2 + 1, device="cuda") + 0.5).contiguous()
for i in range(30): hl.power_noise(filt, (planes // 4, 4, H, W), seed=7, stream_id=100 + i, plane_offset=0, factor=1.0)
torch.cuda.synchronize()
buf = np.zeros(512 * 8 * 8, dtype=np.uint64)
assert raw.sonar_debug_any_trace_a(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(512, 8, 8).astype(np.int64)
n = int((t[:, :, 5] > 0).sum(axis=1).min())
names = ["draw (fill)", "barrier", "columns (line_dft)", "rows (c2r_rows)", "normalise + store"]
print(f"{H} x {W}: {planes} planes, {n} planes per workgroup traced; ticks per plane (2000 per us?), planes 1..{n - 1}")
for k in range(5):
    print(f"   {names[k]:22s} {np.mean(t[:, 1:n, k + 1] - t[:, 1:n, k]):9.0f}")
print(f"   {'to the next plane':22s} {np.mean(t[:, 2:n, 0] - t[:, 1:n - 1, 5]):9.0f}")
print(f"   {'plane':22s} {np.mean(t[:, 2:n, 0] - t[:, 1:n - 1, 0]):9.0f}")
