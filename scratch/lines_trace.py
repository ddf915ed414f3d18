"""Phase stamps of lines_c2r_kernel (the row kernel of planes beyond LDS) on 256 x 256 planes: per batch of rows, load / transform / store.
Trace build: (cd comfyui-sonar_amd/csrc && hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DSONAR_LINES_TRACE -c power_fft.hip -o /tmp/pf.o)
             hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/bin/pwvar/lib_linestrace.so /tmp/pf.o $(ls comfyui-sonar_amd/build/*.o | grep -v /power_fft.o)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SONAR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scratch/bin/pwvar/lib_linestrace.so"))
import numpy as np, torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
raw = C.CDLL(os.environ["SONAR_HIP_LIB"])
shape = (128, 4, 256, 256)
filt = torch.rand(256, 129, device="cuda") + 0.5
for i in range(20): hl.power_noise(filt, shape, seed=1, stream_id=2 + i, plane_offset=0, factor=1.0)
torch.cuda.synchronize()
buf = np.zeros(512 * 4 * 8, dtype=np.uint64)
assert raw.sonar_debug_lines_trace(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(512, 4, 8).astype(np.int64)
n = int((t[:, :, 3] > 0).sum(axis=1).min())
print(f"{n} batches per workgroup traced; ticks per batch (2000 per us?)")
for k, name in enumerate(["load the rows into LDS", "c2r_rows (pre-twiddle + two passes)", "normalise + store"]):
    print(f"   {name:36s} {np.mean(t[:, :n, k + 1] - t[:, :n, k]):9.0f}")
if n > 1: print(f"   {'to the next batch':36s} {np.mean(t[:, 1:n, 0] - t[:, :n - 1, 3]):9.0f}")
