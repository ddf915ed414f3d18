import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
x = torch.zeros((4, 4, 128, 128), device="cuda")
orig = torch._C._storage_Use_Count
def dbg(cdata):
    n = orig(cdata)
    print("   use count", n)
    return n
torch._C._storage_Use_Count = dbg
ns = nz.get_noise_sampler("perlin", x, 0.03, 14.6, seed=5, cpu=False, normalized=True)
real_on_alloc = hl._Recorder.on_alloc
def on_alloc(self, t):
    print("   alloc", tuple(t.shape), t.dtype, hex(t.data_ptr()))
    return real_on_alloc(self, t)
hl._Recorder.on_alloc = on_alloc
for i in range(4):
    print("call", i); ns(*sig)
print(ns._planned.reason)
t = torch.empty(8, device="cuda"); print("lone cuda tensor", orig(t.untyped_storage()._cdata))
