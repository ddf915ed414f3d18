"""What a 134 MB tensor costs to write / read+write on this box, beside the generators that write one: torch fill and copy, the library's
raw Gaussian / uniform fills, Perlin's statistics pass, final pass and one-call forms (batch 512 SDXL)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
dev = "cuda"
x = torch.zeros((512, 4, 128, 128), device=dev); y = torch.empty_like(x)
sig = (torch.tensor(14.6), torch.tensor(10.0))
def t(fn, n=50, w=10):
    return bench.event_us(fn, n, w)
print(f"torch fill_ 134 MB            {t(lambda: x.fill_(1.5)):7.1f} us")
print(f"torch copy_ 134 -> 134 MB     {t(lambda: y.copy_(x)):7.1f} us")
print(f"torch x.mul_(2) (r+w 268 MB)  {t(lambda: x.mul_(1.0001)):7.1f} us")
ctr = [0]
def nxt():
    ctr[0] += 1
    return ctr[0]
print(f"philox_normal raw             {t(lambda: hl.philox_normal(tuple(x.shape), dev, 1, nxt())):7.1f} us")
print(f"philox_uniform raw            {t(lambda: hl.philox_uniform(tuple(x.shape), dev, 1, nxt())):7.1f} us")
for name in ("gaussian", "uniform", "perlin"):
    for norm in (False, True):
        ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=norm)
        old = hl.PLANS_ENABLED
        print(f"{name:9s} normalized={norm!s:5}   {t(lambda: ns(*sig)):7.1f} us per call")
