import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
x = torch.zeros((512, 4, 128, 128), device="cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
ctr = [0]
def nxt():
    ctr[0] += 1
    return ctr[0]
print(os.environ.get("SONAR_TILE_GRID_CAP"), end=": ")
print(f"normal raw {bench.event_us(lambda: hl.philox_normal(tuple(x.shape), 'cuda', 1, nxt()), 50, 10):.1f}", end="  ")
print(f"uniform raw {bench.event_us(lambda: hl.philox_uniform(tuple(x.shape), 'cuda', 1, nxt()), 50, 10):.1f}", end="  ")
ns = nz.get_noise_sampler("perlin", x, 0.03, 14.6, seed=None, cpu=False, normalized=False)
print(f"perlin raw {bench.event_us(lambda: ns(*sig), 50, 10):.1f}")
