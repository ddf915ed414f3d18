"""cfg3 at the configured batch (64 SDXL latents) and at 512: Perlin, pyramid, and the Perlin + pyramid chain, normalised, generate mode;
us per call (HIP events over 200 calls) -- SONAR_HIP_LIB selects a profiling build."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
def ev(fn, n=200, w=100):
    for _ in range(w): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
row = [os.path.basename(os.environ.get("SONAR_HIP_LIB", "product"))]
for B in (64, 512):
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    for name in ("perlin", "pyramid"):
        ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        row.append(f"{name} b{B} {ev(lambda: ns(*sig)):.1f}")
    chain = nz.CustomNoiseChain()
    chain.add(nz.CustomNoiseItem(0.5, noise_type="perlin"))
    chain.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
    ns3 = chain.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    row.append(f"chain b{B} {ev(lambda: ns3(*sig)):.1f}")
print(" | ".join(row))
