"""cfg3 chain / pyramid / Perlin at batch 64 and the power-law call at batch 8: host time per call split into time inside the C ABI
(ctypes conversion + launch) and Python around it."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; lib = hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
acc = {}
def wrap(name):
    fn = getattr(lib, name)
    def w(*a):
        t = time.perf_counter()
        r = fn(*a)
        d = acc.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += time.perf_counter() - t
        return r
    setattr(lib, name, w)
for name in hl.SIGNATURES:
    if name not in ("sonar_last_error",): wrap(name)
sig = (torch.tensor(14.6), torch.tensor(10.0))
def run(tag, ns, n=2000):
    for _ in range(300): ns(*sig)
    torch.cuda.synchronize(); acc.clear()
    t0 = time.perf_counter()
    for _ in range(n): ns(*sig)
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e6
    c = sum(v[1] for v in acc.values()) / n * 1e6
    parts = ", ".join(f"{k[6:]} x{v[0] / n:.0f} {v[1] / v[0] * 1e6:.1f}" for k, v in acc.items())
    print(f"{tag}: host {host:.1f} us (wall {wall:.1f}), inside the C ABI {c:.1f} [{parts}]")
x = torch.zeros((64, 4, 128, 128), device="cuda")
for name in ("perlin", "pyramid", "gaussian", "onef_pinkish"):
    run(name + " b64", nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True))
chain = nz.CustomNoiseChain()
chain.add(nz.CustomNoiseItem(0.5, noise_type="perlin")); chain.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
run("chain b64", chain.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=True))
x1 = torch.zeros((1, 4, 128, 128), device="cuda")
for name in ("gaussian", "perlin", "pyramid"):
    run(name + " b1", nz.get_noise_sampler(name, x1, 0.03, 14.6, seed=None, cpu=False, normalized=True))
# every registry type at 4 latents (what a ComfyUI run asks for): host-bound outliers show up as host >> inside the C ABI
x4 = torch.zeros((4, 4, 128, 128), device="cuda")
for t in nz.NoiseType:
    try:
        ns = nz.get_noise_sampler(t.name.lower(), x4, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        run(t.name.lower() + " b4", ns, n=300)
    except Exception as e:
        print(t.name.lower(), "b4: FAILED", type(e).__name__, str(e)[:100])
