"""cfg3 chain (Perlin + pyramid, normalised) at batch 64: wall time per call against GPU time per call, and the host profile."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
x = torch.zeros((64, 4, 128, 128), device="cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
chain = nz.CustomNoiseChain()
chain.add(nz.CustomNoiseItem(0.5, noise_type="perlin"))
chain.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
ns = chain.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
for _ in range(300): ns(*sig)
torch.cuda.synchronize()
t0 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(500): ns(*sig)
host = (time.perf_counter() - t0) / 500 * 1e6
e1.record(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 500 * 1e6
print(f"host issue time {host:.1f} us per call, wall {wall:.1f} us, GPU span {e0.elapsed_time(e1) * 2:.1f} us per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(500): ns(*sig)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22); st.sort_stats("cumulative").print_stats(60)
