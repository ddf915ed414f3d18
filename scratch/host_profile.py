"""Where the host's ~11 us per replayed call go (Perlin at 4 latents, a one-record plan): cProfile over 20 000 calls."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
x = torch.zeros((4, 4, 128, 128), device="cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for name in (sys.argv[1:] or ["perlin"]):
    ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    for _ in range(300): ns(*sig)
    torch.cuda.synchronize()
    n = 20000
    t0 = time.perf_counter()
    for i in range(n):
        ns(*sig)
        if i % 64 == 63: torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.2f} us per call (host, with a sync every 64 calls)")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        ns(*sig)
        if i % 64 == 63: torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime")
    st.print_stats(18)
