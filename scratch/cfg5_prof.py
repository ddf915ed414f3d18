"""cfg5 sampler steps only (128 x 16 x 128 x 128 shard, scheduled power + Perlin + Brownian chain, DPM++ SDE with momentum) for a kernel trace."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise"); sonar = importlib.import_module("comfyui_sonar_amd.py.sonar")
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
H = W = 128
xf = torch.randn(128, 16, H, W, device="cuda") * 10.0
inner = nz.CustomNoiseChain()
inner.add(pn.PowerNoiseItem(0.5, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0,
                            channel_correlation="1"))
inner.add(nz.CustomNoiseItem(0.3, noise_type="perlin"))
inner.add(nz.CustomNoiseItem(0.2, noise_type="brownian"))
fallback = nz.CustomNoiseChain(); fallback.add(nz.CustomNoiseItem(1.0, noise_type="gaussian"))
chain = nz.CustomNoiseChain()
chain.add(nz.ScheduledNoise(1.0, noise=inner, start_sigma=20.0, end_sigma=0.0, normalize=None, fallback_noise=fallback))
sigmas = torch.cat([torch.linspace(14.6, 0.5, 11), torch.zeros(1)])
ns5 = chain.make_noise_sampler(xf, 0.5, 14.6, seed=3, cpu=False, normalized=True)
run = lambda: sonar.SonarDPMPPSDE.sampler(lambda t, sigma, **_k: hl.mul_scalar(t, 0.5), xf, sigmas[:6], {"seed": 3}, None, True, None, dict(momentum=0.95), 1.0, 1.0, ns5)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): run()
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 20 * 1e3)
