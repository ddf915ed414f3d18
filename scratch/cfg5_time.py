"""cfg5: Flux shard 128 x 16 x 128 x 128, Scheduled(power + perlin + brownian) chain, SonarDPMPPSDE, 10 steps, fake model 0.5 x."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
N = importlib.import_module("comfyui_sonar_amd.py.noise"); S = importlib.import_module("comfyui_sonar_amd.py.sonar")
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x0 = torch.randn(B, 16, 128, 128, device="cuda") * 14.6
sigmas = torch.cat((torch.linspace(14.6, 0.03, 10), torch.zeros(1)))
inner = N.CustomNoiseChain()
inner.add(pn.PowerNoiseItem(0.5, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                            common_mode=0.0, channel_correlation="1"))
inner.add(N.CustomNoiseItem(0.3, noise_type="perlin"))
inner.add(N.CustomNoiseItem(0.2, noise_type="brownian"))
gauss = N.CustomNoiseChain(); gauss.add(N.CustomNoiseItem(1.0, noise_type="gaussian"))
chain = N.CustomNoiseChain()
chain.add(N.ScheduledNoise(1.0, noise=inner, start_sigma=100.0, end_sigma=0.0, normalize=None, fallback_noise=gauss))
model = lambda x, s, **kw: x * 0.5
ns = chain.make_noise_sampler(x0, 0.03, 14.6, seed=3, cpu=False, normalized=True)
def run():
    return S.SonarDPMPPSDE.sampler(model, x0.clone(), sigmas, {"seed": 3}, None, True, None, dict(momentum=0.95), 1.0, 1.0, ns)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); out = run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"cfg5 B={B}: {dt*1e3:.2f} ms for 10 steps -> {dt*100:.2f} ms/step, {B*10/dt/1e3:.1f} k latent-steps/s; finite {bool(torch.isfinite(out).all())}")
for name, f in (("noise call", lambda: ns(torch.tensor(9.0), torch.tensor(7.0))),):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); print(name, f"{(time.perf_counter()-t0)/5*1e3:.2f} ms")
