"""Small-batch steps for a kernel trace: Perlin, pyramid, their chain and the power-law call at 4 and 64 SDXL latents, through prepared plans."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
REPS = int(os.environ.get("PROF_REPS", "200"))
WHICH = os.environ.get("PROF_WHICH", "perlin,pyramid,chain,power").split(",")
for B in [int(v) for v in os.environ.get("PROF_BATCH", "64").split(",")]:
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    todo = {}
    if "perlin" in WHICH: todo["perlin"] = nz.get_noise_sampler("perlin", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    if "pyramid" in WHICH: todo["pyramid"] = nz.get_noise_sampler("pyramid", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    if "chain" in WHICH:
        c = nz.CustomNoiseChain(); c.add(nz.CustomNoiseItem(0.5, noise_type="perlin")); c.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
        todo["chain"] = c.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    if "power" in WHICH:
        todo["power"] = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                                          common_mode=0.0, channel_correlation="1,1,1,1,1,1").make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
    for name, ns in todo.items():
        for _ in range(REPS): ns(*sig)
        torch.cuda.synchronize()
