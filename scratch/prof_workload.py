"""Workload for the PMC passes (rounds 2-4): every kernel whose HBM traffic profiles/r02_traffic.json reports, a few launches each, at the
sizes bench.py times them, plus three calibration kernels with known byte counts (see scratch/calib.py)."""
import importlib, os, sys, types, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
hl.PLANS_ENABLED = False  # the ordinary launches (a plan's Perlin call is another kernel at batch 64: make_traffic_json.py splits by kernel name)
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sonar = importlib.import_module("comfyui_sonar_amd.py.sonar")
wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
dev = "cuda"
sig = (torch.tensor(14.6), torch.tensor(10.0))
REPS = int(os.environ.get("PROF_REPS", "4"))
x = torch.zeros((512, 4, 128, 128), device=dev)
x64 = torch.zeros((64, 4, 128, 128), device=dev)
# calibration: known bytes (134 217 728 B per tensor)
for i in range(REPS):
    t = hl.philox_normal(tuple(x.shape), dev, 1, i)
    p = hl.stats(t)
    hl.scale_noise_(t, 0.5, True, p)
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0,
                         channel_correlation="1,1,1,1,1,1")
ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
for _ in range(REPS): ns(*sig)
for name in ("perlin", "pyramid"):
    for xb in (x, x64):
        s = nz.get_noise_sampler(name, xb, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        for _ in range(REPS): s(*sig)
        torch.cuda.synchronize()
sb = sonar.SonarBase(sonar.SonarBase.get_config(None, {}))
den, xs = torch.randn_like(x), torch.randn_like(x)
sb.momentum_step(0, xs, den, torch.tensor(10.0), torch.tensor(8.0))
for _ in range(REPS): sb.momentum_step(1, xs, den, torch.tensor(8.0), torch.tensor(6.0))
filt = torch.rand(128, 65, device=dev) + 0.5
for _ in range(REPS): hl.spectral_filter(xs, filt)
# a 2048 px latent: 128 latents of 4 x 256 x 256 (33.5 M values) through the column-block route (workspace written and read once)
f256 = torch.rand(256, 129, device=dev) + 0.5
for i in range(REPS): hl.power_noise(f256, (128, 4, 256, 256), seed=1, stream_id=100 + i, plane_offset=0, factor=1.0)
torch.cuda.synchronize()
b4 = 256
ms = types.SimpleNamespace(sigma_min=torch.tensor(0.03), sigma_max=torch.tensor(14.6), timestep=lambda sg: (999 * (1 - (sg.log() - math.log(0.03)) / (math.log(14.6) - math.log(0.03)))).clamp(0, 999))
cond, uncond, xin = (torch.randn(b4, 4, 128, 128, device=dev) for _ in range(3))
wargs = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": xin - cond, "uncond": xin - uncond, "input": xin, "cond_scale": 7.0,
         "sigma": torch.full((b4,), 7.0, device=dev), "model": types.SimpleNamespace(model_sampling=ms),
         "model_options": {"transformer_options": {"sample_sigmas": torch.cat([torch.linspace(14.6, 0.03, 20), torch.zeros(1)])}}}
# WaveletCFG: the placeholder rule (low-pass kernel), a rule with per-orientation difference scales and one that also scales cond / uncond
# (level 1 by the tile kernels, deeper levels by the LDS-resident band kernel), then the same two rules through the single-launch band kernel
rules = {"placeholder": dict(difference=dict(yl_scale=5.0, yh_scales=3.0)),
         "bands_difference": dict(difference=dict(yl_scale=5.0, yh_scales=[[3.0, 2.5, 2.0]] * 5)),
         "bands_pair": dict(cond=dict(yl_scale=1.1, yh_scales=1.0), uncond=dict(yl_scale=1.0, yh_scales=0.9), difference=dict(yl_scale=5.0, yh_scales=3.0))}
for hp in (True, False):
    fns = {k: wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(**v, high_precision_mode=hp)) for k, v in rules.items()}
    for k in ("placeholder", "bands_difference", "bands_pair"):
        for _ in range(REPS): fns[k](wargs)
    wc.WaveletCFG.single_launch_bands = hp  # fp64 only: in fp32 the single-launch kernel and the deeper levels' kernel are one instantiation
    for k in (("bands_difference", "bands_pair") if hp else ()):
        fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(**rules[k], high_precision_mode=hp))
        for _ in range(REPS): fn(wargs)
    wc.WaveletCFG.single_launch_bands = None
    torch.cuda.synchronize()
del cond, uncond, xin, wargs
# cfg5's Brownian source on one rank's shard (128 Flux latents): one new path point per call, bridged between kept tensors; then the
# same folded into a running sum (the chain form)
ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
xf = torch.zeros((128, 16, 128, 128), device=dev)
bt = ng.BrownianTreeNoiseSampler(xf, 0.03, 14.6, seed=7, tree_depth=0)  # the path of bridges (the traffic row is its)
sg = torch.linspace(14.6, 1.0, 2 * REPS + 3).tolist()
for k in range(REPS): bt(torch.tensor(sg[k]), torch.tensor(sg[k + 1]))
acc = torch.randn_like(xf)
for k in range(REPS, 2 * REPS): bt.accumulate(acc, 0.5, 0.2, hl.new_partials(dev), torch.tensor(sg[k]), torch.tensor(sg[k + 1]))
torch.cuda.synchronize()
# round 6: the look-ahead forms a prepared plan runs in a sampler's steady state (one launch per call): normalised uniform fill, Perlin
# (fused: final pass + the next call's statistics in the same waves), pyramid (this call's planes + the next call's statistics), and the
# Brownian tree call (two-stage evaluation over the coarse grid, ~20 node bursts per element)
hl.PLANS_ENABLED = True
for name, xb in (("uniform", x), ("perlin", x), ("pyramid", x), ("pyramid", x64)):
    s = nz.get_noise_sampler(name, xb, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    for _ in range(REPS + 6): s(*sig)   # warm calls, the traced call, then replays
    torch.cuda.synchronize()
hl.PLANS_ENABLED = False
bt = ng.BrownianTreeNoiseSampler(xf, 0.03, 14.6, seed=7, tree_depth=24)
sg = torch.linspace(14.6, 1.0, 2 * REPS + 3).tolist()
for k in range(2 * REPS): bt(torch.tensor(sg[k]), torch.tensor(sg[k + 1]))
torch.cuda.synchronize()
print("workload done")
