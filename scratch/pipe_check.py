"""A/B of the pipelined power kernel against the phase-serial one: run once with SONAR_POWER_PIPE=0 and once without; prints a hash of
the outputs (must agree bit for bit: same streams, same arithmetic) and the launch-pair time at several batch sizes."""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")
sig = (torch.tensor(14.6), torch.tensor(10.0))
tag = "pipe" if os.environ.get("SONAR_POWER_PIPE", "1") != "0" else "serial"
for B, C in ((512, 4), (256, 4), (128, 4), (96, 4), (171, 3), (65, 4)):
    x = torch.zeros((B, C, 128, 128), device="cuda")
    for normalized in (True, False):
        torch.manual_seed(5)
        ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=normalized)
        out = ns(*sig)
        torch.cuda.synchronize()
        h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
        for _ in range(200): ns(*sig)
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(100): ns(*sig)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 10)
        print(f"{tag} B={B} C={C} norm={normalized}: {h} std={out.std().item():.5f} {best:.1f} us/step {B/best:.3f} M latents/s", flush=True)
