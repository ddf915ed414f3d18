"""Off-fast-path planes, 33.5 M values per call: the direct two-launch call bench.py times (hl.power_noise) beside the sampler API (plans, look-ahead)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for tag, (hh, ww, nb) in {"128x128": (128, 128, 512), "64x64": (64, 64, 2048), "104x152": (104, 152, 530), "112x144": (112, 144, 520), "96x168": (96, 168, 520), "256x256": (256, 256, 128), "32x32": (32, 32, 8192)}.items():
    fz = torch.rand(hh, ww // 2 + 1, device="cuda") + 0.5
    shp, ctr = (nb, 4, hh, ww), [0]
    def direct():
        ctr[0] += 1
        return hl.power_noise(fz, shp, seed=11, stream_id=ctr[0], plane_offset=0, factor=1.0)
    try:
        d = bench.event_us(direct, 20, 5)
    except Exception as exc:
        d = float("nan")
    x = torch.zeros(shp, device="cuda")
    ns = bench.power_item(pn).make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
    s = bench.event_us(lambda: ns(*sig), 30, 10)
    la = getattr(getattr(ns, "_planned", ns), "plan", None)
    print(f"{tag:8s} planes {nb * 4:5d}: direct {d:7.1f} us   sampler {s:7.1f} us   ahead_ok={hl.load().sonar_power_noise_ahead_ok(nb * 4, hh, ww, 4)} plan={'yes' if la else 'no'}", flush=True)
