"""One fresh process: the rows named in ROWS (comma separated: b1, b4, 104x152, 256x256, perlin, pyramid), 100 warm-up calls + 16 bursts of 25
calls each; prints every burst over 5 ms (host ms / sync-wait ms) and a one-line summary.  Run many times to count how often a fresh
process meets the ~83 ms freeze and in which rows."""
import importlib, os, sys, time
T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
dev = torch.device("cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
hits = []
for tag in os.environ.get("ROWS", "b1,b4,104x152,256x256").split(","):
    if tag in ("b1", "b4"):
        ns = bench.power_item(pn).make_noise_sampler(torch.zeros((int(tag[1:]), 4, 128, 128), device=dev), None, None, seed=None, cpu=False, normalized=True)
    elif "x" in tag:
        hh, ww = map(int, tag.split("x"))
        ns = bench.power_item(pn).make_noise_sampler(torch.zeros((1, 4, hh, ww), device=dev), None, None, seed=None, cpu=False, normalized=True)
    else:
        ns = nz.get_noise_sampler(tag, torch.zeros((64, 4, 128, 128), device=dev), 0.03, 14.6, seed=None, cpu=False, normalized=True)
    for _ in range(100): ns(*sig)
    for b in range(16):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(25): ns(*sig)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        if t2 - t0 > 5e-3:
            hits.append(f"{tag} burst {b} at +{t0 - T0:.2f}s: host {(t1 - t0) * 1e3:.1f} ms, sync wait {(t2 - t1) * 1e3:.1f} ms")
print(f"{os.environ.get('ROWS')}: {len(hits)} slow bursts {hits}", flush=True)
