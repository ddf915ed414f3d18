"""Hunts the intermittent ~80 ms host stall in launch-bound rows (round 4's VERDICT item 3): the eight launch-bound rows of bench.py,
REPS times each, per-burst host and GPU time logged, with Python's garbage collector watched (gc.callbacks): every collection's
generation, duration and the burst it fell into.  GCMODE=default | freeze | off selects the collector's setting for the timed loops."""
import gc, importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
import bench

pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
dev = torch.device("cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
REPS = int(os.environ.get("REPS", "50"))
MODE = os.environ.get("GCMODE", "default")
C, H, W = 4, 128, 128

# what bench.py does before these rows: two 1 GiB buffers come and go
a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev); b = torch.empty_like(a); b.copy_(a); torch.cuda.synchronize(); del a, b

rows = {}
for tag, bsz in (("power_b1", 1), ("power_b4", 4)):
    rows[tag] = bench.power_item(pn).make_noise_sampler(torch.zeros((bsz, C, H, W), device=dev), None, None, seed=None, cpu=False, normalized=True)
for tag, (hh, ww) in (("power_104x152_b1", (104, 152)), ("power_256x256_b1", (256, 256))):
    rows[tag] = bench.power_item(pn).make_noise_sampler(torch.zeros((1, C, hh, ww), device=dev), None, None, seed=None, cpu=False, normalized=True)
for tag, bsz in (("cfg3_chain_b4", 4), ("cfg3_chain_b64", 64)):
    ch = nz.CustomNoiseChain(); ch.add(nz.CustomNoiseItem(0.5, noise_type="perlin")); ch.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
    rows[tag] = ch.make_noise_sampler(torch.zeros((bsz, C, H, W), device=dev), 0.03, 14.6, seed=None, cpu=False, normalized=True)
x64 = torch.zeros((64, C, H, W), device=dev)
for name in ("perlin", "pyramid"):
    rows[f"{name}_b64"] = nz.get_noise_sampler(name, x64, 0.03, 14.6, seed=None, cpu=False, normalized=True)

events = []  # (t_start, generation, seconds)
_t = [0.0]
def on_gc(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        events.append((_t[0], info["generation"], time.perf_counter() - _t[0], info.get("collected", 0)))
gc.callbacks.append(on_gc)
if MODE == "freeze":
    gc.collect(); gc.freeze()
elif MODE == "off":
    gc.collect(); gc.disable()
print(f"gc mode {MODE}; thresholds {gc.get_threshold()}; objects tracked {len(gc.get_objects())}", flush=True)

for tag, ns in rows.items():
    for _ in range(100):
        ns(*sig)
    torch.cuda.synchronize()
    med_h, med_g, worst = [], [], []
    for rep in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        bursts = []
        for burst in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); e0.record()
            for _ in range(25):
                ns(*sig)
            t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
            bursts.append((t0, (t1 - t0) / 25 * 1e6, e0.elapsed_time(e1) / 25 * 1e3))
        hs = [h for _, h, _ in bursts]; gs = [g for _, _, g in bursts]
        med_h.append(statistics.median(hs)); med_g.append(statistics.median(gs))
        for t0, h, g in bursts:
            if h > 3 * statistics.median(hs) or g > 3 * statistics.median(gs):
                inside = [(gen, round(dt * 1e3, 1)) for (ts, gen, dt, _) in events if t0 <= ts <= t0 + h * 25e-6 + 1e-3]
                worst.append((rep, round(h, 1), round(g, 1), inside))
    print(f"{tag:18s} host median {statistics.median(med_h):6.1f} us (min {min(med_h):.1f} max {max(med_h):.1f})  gpu median {statistics.median(med_g):6.1f} us "
          f"(min {min(med_g):.1f} max {max(med_g):.1f})  slow bursts: {worst[:6]}{' ...' if len(worst) > 6 else ''} ({len(worst)} of {REPS * 8})", flush=True)
print("collections by generation:", {g: sum(1 for e in events if e[1] == g) for g in (0, 1, 2)},
      "; longest:", sorted(((round(dt * 1e3, 1), gen) for _, gen, dt, _ in events), reverse=True)[:8], "ms", flush=True)
