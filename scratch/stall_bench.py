"""bench.py's secondary rows with per-burst logging of the launch-bound rows (the ~80 ms stall of round 4's VERDICT item 3 shows up in one of
their first rows in every bench run): which burst, how long on the host and the GPU clock, wall-clock since process start."""
import gc, importlib, os, sys, time
T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
dev = torch.device("cuda", 0)
row = [0]
gcs = []
gc.callbacks.append(lambda ph, info: gcs.append((time.perf_counter() - T0, ph, info["generation"])))
def logged(fn, steps=200, warmup=100, burst=25):
    row[0] += 1
    tw = time.perf_counter()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    warm_ms = (time.perf_counter() - tw) * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = gpu = 0.0
    log = []
    for b in range(steps // burst):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record()
        for _ in range(burst):
            fn()
        t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        h, g = (t1 - t0) * 1e3, e0.elapsed_time(e1)
        host += t1 - t0; gpu += g
        log.append(f"{h:.2f}/{g:.2f}/{(t2 - t1) * 1e3:.2f}")
    print(f"row {row[0]} at +{(tw - T0):.2f} s: warm-up {warm_ms:.1f} ms; bursts host/gpu/sync-wait ms: {' '.join(log)}", flush=True)
    return host / steps * 1e6, gpu / steps * 1e3
bench.host_and_event_us = logged
torch.manual_seed(0)
x = torch.zeros((bench.BATCH, 4, 128, 128), device=dev)
sig = (torch.tensor(14.6), torch.tensor(10.0))
ns = bench.power_item(pn).make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
for _ in range(1300): ns(*sig)   # what main() runs before the secondary rows
torch.cuda.synchronize()
print(f"headline loop done at +{time.perf_counter() - T0:.2f} s", flush=True)
kernels, extra = bench.secondary_rows(dev, hl, pn, ng, nz, x, sig)
print({k: round(v, 1) for k, v in extra.items() if k.endswith("host_us_per_call")})
print("gc events:", [(round(t, 2), ph, g) for t, ph, g in gcs if g == 2 or ph == "start"][:20])
