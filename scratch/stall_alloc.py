"""Does a fresh device allocation (hipMalloc of a new block by torch's caching allocator) freeze the process's queues for ~83 ms a little
later?  Light 25-call bursts of the single-latent power-law call for a few seconds; every SECS/8 a NEW block is forced from the driver
(MODE=small: 2 MB small-pool blocks; large: 64 MB; free: a cached block is released with empty_cache; none: control).  Bursts over 5 ms are
printed with the time since the last allocation event."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
dev = torch.device("cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
MODE = os.environ.get("MODE", "small")
SECS = float(os.environ.get("SECS", "6"))
ns = bench.power_item(pn).make_noise_sampler(torch.zeros((4, 4, 128, 128), device=dev), None, None, seed=None, cpu=False, normalized=True)
for _ in range(200): ns(*sig)
torch.cuda.synchronize()
keep = []
t_start = time.perf_counter(); last_ev = t_start; next_ev = t_start + SECS / 8; nev = 0; slow = []
nb = 0
while time.perf_counter() - t_start < SECS:
    now = time.perf_counter()
    if now >= next_ev and MODE != "none":
        r0 = torch.cuda.memory_reserved()
        if MODE == "small":
            keep += [torch.empty(300 * 1024, dtype=torch.uint8, device=dev) for _ in range(8)]   # small pool: new 2 MB blocks
        elif MODE == "large":
            keep.append(torch.empty(64 << 20, dtype=torch.uint8, device=dev))
        elif MODE == "free":
            keep.append(torch.empty(64 << 20, dtype=torch.uint8, device=dev)); keep.pop(); torch.cuda.empty_cache()
        nev += 1
        last_ev = time.perf_counter(); next_ev = last_ev + SECS / 8
        print(f"event {nev} ({MODE}) at +{last_ev - t_start:.3f} s: reserved {r0 >> 20} -> {torch.cuda.memory_reserved() >> 20} MiB, took {(last_ev - now) * 1e3:.2f} ms", flush=True)
    torch.cuda.synchronize()
    s = time.perf_counter()
    for _ in range(25): ns(*sig)
    torch.cuda.synchronize()
    d = time.perf_counter() - s
    nb += 1
    if d > 5e-3:
        slow.append(d)
        print(f"   burst at +{s - t_start:.3f} s took {d * 1e3:.1f} ms ({(s - last_ev) * 1e3:.1f} ms after the last allocation event)", flush=True)
print(f"{MODE}: {nb} bursts, {len(slow)} over 5 ms: {[round(x * 1e3, 1) for x in slow]}", flush=True)
