"""Does the ~10-90 ms stall of launch-bound rows follow a heavy -> light load transition (a power-state change)?  Cycles of 0.3 s of 1 GiB
copies followed by LIGHT seconds of 25-call bursts of the single-latent power-law call; every burst longer than 1 ms is printed with the time
since the heavy phase ended.  The clock levels sysfs shows (if readable) are sampled before / after."""
import glob, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg, bench
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
dev = torch.device("cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
LIGHT = float(os.environ.get("LIGHT", "1.5"))
HEAVY = os.environ.get("HEAVY", "copy")
def clocks():
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_[smf]clk")):
        try:
            cur = [l.strip() for l in open(f) if "*" in l]
            out.append(os.path.basename(f)[7:] + "=" + (cur[0] if cur else "?"))
        except OSError as e:
            out.append(os.path.basename(f) + ":" + type(e).__name__)
    return " ".join(out) or "no sysfs clocks"
ns = bench.power_item(pn).make_noise_sampler(torch.zeros((4, 4, 128, 128), device=dev), None, None, seed=None, cpu=False, normalized=True)
a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev); b = torch.empty_like(a)
x512 = torch.zeros((512, 4, 128, 128), device=dev)
ns512 = bench.power_item(pn).make_noise_sampler(x512, None, None, seed=None, cpu=False, normalized=True)
for _ in range(100): ns(*sig)
torch.cuda.synchronize()
print("start:", clocks(), flush=True)
for cycle in range(6):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        if HEAVY == "copy":
            b.copy_(a)
        elif HEAVY == "power":
            for _ in range(20): ns512(*sig)
        torch.cuda.synchronize()
    t_end = time.perf_counter()
    print(f"cycle {cycle}: heavy ({HEAVY}) done:", clocks(), flush=True)
    nburst = slow = 0
    worst = 0.0
    while time.perf_counter() - t_end < LIGHT:
        torch.cuda.synchronize()
        s = time.perf_counter()
        for _ in range(25): ns(*sig)
        torch.cuda.synchronize()
        d = time.perf_counter() - s
        nburst += 1
        worst = max(worst, d)
        if d > 1e-3:
            slow += 1
            print(f"   burst {nburst} at +{(s - t_end) * 1e3:7.1f} ms took {d * 1e3:6.2f} ms", flush=True)
    print(f"   {nburst} bursts, {slow} over 1 ms, longest {worst * 1e3:.2f} ms; light done:", clocks(), flush=True)
