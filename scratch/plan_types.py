"""Every registry noise type at 4 latents, normalised and raw: which steps get a prepared plan (or why not), that the replayed step is the
ordinary step bit for bit, and the host / GPU time per call either way."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
B = int(os.environ.get("PLAN_BATCH", "4"))
x = torch.zeros((B, 4, 128, 128), device="cuda")

def run(ns, n, plans):
    hl.PLANS_ENABLED = plans
    torch.manual_seed(99)
    out = [ns(*sig).clone() for _ in range(n)]
    hl.PLANS_ENABLED = True
    return out

def timeit(ns, plans, n=300):
    hl.PLANS_ENABLED = plans
    for _ in range(60): ns(*sig)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): ns(*sig)
    host = (time.perf_counter() - t0) / n * 1e6
    e1.record(); torch.cuda.synchronize()
    hl.PLANS_ENABLED = True
    return host, e0.elapsed_time(e1) / n * 1e3

bad = 0
for t in nz.NoiseType:
    name = t.name.lower()
    for normalized in (True, False):
        try:
            mk = lambda: nz.get_noise_sampler(name, x, 0.03, 14.6, seed=5, cpu=False, normalized=normalized)
            a, b = mk(), mk()
            ra, rb = run(a, 8, True), run(b, 8, False)
        except NotImplementedError as e:
            print(f"{name:24s} not on this path ({str(e)[:50]})"); break
        except Exception as e:
            print(f"{name:24s} FAILED {type(e).__name__}: {str(e)[:100]}"); bad += 1; break
        same = all(torch.equal(p, q) or (torch.isnan(p) == torch.isnan(q)).all() and torch.equal(torch.nan_to_num(p), torch.nan_to_num(q)) for p, q in zip(ra, rb))
        pl = getattr(a, "_planned", None)
        state = "static? no" if pl is None else (f"plan x{hl.load().sonar_plan_length(pl.plan.handle)}" if pl.plan is not None else f"NO PLAN ({pl.reason})")
        hp, gp = timeit(a, True); ho, go = timeit(b, False)
        bad += not same
        print(f"{name:24s} norm={int(normalized)} {'same' if same else 'DIFFERENT'} | {state:60s} | plan {hp:6.1f} / {gp:6.1f} us | ordinary {ho:6.1f} / {go:6.1f} us", flush=True)
print("FAILURES:", bad)
