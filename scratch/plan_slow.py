import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
x = torch.zeros((4, 4, 128, 128), device="cuda")
ns = nz.get_noise_sampler("onef_pinkish_mix", x, 0.03, 14.6, seed=5, cpu=False, normalized=True)
for _ in range(10): ns(*sig)
plan = ns._planned.plan
print("records", hl.load().sonar_plan_length(plan.handle), "fresh", plan.fresh, "scratch", [(s[1], s[2]) for s in plan.scratch])
import cProfile, pstats
times = []
for i in range(400):
    t0 = time.perf_counter(); ns(*sig); times.append((time.perf_counter() - t0) * 1e6)
torch.cuda.synchronize()
times.sort()
print("host per call: median %.1f  p90 %.1f  max %.1f  mean %.1f" % (times[200], times[360], times[-1], sum(times) / 400))
# where: time the pieces of Plan.run by hand
real_run = hl._lib.sonar_plan_run
acc = {"run": 0.0, "n": 0}
class Lib:
    def __getattr__(self, k): return getattr(real, k)
    def sonar_plan_run(self, *a):
        t0 = time.perf_counter(); r = real.sonar_plan_run(*a); acc["run"] += time.perf_counter() - t0; acc["n"] += 1; return r
real = hl._lib; hl._lib = Lib()
t0 = time.perf_counter()
for i in range(400): ns(*sig)
tot = time.perf_counter() - t0
torch.cuda.synchronize()
print("total %.1f us per call, inside sonar_plan_run %.1f us" % (tot / 400 * 1e6, acc["run"] / acc["n"] * 1e6))
hl._lib = real
pr = cProfile.Profile(); pr.enable()
for i in range(300): ns(*sig)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
