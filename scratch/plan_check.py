"""Prepared plans against the ordinary path: same bits over a run of calls, which steps got a plan (or why not), host and GPU time per call."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
sig = (torch.tensor(14.6), torch.tensor(10.0))


def power_item(factor=1.0):
    return pn.PowerNoiseItem(factor, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                             mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")


def makers(x):
    def chain(*items, normalized=True):
        def mk():
            c = nz.CustomNoiseChain()
            for f, t in items:
                c.add(power_item(f) if t == "power" else nz.CustomNoiseItem(f, noise_type=t))
            return c.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=normalized)
        return mk
    out = {}
    for name in ("gaussian", "uniform", "perlin", "pyramid", "pyramid_area"):
        out[name] = lambda name=name: nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        out[name + "_raw"] = lambda name=name: nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=False, factor=0.7)
    out["power"] = lambda: power_item().make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
    out["chain perlin+pyramid"] = chain((0.5, "perlin"), (0.5, "pyramid"))
    out["chain gaussian+perlin"] = chain((0.6, "gaussian"), (0.4, "perlin"))
    out["chain power+perlin+gaussian"] = chain((0.5, "power"), (0.3, "perlin"), (0.2, "gaussian"))
    out["chain pyramid unnormalised"] = chain((1.0, "pyramid"), normalized=False)
    return out


def run(ns, n, plans):
    hl.PLANS_ENABLED = plans
    torch.manual_seed(1234)
    outs = []
    for _ in range(n):
        t = ns(*sig)
        outs.append((t.clone(), getattr(t, hl.STATS_ATTR, None) is not None))
    hl.PLANS_ENABLED = True
    return outs


def timeit(ns, plans, n=400):
    hl.PLANS_ENABLED = plans
    for _ in range(100): ns(*sig)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): ns(*sig)
    host = (time.perf_counter() - t0) / n * 1e6
    e1.record(); torch.cuda.synchronize()
    hl.PLANS_ENABLED = True
    return host, e0.elapsed_time(e1) / n * 1e3


bad = 0
for B in (1, 4, 64, 512):
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    for name, mk in makers(x).items():
        a, b = mk(), mk()
        ra, rb = run(a, 9, True), run(b, 9, False)
        same = all(torch.equal(p[0], q[0]) and p[1] == q[1] for p, q in zip(ra, rb))
        inner = a if isinstance(a, hl.Planned) else getattr(a, "_planned", None)
        state = "no Planned wrapper" if inner is None else ("plan of %d records, %d runs" % (hl.load().sonar_plan_length(inner.plan.handle), inner.plan.runs)
                                                            if inner.plan is not None else f"NO PLAN: {inner.reason}")
        dfr = getattr(a, "deferred", None)
        if dfr is not None and isinstance(dfr, hl.Planned):
            da, db = [], []
            for ns_, acc_, pl in ((a, da, True), (b, db, False)):
                hl.PLANS_ENABLED = pl
                torch.manual_seed(77)
                for _ in range(7):
                    t, norm = ns_.deferred(*sig)
                    acc_.append((t.clone(), None if norm is None else norm.clone()))
            hl.PLANS_ENABLED = True
            same_d = all(torch.equal(p[0], q[0]) and ((p[1] is None) == (q[1] is None)) and (p[1] is None or torch.equal(p[1], q[1])) for p, q in zip(da, db))
            state += f"; deferred: {'same' if same_d else 'DIFFERENT'} ({'plan' if dfr.plan is not None else 'no plan: ' + str(dfr.reason)})"
            bad += not same_d
        hp, gp = timeit(a, True)
        ho, go = timeit(b, False)
        bad += not same
        print(f"B={B:3d} {name:30s} {'same bits' if same else 'DIFFERENT'} | {state} | plan: host {hp:5.1f} us, GPU span {gp:6.1f} us | ordinary: host {ho:5.1f}, GPU span {go:6.1f}", flush=True)
print("FAILURES:", bad)
