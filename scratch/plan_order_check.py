"""Does a planned power-law step slow down after steps of other shapes ran in the same process?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
item = pn.PowerNoiseItem(1.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1")


def run(shape, tag):
    x = torch.zeros(shape, device="cuda")
    ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
    for _ in range(60): ns(None, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(300): ns(None, None)
    host = (time.perf_counter() - t0) / 300 * 1e6
    e1.record(); torch.cuda.synchronize()
    info = {k: getattr(ns, k, None) for k in ("reason", "runs", "fallbacks", "attempts")}
    print(tag, shape, f"host {host:5.1f} us, GPU span {e0.elapsed_time(e1) / 300 * 1e3:5.1f} us", info, flush=True)


order = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1, 4, 128, 128), (1, 4, 104, 152), (1, 4, 128, 128), (4, 4, 128, 128), (1, 4, 128, 128)]
big = torch.zeros(64 << 20, device="cuda")
for i, shape in enumerate(order):
    if os.environ.get("HEAVY_BETWEEN"):
        for _ in range(400): big.add_(1.0)  # ~30 ms of bandwidth-bound work: keeps the clocks up
    run(shape, f"#{i}")

if os.environ.get("PROFILE_LAST"):
    import cProfile, pstats
    x = torch.zeros(order[-1], device="cuda")
    ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
    for _ in range(60): ns(None, None)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): ns(None, None)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    print("plan:", ns.plan is not None, "reason", ns.reason, "lookahead hits/misses", [(h.la.hits, h.la.misses) for h in (ns.plan.hooks if ns.plan else []) if hasattr(h, "la") and hasattr(h.la, "hits")])
