import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
sig = (torch.tensor(14.6), torch.tensor(10.0))
x64 = torch.zeros((64, 4, 128, 128), device="cuda")
for seed in (0, 1, 12345):
    torch.manual_seed(seed)
    ns = nz.get_noise_sampler("pyramid", x64, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    times = []
    for i in range(3000):
        t0 = time.perf_counter(); ns(*sig); times.append((time.perf_counter() - t0) * 1e6)
        if i % 500 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    pl = ns._planned
    slow = [i for i, t in enumerate(times) if t > 100]
    print(seed, f"mean {sum(times)/len(times):.1f} us, slow calls {len(slow)} first {slow[:10]}", "runs", pl.plan.runs if pl.plan else None, flush=True)
    # which level tables do the slow calls have?
    gen = ns.noise_sampler
    torch.manual_seed(seed)
    bad = 0
    for i in range(3000):
        s, st = ng.DeviceRNG.take(2 + gen.iterations)
        lv = hl.AutoLevels(128, 128, gen.iterations, gen.discount, s, st)
        grid = sum(h * w for _, h, w, _ in lv if (h, w) != (128, 128))
        rows = sum(h for _, h, w, _ in lv if (h, w) != (128, 128))
        n = sum(1 for _, h, w, _ in lv if (h, w) != (128, 128))
        lds = grid * 4 + n * 256 * 16
        if lds > 65536: bad += 1
        if i in slow[:3]: print("   call", i, [(h, w) for _, h, w, _ in lv], "lds", lds, "with rows", lds + rows * 512)
    print("   level tables beyond the LDS budget:", bad)
