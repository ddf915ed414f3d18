"""Randomised check of the virtual Brownian tree (two-stage evaluation, round 6): an increment is the same BITS after any query history,
on any tree depth and sigma range, for shards, per-latent seeds and folded calls; increments over abutting intervals add up; times inside
one grid cell give zeros.  python scratch/fuzz_brownian_r6.py [iterations] [seed]"""
import importlib, math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
T = lambda v: torch.tensor(v, dtype=torch.float64)  # noqa: E731
for it in range(iters):
    b, c = rnd.randint(1, 4), rnd.choice([1, 4, 16])
    h, w = rnd.choice([(32, 32), (64, 64), (16, 64), (20, 12), (128, 128), (7, 9)])
    per_latent_ok = (c * h * w) % 4 == 0  # (per-latent seeds need latents of a multiple of four elements: the entry point says so)
    depth = rnd.choice([6, 12, 16, 24, 24, 30])
    lo = rnd.choice([0.03, 0.0292, 0.5, 1e-3])
    hi = lo + rnd.choice([14.57, 1.0, 80.0, 0.25])
    seed = rnd.choice([rnd.randrange(1 << 40), [rnd.randrange(1000) for _ in range(b)] if per_latent_ok else 7])
    x = torch.zeros((b, c, h, w), device="cuda")
    mk = lambda: ng.BrownianTreeNoiseSampler(x, lo, hi, seed=seed, tree_depth=depth)  # noqa: E731
    lo, hi = mk().path.t_lo, mk().path.t_hi  # (the range as the sampler holds it: the constructor's float32 tensors rounded the ends)
    pick = lambda: lo + (hi - lo) * rnd.choice([rnd.random(), rnd.random() ** 3, 0.5, 0.25, 1.0, 0.0, rnd.randrange(1 << 12) / (1 << 12)])  # noqa: E731
    try:
        a_, b_ = mk(), mk()
        hist = []
        for _ in range(rnd.randint(0, 6)):  # a history for the first instance
            t0, t1 = pick(), pick()
            if a_.path.resolve(t0) != a_.path.resolve(t1):
                a_(T(t0), T(t1))
                hist.append((t0, t1))
        ta, tb, tc = sorted((pick(), pick(), pick()))
        ra, rb, rc = (a_.path.resolve(v) for v in (ta, tb, tc))
        if ra == rb or rb == rc:
            continue
        one = a_(T(tc), T(ta))
        two = b_(T(tc), T(ta))
        ok = torch.equal(one, two)
        p1, p2 = b_(T(tc), T(tb)), b_(T(tb), T(ta))
        whole = (p1 * math.sqrt(rc - rb) + p2 * math.sqrt(rb - ra)) / math.sqrt(rc - ra)
        err = float((whole - two).abs().max())
        acc = torch.full_like(x, 1.5)
        assert mk().accumulate(acc, 0.5, 2.0, None, T(tc), T(ta))
        ferr = float((acc - (0.75 + 2.0 * two)).abs().max())
        if not ok or err > 5e-5 or ferr > 2e-6 or not torch.isfinite(two).all():
            bad += 1
            print(f"[{it}] shape {tuple(x.shape)} depth {depth} range [{lo}, {hi}] times {ta}, {tb}, {tc}: equal={ok} (max diff {float((one - two).abs().max()):.2e}) additivity {err:.2e} fold {ferr:.2e} history {hist}", flush=True)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print(f"[{it}] shape {tuple(x.shape)} depth {depth} range [{lo}, {hi}]: {type(exc).__name__}: {exc}", flush=True)
print(f"{iters} cases, {bad} bad")
sys.exit(1 if bad else 0)
