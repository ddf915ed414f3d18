"""Randomised differential run of noise chains: random latent shapes (odd sizes, 5-D video latents, tiny planes) and random item lists;
the hosted route (an item evaluated inside the next item's kernel), the folded route (every item folds by itself) and the plain route
(generate + accumulation kernel) must give the same unnormalised sum, bit for bit.  python scratch/fuzz_chains.py [iterations] [seed]"""
import importlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
N = importlib.import_module("comfyui_sonar_amd.py.noise")
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KINDS = ["gaussian", "perlin", "pyramid", "brownian", "uniform", "power", "laplacian", "studentt", "onef_pinkish", "pyramid_area", "highres_pyramid", "velvet"]
bad = 0
for it in range(iters):
    b, c = rnd.randint(1, 4), rnd.choice([1, 3, 4, 16])
    h, w = rnd.choice([(8, 8), (16, 24), (32, 32), (64, 64), (20, 12), (18, 30), (7, 9), (40, 56), (128, 128), (33, 17), (4, 4)])
    frames = rnd.choice([0, 0, 0, 3])
    shape = (b, c, frames, h, w) if frames else (b, c, h, w)
    items = [rnd.choice(KINDS) for _ in range(rnd.randint(2, 4))]
    if "power" in items and (h % 2 or w % 2 or frames):
        items = [k if k != "power" else "gaussian" for k in items]
    factors = [rnd.choice([1.0, 0.5, 0.3, -0.7, 0.25]) for _ in items]
    offset = rnd.choice([0, 0, 2, 5])

    def build():
        chain = N.CustomNoiseChain()
        for f, name in zip(factors, items):
            if name == "power":
                chain.add(pn.PowerNoiseItem(f, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                                            mix=1.0, common_mode=0.0, channel_correlation="1"))
            else:
                chain.add(N.CustomNoiseItem(f, noise_type=name))
        return chain

    x = torch.zeros(shape, device="cuda")
    runs = []
    real_prefix, real_acc = N.NoiseSampler.accepts_prefix, N.NoiseSampler.accumulate
    try:
        for variant in ("hosted", "folded", "plain"):
            if variant == "folded":
                N.NoiseSampler.accepts_prefix = property(lambda self: False)
            elif variant == "plain":
                del N.NoiseSampler.accumulate
            torch.manual_seed(1000 + it)
            with ng.shard_offset(offset):
                ns = build().make_noise_sampler(x, 0.03, 14.6, seed=5 + it, cpu=False, normalized=False)
                runs.append([ns(torch.tensor(s), torch.tensor(sn)).clone() for s, sn in ((10.0, 7.0), (7.0, 4.0))])
            N.NoiseSampler.accepts_prefix, N.NoiseSampler.accumulate = real_prefix, real_acc
    except Exception as exc:  # noqa: BLE001
        N.NoiseSampler.accepts_prefix, N.NoiseSampler.accumulate = real_prefix, real_acc
        print(f"[{it}] {shape} {items} {factors} offset {offset}: {type(exc).__name__}: {str(exc)[:160]}", flush=True)
        bad += 1
        continue
    ok = all(torch.equal(a, b_) and torch.equal(b_, c_) and bool(torch.isfinite(a).all()) for a, b_, c_ in zip(*runs))
    if not ok:
        bad += 1
        d = max(float((a - c_).abs().max()) for a, _b, c_ in zip(*runs))
        print(f"[{it}] MISMATCH {shape} {items} {factors} offset {offset}: max |hosted - plain| = {d:.3e}", flush=True)
print(f"{iters} chains, {bad} problems")
