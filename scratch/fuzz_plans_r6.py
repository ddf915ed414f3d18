"""Randomised differential run of the round-6 look-ahead forms: a normalised sampler of a random type on a random latent shape (odd sizes,
5-D video latents, shards at an element offset) is called a dozen times with prepared plans on and off -- reseeds and foreign draws at
random places -- and both runs must give the same bits.  python scratch/fuzz_plans_r6.py [iterations] [seed]"""
import importlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BASE = ["gaussian", "uniform", "perlin", "pyramid", "pyramid_area", "pyramid_discount5", "pyramid_mix", "pyramid_mix_area", "pyramid_old", "pyramid_old_area",
        "laplacian", "power_old", "pink_old", "white", "grey", "velvet", "violet", "onef_pinkish", "onef_greenish", "onef_pinkishgreenish", "onef_pinkish_mix",
        "onef_greenish_mix", "green_test", "rainbow_mild", "rainbow_intense", "brownian", "power"]
if os.environ.get("FUZZ_ALL"):
    KINDS = BASE + ["chain"] * 12
else:
    KINDS = ["uniform", "gaussian", "gaussian", "perlin", "pyramid", "pyramid", "chain:uniform+pyramid", "chain:gaussian+perlin", "chain:pyramid+uniform", "laplacian"]
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")


def power_item(f):
    return pn.PowerNoiseItem(f, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0,
                             channel_correlation="1,1,1,1,1,1")
sig = (torch.tensor(14.6), torch.tensor(10.0))
bad = 0
hooks_seen = {}
for it in range(iters):
    b, c = rnd.randint(1, 6), rnd.choice([1, 3, 4, 16])
    h, w = rnd.choice([(8, 8), (16, 24), (32, 32), (64, 64), (20, 12), (18, 30), (7, 9), (40, 56), (128, 128), (33, 17), (4, 4), (104, 152), (128, 64)])
    frames = rnd.choice([0, 0, 0, 3])
    shape = (b, c, frames, h, w) if frames else (b, c, h, w)
    kind = rnd.choice(KINDS)
    if kind == "chain":
        kind = "chain:" + "+".join(rnd.choice(BASE) for _ in range(rnd.randint(2, 3)))
    if "power" in kind.replace("power_old", "") and (h % 2 or w % 2 or frames):
        kind = kind.replace("power_old", "PO").replace("power", "gaussian").replace("PO", "power_old")
    normalized = rnd.choice([True, True, False])
    factor = rnd.choice([1.0, 1.0, 0.7, 1.3])
    offset = rnd.choice([0, 0, 0, 2, 5])
    events = {rnd.randint(3, 11): rnd.choice(["reseed", "foreign"]) for _ in range(rnd.randint(0, 2))}
    x = torch.zeros(shape, device="cuda")

    def make():
        if kind.startswith("chain:"):
            chain = nz.CustomNoiseChain()
            for name in kind[6:].split("+"):
                chain.add(power_item(0.5) if name == "power" else nz.CustomNoiseItem(0.5, noise_type=name))
            return chain.make_noise_sampler(x, 0.03, 14.6, seed=(5 if "brownian" in kind else None), cpu=False, normalized=normalized)
        if kind == "power":
            return power_item(factor).make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=normalized)
        return nz.get_noise_sampler(kind, x, 0.03, 14.6, seed=(5 if kind == "brownian" else None), cpu=False, normalized=normalized, factor=factor)

    def run(plans):
        hl.PLANS_ENABLED = plans
        torch.manual_seed(1000 + it)
        with ng.shard_offset(offset):
            ns = make()
            outs = []
            for k in range(12):
                ev = events.get(k)
                if ev == "reseed":
                    torch.manual_seed(5000 + it)
                elif ev == "foreign":
                    torch.randn(3, device="cuda")
                sg = (torch.tensor(14.6 * 0.9**k), torch.tensor(14.6 * 0.9**(k + 1))) if "brownian" in kind else sig
                outs.append(ns(*sg).clone())
        return ns, outs

    try:
        na, a = run(True)
        _, bb = run(False)
    except Exception as exc:  # noqa: BLE001
        print(f"[{it}] {kind} {shape} factor {factor} offset {offset}: {type(exc).__name__}: {exc}", flush=True)
        bad += 1
        continue
    finally:
        hl.PLANS_ENABLED = True
    diff = [k for k, (p, q) in enumerate(zip(a, bb)) if not torch.equal(p, q)]
    planned = na if isinstance(na, hl.Planned) else getattr(na, "_planned", None)
    plan = planned.plan if planned is not None else None
    for hk in (plan.hooks if plan else []):
        hooks_seen[type(hk).__name__] = hooks_seen.get(type(hk).__name__, 0) + 1
    if diff:
        bad += 1
        worst = max(float((p - q).abs().max()) for p, q in zip(a, bb))
        print(f"[{it}] {kind} {shape} normalized {normalized} factor {factor} offset {offset} events {events}: calls {diff} differ (max {worst:.3e}); plan={'yes' if plan else 'no'}", flush=True)
print(f"{iters} cases, {bad} bad; hooks in the plans: {hooks_seen}")
sys.exit(1 if bad else 0)
