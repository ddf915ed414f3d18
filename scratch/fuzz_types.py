"""Generate mode (cpu=False) of every registry type on random shapes: runs, is finite, and a batch drawn as two shards (shard_offset)
equals the batch drawn at once.  python scratch/fuzz_types.py [iterations] [seed]"""
import importlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise"); ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SKIP = {"COLLATZ", "DISTRO", "VORONOI_FUZZ", "VORONOI_MIX", "PYRAMID_BISLERP", "HIGHRES_PYRAMID_BISLERP", "PYRAMID_MIX_BISLERP", "PYRAMID_OLD_BISLERP"}
types = [t for t in ng.NoiseType if t.name not in SKIP]
# normalise INSIDE the generator over the whole batch tensor (py/noise_generation.py:212-249, :680-704): a shard uses its own statistics
BATCH_NORMALISED = {"ONEF_GREENISH_MIX", "ONEF_PINKISH_MIX", "ONEF_PINKISHGREENISH", "RAINBOW_INTENSE", "RAINBOW_MILD", "PYRAMID_MIX", "PYRAMID_MIX_AREA",
                    "GREEN_TEST"}
bad = expected = 0
for it in range(iters):
    t = rnd.choice(types)
    b = rnd.randint(2, 5)
    c = rnd.choice([1, 3, 4, 16])
    h, w = rnd.choice([(8, 8), (16, 24), (32, 32), (64, 64), (20, 12), (18, 30), (7, 9), (40, 56), (33, 17), (4, 4), (128, 128), (135, 24), (5, 64)])
    frames = rnd.choice([0, 0, 0, 2])
    tail = (c, frames, h, w) if frames else (c, h, w)
    cut = rnd.randint(1, b - 1)
    seed = 100 + it

    def run(b0, n):
        torch.manual_seed(seed)
        with ng.shard_offset(b0):
            x = torch.zeros((n, *tail), device="cuda")
            ns = nz.get_noise_sampler(t, x, 0.03, 14.6, seed=seed, cpu=False, normalized=False)
            return [ns(torch.tensor(s), torch.tensor(sn)).clone() for s, sn in ((10.0, 7.0), (7.0, 4.0))]

    try:
        whole, lo, hi = run(0, b), run(0, cut), run(cut, b - cut)
    except Exception as exc:  # noqa: BLE001
        msg = str(exc)[:150]
        if isinstance(exc, ValueError) and ("dimension" in msg or "dims" in msg.lower()):
            continue  # the type does not take this rank (the reference refuses it too)
        print(f"[{it}] {t.name} {(b, *tail)} cut {cut}: {type(exc).__name__}: {msg}", flush=True)
        bad += 1
        continue
    for k in range(2):
        if not bool(torch.isfinite(whole[k]).all()):
            print(f"[{it}] {t.name} {(b, *tail)}: non-finite output", flush=True); bad += 1; break
        if not torch.equal(torch.cat([lo[k], hi[k]]), whole[k]):
            if t.name in BATCH_NORMALISED:
                expected += 1
                break
            d = float((torch.cat([lo[k], hi[k]]) - whole[k]).abs().max())
            print(f"[{it}] {t.name} {(b, *tail)} cut {cut} call {k}: shards != whole, max diff {d:.3e}", flush=True); bad += 1; break
print(f"{iters} draws, {bad} problems ({expected} shard differences of the types that normalise inside the generator over the batch)")
