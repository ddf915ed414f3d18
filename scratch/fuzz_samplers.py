"""Randomised sampler runs: random momentum configurations (modes, blends, step gates, directions, history inits), shapes, step counts
and eta / s_noise, noise from a fixed bank -- the device samplers (comfyui_sonar_amd.py.sonar) against oracle/sonar_oracle.py's
restatement (pinned bit-exactly to the reference by tests/golden/make_golden.py).  python scratch/fuzz_samplers.py [iterations] [seed]"""
import importlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
from oracle import sonar_oracle as orc
pkg = sonar_pkg.load(); pkg.hip_lib.load()
S = importlib.import_module("comfyui_sonar_amd.py.sonar")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BLENDS = ["lerp", "inject", "subtract_b", None]


def fake_model(x, sigma, **_kw):
    s = sigma.reshape(-1, *([1] * (x.ndim - 1)))
    return x * 0.5 + torch.tanh(x) * (0.1 * s / (1.0 + s))


bad = 0
for it in range(iters):
    kw = {}
    if rnd.random() < 0.7: kw["momentum"] = rnd.choice([0.95, 0.8, 0.3, 1.0, 0.0, 1.2])
    if rnd.random() < 0.7: kw["momentum_hist"] = rnd.choice([0.75, 0.5, 1.0, 0.0, 0.2])
    if rnd.random() < 0.5: kw["direction"] = rnd.choice([1.0, -0.5, 1.5, -1.0])
    if rnd.random() < 0.6: kw["momentum_mode"] = rnd.choice(["NEW", "CLASSIC", "DENOISED"])
    if rnd.random() < 0.6: kw["init"] = rnd.choice(["ZERO", "SAMPLE", "SAMPLE_NORM"])
    if rnd.random() < 0.3: kw["momentum_start_step"] = rnd.randint(0, 3)
    if rnd.random() < 0.3: kw["momentum_end_step"] = rnd.randint(1, 6)
    if rnd.random() < 0.3: kw["always_update_history"] = rnd.random() < 0.5
    for k in ("blend_mode", "momentum_blend_mode", "history_blend_mode"):
        if rnd.random() < 0.3:
            v = rnd.choice(BLENDS)
            if v is not None or k != "blend_mode":
                kw[k] = v
    shape = rnd.choice([(2, 4, 8, 8), (1, 3, 10, 14), (1, 4, 2, 6, 10), (3, 16, 4, 4), (1, 1, 7, 9), (2, 4, 32, 32)])
    steps = rnd.randint(2, 7)
    kind = rnd.choice(["euler", "ancestral", "dpmpp"])
    eta, s_noise = rnd.choice([1.0, 0.8, 0.0, 0.5]), rnd.choice([1.0, 1.1, 0.9])
    g = torch.Generator().manual_seed(it)
    x0 = torch.randn(shape, generator=g) * 14.6
    sigmas = torch.cat((torch.linspace(14.6, rnd.choice([0.03, 0.5, 2.0]), steps), torch.zeros(1))) if rnd.random() < 0.8 else torch.linspace(14.6, 0.1, steps + 1)
    bank = torch.randn(2 * steps + 4, *shape, generator=g)

    def cpu_bank():
        ci = iter(bank)
        return lambda s, sn: next(ci).clone()

    def gpu_bank():
        ci = iter(bank)
        return lambda s, sn: next(ci).cuda()

    okw = dict(kw)
    if "momentum_mode" in okw: okw["mode"] = okw.pop("momentum_mode")
    try:
        cfg = orc.MomentumCfg(**okw)
        want_trace = []
        if kind == "euler":
            orc.sonar_euler(fake_model, x0.clone(), sigmas, cfg, trace=want_trace)
        elif kind == "ancestral":
            orc.sonar_euler(fake_model, x0.clone(), sigmas, cfg, ancestral=True, eta=eta, s_noise=s_noise, noise_fn=cpu_bank(), trace=want_trace)
        else:
            orc.sonar_dpmpp_sde(fake_model, x0.clone(), sigmas, cfg, eta=eta, s_noise=s_noise, noise_fn=cpu_bank(), trace=want_trace)
    except Exception as exc:  # noqa: BLE001
        print(f"[{it}] oracle {type(exc).__name__}: {str(exc)[:100]} {kind} {kw}", flush=True)
        continue
    trace = []
    cb = lambda d: trace.append(d["x"].clone())  # noqa: E731
    try:
        if kind == "euler":
            S.SonarEuler.sampler(fake_model, x0.cuda(), sigmas, {"seed": 0}, cb, True, gpu_bank(), None, dict(kw))
        elif kind == "ancestral":
            S.SonarEulerAncestral.sampler(fake_model, x0.cuda(), sigmas, {"seed": 0}, cb, True, None, dict(kw), eta, s_noise, gpu_bank())
        else:
            S.SonarDPMPPSDE.sampler(fake_model, x0.cuda(), sigmas, {"seed": 0}, cb, True, None, dict(kw), eta, s_noise, gpu_bank())
    except Exception as exc:  # noqa: BLE001
        print(f"[{it}] {type(exc).__name__}: {str(exc)[:140]}  {kind} {kw} {shape} eta {eta}", flush=True); bad += 1
        continue
    if len(trace) != len(want_trace):
        print(f"[{it}] {len(trace)} steps, oracle {len(want_trace)}  {kind} {kw}", flush=True); bad += 1
        continue
    for i, (a, b) in enumerate(zip(trace, want_trace)):
        b = b[0] if isinstance(b, (tuple, list)) else b
        ok = torch.allclose(a.cpu(), b, rtol=2e-4, atol=2e-4 * max(1.0, float(b.abs().max()))) or (bool(torch.isnan(b).all()) and bool(torch.isnan(a).all()))
        if not ok:
            print(f"[{it}] step {i}: max diff {float((a.cpu() - b).abs().max()):.3e} (peak {float(b.abs().max()):.2e})  {kind} {kw} {shape} eta {eta} s_noise {s_noise}", flush=True)
            bad += 1
            break
print(f"{iters} runs, {bad} problems")
