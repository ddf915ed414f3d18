"""Non-contiguous and half-precision latents through the drop-in entry points (noise samplers, the three samplers, WaveletCFG): the
reference takes any strides / float dtype; these must run and agree with the same call on a contiguous fp32 copy."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
from tests.golden import wavelet_cases as wc
pkg = sonar_pkg.load(); pkg.hip_lib.load()
S = importlib.import_module("comfyui_sonar_amd.py.sonar"); N = importlib.import_module("comfyui_sonar_amd.py.noise")
W = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
bad = 0


def fake_model(x, sigma, **_kw):
    s = sigma.reshape(-1, *([1] * (x.ndim - 1))).to(x.dtype)
    return x * 0.5 + torch.tanh(x) * (0.1 * s / (1.0 + s))


def views(t):
    yield "contiguous", t
    yield "channels_last", t.contiguous(memory_format=torch.channels_last)
    yield "transposed", t.transpose(2, 3).contiguous().transpose(2, 3)
    big = torch.zeros(t.shape[0], t.shape[1], t.shape[2] * 2, t.shape[3] + 3, device=t.device, dtype=t.dtype)
    big[:, :, ::2, 1:-2] = t
    yield "strided_slice", big[:, :, ::2, 1:-2]


g = torch.Generator().manual_seed(1)
x0 = (torch.randn(2, 4, 16, 24, generator=g) * 10).cuda()
sigmas = torch.cat((torch.linspace(14.6, 0.03, 5), torch.zeros(1)))
for kind in ("euler", "ancestral", "dpmpp"):
    outs = {}
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        for name, xv in views(x0.to(dtype)):
            try:
                torch.manual_seed(3)
                ns = N.get_noise_sampler("gaussian", xv, 0.03, 14.6, seed=3, cpu=True, normalized=True)
                if kind == "euler":
                    out = S.SonarEuler.sampler(fake_model, xv.clone() if name == "contiguous" else xv, sigmas, {"seed": 3}, None, True, ns, None, {})
                elif kind == "ancestral":
                    out = S.SonarEulerAncestral.sampler(fake_model, xv, sigmas, {"seed": 3}, None, True, None, {}, 0.8, 1.1, ns)
                else:
                    out = S.SonarDPMPPSDE.sampler(fake_model, xv, sigmas, {"seed": 3}, None, True, None, {}, 0.9, 1.05, ns)
                outs[(dtype, name)] = out.float()
                base = outs[(dtype, "contiguous")]
                tol = 1e-5 if dtype == torch.float32 else 0.0
                if out.dtype != dtype or not torch.allclose(out.float(), base, rtol=tol, atol=tol * 10):
                    print(f"{kind} {dtype} {name}: dtype {out.dtype}, max diff vs contiguous {float((out.float() - base).abs().max()):.3e}"); bad += 1
            except Exception as exc:  # noqa: BLE001
                print(f"{kind} {dtype} {name}: {type(exc).__name__}: {str(exc)[:160]}"); bad += 1
    for dtype in (torch.float16, torch.bfloat16):
        if (dtype, "contiguous") in outs:
            d = float((outs[(dtype, "contiguous")] - outs[(torch.float32, "contiguous")]).abs().max())
            print(f"{kind} {dtype} vs fp32: max diff {d:.3e}")
# noise samplers on non-contiguous / half latents (generate mode)
for t in ("gaussian", "perlin", "pyramid", "brownian", "onef_pinkish", "studentt"):
    for dtype in (torch.float32, torch.float16):
        for name, xv in views(torch.zeros(2, 4, 16, 24, device="cuda", dtype=dtype)):
            try:
                torch.manual_seed(5)
                out = N.get_noise_sampler(t, xv, 0.03, 14.6, seed=5, cpu=False, normalized=True)(torch.tensor(10.0), torch.tensor(7.0))
                if not bool(torch.isfinite(out.float()).all()) or tuple(out.shape) != tuple(xv.shape):
                    print(f"noise {t} {dtype} {name}: bad output {tuple(out.shape)} {out.dtype}"); bad += 1
            except Exception as exc:  # noqa: BLE001
                print(f"noise {t} {dtype} {name}: {type(exc).__name__}: {str(exc)[:160]}"); bad += 1
# WaveletCFG on non-contiguous / half inputs
case = dict(shape=(2, 4, 16, 24), sigma=7.0, params=dict(difference=dict(yl_scale=5.0, yh_scales=3.0), level=2))
import json
for dtype in (torch.float32, torch.float16):
    base = None
    for name, _ in views(torch.zeros(2, 4, 16, 24, device="cuda")):
        args = wc.wcfg_inputs(case, "layout")
        conv = {}
        for k, v in args.items():
            if isinstance(v, torch.Tensor) and v.ndim == 4:
                v = dict(views(v.cuda().to(dtype)))[name]
            elif isinstance(v, torch.Tensor):
                v = v.cuda()
            conv[k] = v
        conv["model"] = wc.FakeModel(); conv["model_options"] = {"transformer_options": {"sample_sigmas": wc.SAMPLE_SIGMAS["karras12"]}}
        try:
            fn = W.WaveletCFG(existing_cfg=None, rules=W.WCFGRules.build(**json.loads(json.dumps(case["params"]))))
            out = fn(conv)
            base = out.float() if base is None else base
            if not torch.allclose(out.float(), base, rtol=1e-5, atol=1e-5):
                print(f"wcfg {dtype} {name}: max diff vs contiguous {float((out.float() - base).abs().max()):.3e}"); bad += 1
        except Exception as exc:  # noqa: BLE001
            print(f"wcfg {dtype} {name}: {type(exc).__name__}: {str(exc)[:160]}"); bad += 1
print(f"{bad} problems")
