import sys, importlib
sys.path.insert(0, '/root/repo')
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
ok, bad = [], []
x = torch.zeros(2, 4, 32, 32, device="cuda")
for t in ng.NoiseType:
    try:
        ns = nz.get_noise_sampler(t, x, 0.03, 14.6, seed=1, cpu=False, normalized=True)
        out = ns(torch.tensor(9.0), torch.tensor(6.0))
        assert out.shape == x.shape and torch.isfinite(out).all()
        ok.append(t.name.lower())
    except NotImplementedError as e:
        bad.append(t.name.lower())
    except Exception as e:
        bad.append(t.name.lower() + "!" + type(e).__name__)
print(len(ok), ok); print(len(bad), bad)
