"""Every NoiseType through get_noise_sampler on the device, replay (cpu=True) and generate (cpu=False) mode: which run, which raise."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); pkg.hip_lib.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise"); ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
x = torch.zeros(2, 4, 32, 32, device="cuda")
for t in ng.NoiseType:
    row = []
    for cpu in (True, False):
        try:
            torch.manual_seed(1)
            out = nz.get_noise_sampler(t, x, 0.03, 14.6, seed=3, cpu=cpu, normalized=True)(torch.tensor(10.0), torch.tensor(7.0))
            ok = out.is_cuda and bool(torch.isfinite(out).all())
            row.append(f"ok std={out.std().item():.3f}" if ok else "BAD")
        except Exception as exc:  # noqa: BLE001
            row.append(f"{type(exc).__name__}: {str(exc)[:70]}")
    print(f"{t.name:28s} replay: {row[0]:45s} generate: {row[1]}")
