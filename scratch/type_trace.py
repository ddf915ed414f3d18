"""kernel trace workload: TYPES=a,b,c at B latents of 4 x 128 x 128, 50 calls each (run under scratch/prof_any.sh)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
B = int(os.environ.get("B", "4"))
x = torch.zeros((B, 4, 128, 128), device="cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
for name in os.environ["TYPES"].split(","):
    ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    for _ in range(50): ns(*sig)
torch.cuda.synchronize()
