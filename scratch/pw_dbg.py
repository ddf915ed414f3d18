import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
hl = sonar_pkg.load().hip_lib; hl.load()
z = hl.power_spectrum((5, 4, 128, 128), "cuda", seed=77, stream_id=9, plane_offset=12)
print("std re", z.real.std().item(), "im mean", z.imag.mean().item(), "nan", torch.isnan(z.real).sum().item())
print("col stds", z.real.std(dim=(0,1,2))[:4].tolist(), z.real.std(dim=(0,1,2))[-3:].tolist())
print("row stds", z.real.std(dim=(0,1,3))[:3].tolist(), z.real.std(dim=(0,1,3))[62:66].tolist())
