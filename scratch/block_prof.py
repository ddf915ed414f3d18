import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
shape = (128, 4, 256, 256)
filt = torch.rand(256, 129, device="cuda") + 0.5
for _ in range(30):
    hl.power_noise(filt, shape, seed=1, stream_id=2, plane_offset=0, factor=1.0)
torch.cuda.synchronize()
