"""PMC calibration: kernels with a known HBM byte count at the bench's size (512 x 4 x 128 x 128 fp32 = 134 217 728 B).
Run under the same rocprofv3 --pmc passes as bench.py; WRITE_SIZE / FETCH_SIZE per dispatch are compared with:
  stream_fill_kernel (philox_normal): 0 read, 134 MB written
  stats_kernel:                      134 MB read, ~0 written
  scale_noise_kernel:                134 MB read, 134 MB written"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sonar_pkg

hl = sonar_pkg.load().hip_lib
hl.load()
shape = (512, 4, 128, 128)
for i in range(6):
    x = hl.philox_normal(shape, "cuda", 1, i)
    p = hl.stats(x)
    hl.scale_noise_(x, 0.5, True, p)
torch.cuda.synchronize()
print("calib done")
