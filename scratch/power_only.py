import sys; sys.path.insert(0, '.')
import torch, sonar_pkg
hl = sonar_pkg.load().hip_lib; hl.load()
B = 512
shape = (B, 4, 128, 128)
filt = torch.rand(128, 65, device='cuda') + 0.5
part = hl.new_partials('cuda')
mode = sys.argv[1] if len(sys.argv) > 1 else 'gen'
z = torch.randn(B, 4, 128, 65, dtype=torch.complex64, device='cuda') if mode == 'replay' else None
for _ in range(10):
    hl.power_irfft2(z, filt, shape, seed=1, partials=part)
torch.cuda.synchronize()
