import sys, time, math
sys.path.insert(0, '.')
import torch, sonar_pkg
hl = sonar_pkg.load().hip_lib
hl.load()
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us
B = 512
shape = (B, 4, 128, 128)
n = B * 4 * 128 * 128
filt = torch.rand(128, 65, device='cuda') + 0.5
part = hl.new_partials('cuda')
x = torch.randn(shape, device='cuda')
res = {}
res['stats'] = timeit(lambda: hl.stats(x, part))
res['scale_apply'] = timeit(lambda: hl.scale_noise_(x, 1.0, True, part))
res['philox_normal'] = timeit(lambda: hl.philox_normal(shape, 'cuda', 1, 0, out=x))
res['philox_normal+stats'] = timeit(lambda: hl.philox_normal(shape, 'cuda', 1, 0, partials=part, out=x))
res['philox_uniform'] = timeit(lambda: hl.philox_uniform(shape, 'cuda', 1, 0, out=x))
terms = torch.randn(2, 4, 128, 128, device='cuda')
res['perlin_generate+stats'] = timeit(lambda: hl.perlin_generate(shape, terms, 2.0, 1, 0, 0, part))
res['power_gen+stats'] = timeit(lambda: hl.power_irfft2(None, filt, shape, seed=1, partials=part))
res['power_gen'] = timeit(lambda: hl.power_irfft2(None, filt, shape, seed=1))
z = torch.randn(B, 4, 128, 65, dtype=torch.complex64, device='cuda')
res['power_replay'] = timeit(lambda: hl.power_irfft2(z, filt, shape))
y = torch.empty_like(x)
res['torch_copy'] = timeit(lambda: y.copy_(x))
res['torch_irfft2'] = timeit(lambda: torch.fft.irfft2(z, s=(128,128), norm='ortho'))
res['torch_randn'] = timeit(lambda: torch.randn(shape, device='cuda'))
h = torch.randn(shape, device='cuda'); den = torch.randn(shape, device='cuda')
cfg = hl.MomentumCfg(); cfg.momentum=0.95; cfg.hist_ratio=0.75; cfg.hist_scale=1.0; cfg.md_scale=1.0; cfg.mode=1; cfg.use_momentum=1; cfg.update_hist=1
xo = torch.empty_like(x); ho = torch.empty_like(x)
res['momentum_euler'] = timeit(lambda: hl.momentum_euler(x, den, h, cfg, 3.0, -0.5, x_out=xo, h_out=ho))
for k, v in res.items():
    print(f"{k:28s} {v:9.1f} us   {n*4/v/1e6:8.2f} TB/s-per-4N")
res2 = {}
res2['perlin_noise(fused)'] = timeit(lambda: hl.perlin_noise(shape, terms, 2.0, 1, 0, 0, 1.0))
res2['power_noise(fused)'] = timeit(lambda: hl.power_noise(filt, shape, seed=1, stream_id=0, plane_offset=0, factor=1.0))
for k, v in res2.items():
    print(f"{k:28s} {v:9.1f} us   12N-equiv {n*12/v/1e6:8.2f} TB/s")
