"""A/B of store hints in the elementwise kernels (SONAR_HIP_LIB picks the build): the Euler momentum step at 512 / 64 / 4 latents and cfg5's shard step."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
sonar = importlib.import_module("comfyui_sonar_amd.py.sonar")


def timed(fn, n=100, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (512, 64, 4):
    x = torch.randn(B, 4, 128, 128, device="cuda"); den = torch.randn_like(x)
    sb = sonar.SonarBase(sonar.SonarBase.get_config(None, {}))
    sb.momentum_step(0, x, den, torch.tensor(10.0), torch.tensor(8.0))
    print(f"euler momentum step, {B} latents: {timed(lambda: sb.momentum_step(1, x, den, torch.tensor(8.0), torch.tensor(6.0))):7.1f} us", flush=True)
    t = torch.randn(B, 4, 128, 128, device="cuda"); p = hl.stats(t)
    print(f"scale_noise, {B} latents: {timed(lambda: hl.scale_noise_(t, 0.999, False, None)):7.1f} us", flush=True)
import bench
try:
    pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise"); nz = importlib.import_module("comfyui_sonar_amd.py.noise")
    for _ in range(3): print(f"cfg5 shard step: {bench.cfg5_shard_step_ms(torch.device('cuda'), hl, pn, nz, sonar):.3f} ms", flush=True)
except Exception as e:
    print("cfg5:", repr(e)[:200])
