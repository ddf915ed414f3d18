"""cfg5 on one rank's shard (128 Flux latents): scheduled power + Perlin + Brownian chain, SonarDPMPPSDE with momentum -- for a kernel trace."""
import importlib, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise"); nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sonar = importlib.import_module("comfyui_sonar_amd.py.sonar")
for _ in range(3):
    print("cfg5 step ms", b.cfg5_shard_step_ms(torch.device("cuda", 0), hl, pn, nz, sonar))
