import importlib, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; hl.load()
pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise"); ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation"); nz = importlib.import_module("comfyui_sonar_amd.py.noise")
real = b.host_and_event_us
def dbg(fn, *a, **k):
    r = real(fn, *a, **k)
    cell = fn.__closure__[0].cell_contents if fn.__closure__ else None
    pl = cell if isinstance(cell, hl.Planned) else getattr(cell, "_planned", None)
    print("host/gpu", [round(v, 1) for v in r], type(cell).__name__, None if pl is None else (pl.plan is not None, pl.reason, pl.calls, pl.attempts, pl.plan.runs if pl.plan else None), file=sys.stderr, flush=True)
    return r
b.host_and_event_us = dbg
torch.manual_seed(0)
x = torch.zeros((512, 4, 128, 128), device="cuda")
sig = (torch.tensor(14.6), torch.tensor(10.0))
b.secondary_rows(torch.device("cuda", 0), hl, pn, ng, nz, x, sig)
