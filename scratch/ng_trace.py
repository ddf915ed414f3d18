"""Phase stamps of the pyramid plane kernel (trace build: scratch/ng_build_variants.sh trace "-DSONAR_NG_TRACE"): cycles between the
stamps of thread 0, averaged over the first 256 workgroups, for the pyramid call and the Perlin + pyramid chain at a given batch."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SONAR_HIP_LIB", "scratch/bin/ngvar/lib_trace.so")
import torch, sonar_pkg
pkg = sonar_pkg.load(); hl = pkg.hip_lib; lib = hl.load()
nz = importlib.import_module("comfyui_sonar_amd.py.noise")
sig = (torch.tensor(14.6), torch.tensor(10.0))
names = ["start", "tables", "loop top", "grids drawn", "barrier", "xrows", "barrier", "tile seeded", "tiles done", "partials"]
raw = C.CDLL(os.environ["SONAR_HIP_LIB"])
for B in [int(v) for v in (sys.argv[1:] or ["64"])]:
    x = torch.zeros((B, 4, 128, 128), device="cuda")
    c = nz.CustomNoiseChain(); c.add(nz.CustomNoiseItem(0.5, noise_type="perlin")); c.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
    for tag, ns in (("pyramid", nz.get_noise_sampler("pyramid", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)),
                    ("chain", c.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=True))):
        for _ in range(50): ns(*sig)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (256 * 16))()
        assert lib.sonar_debug_ng_trace(buf) == 0 if hasattr(lib, "sonar_debug_ng_trace") else raw.sonar_debug_ng_trace(buf) == 0
        n = min(256, B * 4)
        rows = [[buf[b * 16 + k] for k in range(10)] for b in range(n)]
        deltas = [sum(r[k + 1] - r[k] for r in rows) / n for k in range(9)]
        total = sum(r[9] - r[0] for r in rows) / n
        spread = (max(r[9] for r in rows) - min(r[0] for r in rows))
        waves = [sum(buf[b * 16 + 9 + w] - buf[b * 16] for b in range(n)) / n for w in range(1, 4)]
        extra = [sum(buf[b * 16 + k] - buf[b * 16] for b in range(n)) / n for k in (13, 14, 15)]
        print(f"   cycles after start: level parameters in LDS {extra[0]:.0f} (thread 0) / {extra[1]:.0f} (part 1), part 1's generators stepped ahead {extra[2]:.0f}")
        print("   tile loop left, cycles after start: wave 0 " + f"{sum(r[8] - r[0] for r in rows) / n:.0f}, waves 1-3 " + ", ".join(f"{v:.0f}" for v in waves))
        print(f"B={B} {tag}: " + ", ".join(f"{names[k + 1]} {deltas[k]:.0f}" for k in range(9)) + f" | workgroup total {total:.0f} cycles, first start to last end {spread} (100 MHz counter: x10 ns)")
