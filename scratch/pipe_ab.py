"""A/B timing of library builds on the 512-latent power-law call, robust against drift: the builds are measured round-robin, ROUNDS
times, 100 launches per measurement, and the median and minimum per build are reported (a single pass per build, pipe_time.py,
moves by +-1 us with the box's clocks):
    MODE=ahead|final python scratch/pipe_ab.py lib_a.so lib_b.so ..."""
import ctypes as C, os, statistics, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sys.argv[1:]
planes, H, W = 2048, 128, 128
ROUNDS = int(os.environ.get("ROUNDS", "9"))
dev = torch.device("cuda")
filt = (torch.rand(H, W // 2 + 1, device=dev) + 0.5).contiguous()
out = torch.empty(planes, H, W, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ahead = os.environ.get("MODE", "ahead") == "ahead"
runs = []
for path in paths:
    lib = C.CDLL(os.path.join(ROOT, path))
    lib.sonar_power_irfft2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    lib.sonar_power_noise_ahead_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
    ws = [torch.zeros(2048, dtype=torch.float64, device=dev) for _ in range(2)]
    state = {"k": 0}
    def launch(lib=lib, ws=ws, state=state):
        if ahead:
            k = state["k"]
            state["k"] += 1
            assert lib.sonar_power_noise_ahead_f32(filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2 + k, 0, 4, 1.0, 2.5, ws[k & 1].data_ptr(), int(k > 0), 3 + k, ws[(k + 1) & 1].data_ptr(), stream) == 0
        else:
            assert lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), planes, H, W, 1, 2, 0, 4, None, stream) == 0
    for _ in range(300):
        launch()
    runs.append((os.path.basename(path), launch, []))
torch.cuda.synchronize()
for rnd in range(ROUNDS):
    for name, launch, times in runs:
        for _ in range(20):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            launch()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 10)
for name, _, times in runs:
    print(f"{name:24s} median {statistics.median(times):6.2f}  min {min(times):6.2f}  max {max(times):6.2f} us per launch ({'look-ahead' if ahead else 'final pass alone'})", flush=True)
