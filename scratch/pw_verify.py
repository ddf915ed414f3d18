"""Generated 128 x 128 power-law planes of any library build against torch.fft.irfft2 of the spectrum the same build dumps, and the
pipelined kernel against the phase-serial one (must agree bit for bit):
    python scratch/pw_verify.py scratch/bin/pwvar/lib_a.so ...      (default: the product library)"""
import ctypes as C, os, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["comfyui-sonar_amd/libsonar_hip.so"]
H = W = 128
dev = torch.device("cuda")
stream = torch.cuda.current_stream().cuda_stream
for path in libs:
    lib = C.CDLL(os.path.join(ROOT, path))
    lib.sonar_power_irfft2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    lib.sonar_power_spectrum_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p]
    lib.sonar_power_noise_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.sonar_power_pipeline.argtypes = [C.c_int]
    lib.sonar_last_error.restype = C.c_char_p
    ok = True
    for planes, group, signed in ((2048, 4, False), (2048, 4, True), (516, 4, False), (771, 1, True), (260, 4, False)):
        torch.manual_seed(planes)
        filt = (torch.rand(H, W // 2 + 1, device=dev) + 0.25).contiguous()
        if signed:
            filt = filt * torch.where(torch.rand_like(filt) < 0.3, -1.0, 1.0)
            filt[5, 7] = 0.0
        z = torch.empty(planes, H, W // 2 + 1, 2, device=dev)
        assert lib.sonar_power_spectrum_f32(z.data_ptr(), planes, H, W, 7, 11, 0, group, stream) == 0, lib.sonar_last_error()
        want = torch.fft.irfft2(torch.view_as_complex(z) * filt, s=(H, W), norm="ortho")
        outs = []
        for pipe in (1, 0):
            lib.sonar_power_pipeline(pipe)
            out = torch.full((planes, H, W), float("nan"), device=dev)
            assert lib.sonar_power_irfft2_f32(None, filt.data_ptr(), out.data_ptr(), planes, H, W, 7, 11, 0, group, None, stream) == 0, lib.sonar_last_error()
            outs.append(out)
        lib.sonar_power_pipeline(1)
        err = (outs[0] - want).abs().max().item()
        same = torch.equal(outs[0], outs[1])
        # normalised call: statistics by Parseval against the output's own
        ws = torch.zeros(2048 * 2, dtype=torch.float64, device=dev)
        outn = torch.empty((planes, H, W), device=dev)
        assert lib.sonar_power_noise_f32(filt.data_ptr(), outn.data_ptr(), planes, H, W, 7, 11, 0, group, 1.0, 2.5, ws.data_ptr(), stream) == 0, lib.sonar_last_error()
        sd, mean = outn.double().std().item(), outn.double().mean().item()
        rel = ((outn - outs[0] / outs[0].double().std().float()).abs().max() ).item()
        good = err < 2e-5 and same and abs(sd - 1.0) < 2e-6 and rel < 1e-4
        ok = ok and good
        print(f"{os.path.basename(path):22s} planes {planes:5d} group {group} signed {int(signed)}: max|gen - irfft2(dump x filter)| {err:.2e}, pipe == serial {same}, "
              f"normalised std {sd:.7f} mean {mean:.1e}, max|norm - gen / std| {rel:.1e}  {'ok' if good else 'FAIL'}", flush=True)
    print(f"{os.path.basename(path)}: {'all ok' if ok else 'FAILED'}", flush=True)
