"""-m gpu: the dual-tree complex wavelet transform (Wavelet(use_dtcwt=True): py/dtcwt.py over sonar_axis_taps_* / sonar_dtcwt_q2c_* /
c2q_*) against oracle/dtcwt_oracle.py.  pytorch_wavelets is absent, so this row's parity is UNPINNED; the oracle is the
published algorithm, tested on the CPU by perfect reconstruction, filter identities and orientation selectivity (tests/test_dtcwt_cpu.py).
Tolerances: fp64 1e-12, fp32 2e-5 of the coefficient peak."""
import importlib
import types

import numpy as np
import pytest
import torch

from oracle import dtcwt_oracle as dto
from oracle import dwt_oracle as dwo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(pkg):
    pkg.hip_lib.load()
    mods = {m: importlib.import_module(f"comfyui_sonar_amd.py.{m}") for m in ("wavelet_functions", "wavelet_cfg", "dtcwt", "noise", "noise_generation")}
    return types.SimpleNamespace(**mods, hl=pkg.hip_lib)


def _close(got, want, tol):
    want = np.asarray(want)
    torch.testing.assert_close(got.detach().cpu().double(), torch.from_numpy(want).double(), rtol=0, atol=tol * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize("shape,levels,biort", [((1, 2, 64, 64), 3, "near_sym_a"), ((2, 1, 32, 48), 2, "legall"), ((2, 4, 128, 128), 4, "near_sym_a"),
                                                ((1, 1, 36, 52), 3, "antonini"), ((1, 3, 33, 47), 2, "near_sym_a"), ((1, 1, 64, 64), 1, "legall")])
def test_forward_inverse_match_the_oracle(api, shape, levels, biort, dtype, tol):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape)
    w = api.wavelet_functions.Wavelet(level=levels, mode="symmetric", use_dtcwt=True, biort=biort)
    yl, yh = w.forward(torch.from_numpy(x).to("cuda", dtype))
    want_yl, want_yh = dto.forward(x, levels, biort)
    assert tuple(yl.shape) == want_yl.shape and [tuple(h.shape) for h in yh] == [h.shape for h in want_yh]
    _close(yl, want_yl, tol)
    for got, want in zip(yh, want_yh):
        _close(got, want, tol)
    back = w.inverse(yl, yh)
    _close(back[..., : shape[2], : shape[3]], x, 10 * tol)          # perfect reconstruction
    _close(w.inverse(yl, yh, two_step_inverse=True)[..., : shape[2], : shape[3]], x, 10 * tol)
    # the inverse alone, from the oracle's coefficients
    back = w.inverse(torch.from_numpy(want_yl).to("cuda", dtype), tuple(torch.from_numpy(h).to("cuda", dtype) for h in want_yh))
    _close(back, dto.inverse(want_yl, want_yh, biort), 10 * tol)


def test_unavailable_banks_and_modes_raise(api):
    W = api.wavelet_functions.Wavelet
    with pytest.raises(NotImplementedError):
        W(level=2, mode="symmetric", use_dtcwt=True, biort="near_sym_b")  # 13 / 19 taps: not in closed form here
    with pytest.raises(NotImplementedError):
        W(level=2, mode="symmetric", use_dtcwt=True, qshift="qshift_d")
    with pytest.raises(NotImplementedError):  # pytorch_wavelets' q-shift levels implement symmetric extension only
        W(level=3, mode="periodization", use_dtcwt=True)


def test_band_scaling_and_wavelet_filtered_noise(api):
    """wavelet_scaling over six orientations (py/wavelet_functions.py:193-216) and WaveletFilteredNoise with use_dtcwt (py/noise_generation.py:1908-2032)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 4, 32, 32)).astype(np.float32)
    scales = [[1.5, 0.5, 2.0, 1.0, 0.25, 3.0], 0.5, "fill"]
    w = api.wavelet_functions.Wavelet(level=3, mode="symmetric", use_dtcwt=True)
    yl, yh = w.forward(torch.from_numpy(x).cuda())
    yl2, yh2 = api.wavelet_functions.wavelet_scaling(yl, yh, 0.75, scales)
    oyl, oyh = dto.forward(x.astype(np.float64), 3)
    wyl, wyh = dwo.wavelet_scaling(oyl, oyh, 0.75, scales)
    _close(yl2, wyl, 2e-5)
    for got, want in zip(yh2, wyh):
        _close(got, want, 2e-5)
    gen = api.noise_generation.WaveletFilteredNoiseGenerator(torch.zeros(2, 4, 32, 32, device="cuda"), cpu=True, normalized=False, mode="symmetric", level=3,
                                                            use_dtcwt=True, yl_scale=0.75, yh_scales=scales, noise_sampler=lambda *_a: torch.from_numpy(x).cuda())
    got = gen(torch.tensor(1.0), torch.tensor(0.5))
    _close(got, dto.inverse(wyl, wyh), 5e-5)


@pytest.mark.parametrize("high_precision", [True, False])
def test_wavelet_cfg_over_the_dual_tree_transform(api, high_precision):
    """WaveletCFG with use_dtcwt (py/wavelet_cfg.py:750-791 over the six-orientation complex bands): x - IDT(blend(U, D (C - U), t))."""
    from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel

    torch.manual_seed(2)
    shape = (2, 4, 64, 64)
    cond, uncond, x = (torch.randn(shape, device="cuda") for _ in range(3))
    args = {"input": x, "cond_scale": 7.0, "cond": x - cond, "uncond": x - uncond, "cond_denoised": cond, "uncond_denoised": uncond,
            "sigma": torch.full((shape[0],), 7.0, device="cuda"), "model": FakeModel(),
            "model_options": {"transformer_options": {"sample_sigmas": SAMPLE_SIGMAS["karras12"]}}}
    diff = dict(yl_scale=2.0, yh_scales=[[3.0, 1.0, 0.5, 2.0, 1.5, 0.25], 0.5, 1.25])
    fn = api.wavelet_cfg.WaveletCFG(existing_cfg=None, rules=api.wavelet_cfg.WCFGRules.build(
        difference=diff, level=3, use_dtcwt=True, high_precision_mode=high_precision, difference_blend_mode="lerp", difference_blend_strength=0.8))
    got = fn(args)
    f = np.float64 if high_precision else np.float32
    c, u = cond.cpu().numpy().astype(f), uncond.cpu().numpy().astype(f)
    cw, uw = dto.forward(c, 3), dto.forward(u, 3)
    dw = dwo.wavelet_scaling(cw[0] - uw[0], [a - b for a, b in zip(cw[1], uw[1])], diff["yl_scale"], diff["yh_scales"])
    rw = dwo.wavelet_blend(uw, dw, yl_factor=0.8, blend="lerp")
    want = x.cpu().numpy() - dto.inverse(*rw).astype(np.float32)
    _close(got, want, 1e-6 if high_precision else 2e-5)
