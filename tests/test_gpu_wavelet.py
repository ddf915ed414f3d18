"""-m gpu: 2-D DWT / IDWT kernels against PyWavelets 1.1.1 golden vectors, and WaveletCFG at SDXL sizes against the numpy oracle
(oracle/dwt_oracle.py, itself pinned to the reference's own outputs by tests/test_wavelet_oracle_cpu.py; the reference-run
fixtures proper are compared in tests/test_gpu_wavelet_golden.py).

Tolerance: fp64 path 1e-11 (same arithmetic, different summation order); fp32 path rtol 2e-5 / atol 2e-5 on
O(1) data (taps rounded to fp32, 8-tap dot products over up to 5 levels).  WaveletCFG outputs: absolute, relative to the
output's peak (the placeholder rule scales the approximation band by 5 and the detail bands by 3, so outputs reach O(20) and
the fp32 transform's 2e-5-class error grows with them): 2e-6 x peak for fp64 rules, 5e-5 x peak for fp32 rules."""
import math
import importlib
import types

import numpy as np
import pytest
import torch

from oracle import dwt_oracle as dwo
from tests.conftest import GOLDEN
from tests import wavelet_helpers as wh

pytestmark = pytest.mark.gpu

G = np.load(f"{GOLDEN}/dwt.npz", allow_pickle=False)
TAGS = sorted({k.split("__")[0] for k in G.files})


@pytest.fixture(scope="module")
def api(pkg):
    pkg.hip_lib.load()
    return types.SimpleNamespace(hl=pkg.hip_lib, wf=importlib.import_module("comfyui_sonar_amd.py.wavelet_functions"),
                                 wc=importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg"))


def tol(dtype):
    return dict(rtol=1e-11, atol=1e-11) if dtype == torch.float64 else dict(rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("tag", TAGS)
def test_forward_and_inverse_match_pywt(api, tag, dtype):
    wave, mode, level = G[f"{tag}__meta"]
    level = int(level)
    w = api.wf.Wavelet(wave=str(wave), level=level, mode=str(mode), use_1d_dwt=tag.startswith("d1_"))  # d1_*: pywt.wavedec vectors
    x = torch.from_numpy(G[f"{tag}__x"]).to(dtype).cuda()
    yl, yh = w.forward(x)
    torch.testing.assert_close(yl.cpu().double(), torch.from_numpy(G[f"{tag}__yl"]), **tol(dtype))
    assert len(yh) == level
    for j in range(level):
        want = torch.from_numpy(G[f"{tag}__yh{j}"])
        assert tuple(yh[j].shape) == tuple(want.shape)
        torch.testing.assert_close(yh[j].cpu().double(), want, **tol(dtype))
    gl = torch.from_numpy(G[f"{tag}__yl"]).to(dtype).cuda()
    gh = [torch.from_numpy(G[f"{tag}__yh{j}"]).to(dtype).cuda() for j in range(level)]
    rec = w.inverse(gl, gh)
    torch.testing.assert_close(rec.cpu().double(), torch.from_numpy(G[f"{tag}__rec"]), **tol(dtype))
    two = w.inverse(gl, gh, two_step_inverse=True)
    torch.testing.assert_close(two.cpu().double(), torch.from_numpy(G[f"{tag}__rec"]), **tol(dtype))


@pytest.mark.parametrize("wave,mode,level", [("db4", "symmetric", 5), ("haar", "periodization", 3), ("sym8", "reflect", 4), ("coif3", "zero", 2)])
def test_perfect_reconstruction_full_batch(api, wave, mode, level):
    """cfg4 size: 256 x 4 x 128 x 128 (fp32) — IDWT(DWT(x)) == x."""
    torch.manual_seed(0)
    x = torch.randn(256, 4, 128, 128, device="cuda")
    w = api.wf.Wavelet(wave=wave, level=level, mode=mode)
    yl, yh = w.forward(x)
    rec = w.inverse(yl, yh)[..., :128, :128]
    assert (rec - x).abs().max().item() < 5e-5
    # linearity: DWT(a x) = a DWT(x)
    yl2, _ = w.forward(x * 2)
    torch.testing.assert_close(yl2, yl * 2, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("wave,mode,level", [("db4", "symmetric", 5), ("haar", "periodization", 3), ("sym8", "periodic", 4)])
def test_1d_transform_full_size(api, wave, mode, level):
    """use_1d_dwt on flattened SDXL latents ([64, 4, 16384], fp32 and fp64): perfect reconstruction, linearity, band lengths."""
    torch.manual_seed(0)
    x = torch.randn(64, 4, 128 * 128, device="cuda")
    w = api.wf.Wavelet(wave=wave, level=level, mode=mode, use_1d_dwt=True)
    yl, yh = w.forward(x)
    n, flen = 128 * 128, len(w.dec_lo)
    for band in yh:
        n = api.hl.dwt_out_len(n, flen, mode)
        assert tuple(band.shape) == (64, 4, n)
    assert tuple(yl.shape) == (64, 4, n)
    rec = w.inverse(yl, yh)[..., : 128 * 128]
    assert (rec - x).abs().max().item() < 5e-5
    yl2, _ = w.forward(x * 2)
    torch.testing.assert_close(yl2, yl * 2, rtol=1e-6, atol=1e-6)
    xd = x[:4].double()
    rec64 = w.inverse(*w.forward(xd))[..., : 128 * 128]
    assert (rec64 - xd).abs().max().item() < 1e-11
    with pytest.raises(api.hl.SonarHipError):
        w.forward(x.reshape(64, 4, 128, 128))


def test_longest_filter_matches_oracle(api):
    """dmey (62 taps, only approximately orthogonal: no PR property) — compare with the oracle directly."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 2, 70, 66))
    w = api.wf.Wavelet(wave="dmey", level=2, mode="symmetric")
    yl, yh = w.forward(torch.from_numpy(x).cuda())
    ol, oh = dwo.wavedec2(x, "dmey", "symmetric", 2)
    np.testing.assert_allclose(yl.cpu().numpy(), ol, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(yh[1].cpu().numpy(), oh[1], rtol=1e-10, atol=1e-10)
    rec = w.inverse(yl, yh).cpu().numpy()
    np.testing.assert_allclose(rec, dwo.waverec2(ol, oh, "dmey", "symmetric"), rtol=1e-10, atol=1e-10)


def test_wavelist_and_errors(api):
    names = api.wf.Wavelet.wavelist()
    assert len(names) == 106 and {"haar", "db4", "sym5", "bior2.2", "coif1", "dmey"} <= set(names)
    with pytest.raises(ValueError):
        api.wf.Wavelet(wave="nope")
    with pytest.raises(NotImplementedError):
        api.wf.Wavelet(use_dtcwt=True, biort="near_sym_b")  # near_sym_a / legall / antonini / qshift_a are built in (tests/test_gpu_dtcwt.py)


from tests.golden.wavelet_cases import SAMPLE_SIGMAS, FakeModel  # noqa: E402

# ComfyUI hands the sampler's sigma schedule to every cfg function; the reference cannot build its percentages without it
MODEL_OPTIONS = {"transformer_options": {"sample_sigmas": SAMPLE_SIGMAS["karras12"]}}


def wcfg_close(out, want, high_precision):
    want = want.float() if isinstance(want, torch.Tensor) else torch.from_numpy(np.asarray(want, dtype=np.float32))
    torch.testing.assert_close(out.cpu(), want, rtol=0, atol=(2e-6 if high_precision else 5e-5) * max(1.0, float(want.abs().max())))


PLACEHOLDER_RULE = dict(difference=dict(yl_scale=5.0, yh_scales=3.0))  # the node's placeholder YAML (py/nodes/misc.py)


@pytest.mark.parametrize("high_precision", [True, False])
@pytest.mark.parametrize("extra", [{}, {"difference_blend_strength": 0.7}, {"wave": "haar", "level": 3, "padding_mode": "periodization"},
                                   {"difference": {"yl_scale": 2.0, "yh_scales": [1.5, [2.0, 0.5], "fill", 0.25]}, "difference_blend_mode": "lerp",
                                    "difference_blend_strength": 0.8}])
def test_wavelet_cfg_matches_oracle(api, high_precision, extra):
    torch.manual_seed(1)
    shape = (2, 4, 128, 128)
    cond, uncond, x = (torch.randn(shape) for _ in range(3))
    params = dict(PLACEHOLDER_RULE, high_precision_mode=high_precision)
    params.update(extra)
    rules = api.wc.WCFGRules.build(**params)
    fn = api.wc.WaveletCFG(existing_cfg=None, rules=rules)
    args = {"input": x.cuda(), "cond_scale": 7.0, "cond": (x - cond).cuda(), "uncond": (x - uncond).cuda(), "cond_denoised": cond.cuda(),
            "uncond_denoised": uncond.cuda(), "sigma": torch.full((2,), 7.0, device="cuda"), "model": FakeModel(), "model_options": MODEL_OPTIONS}
    out = fn(args)
    assert out.is_contiguous() and out.shape == x.shape and out.dtype == torch.float32
    rule = rules[0]
    ws = rule.wavelet
    diff = params["difference"]
    dt = np.float64 if high_precision else np.float32
    res = dwo.wavelet_cfg(cond.numpy().astype(dt), uncond.numpy().astype(dt), ws.wave, ws.padding_mode, ws.level,
                          diff_scales=(diff["yl_scale"], diff["yh_scales"]), strength=params.get("difference_blend_strength", 1.0),
                          blend=params.get("difference_blend_mode", "inject"))
    want = x - torch.from_numpy(res[..., :128, :128].astype(np.float32))
    wcfg_close(out, want, high_precision)


@pytest.mark.parametrize("high_precision", [True, False])
def test_wavelet_cfg_1d_mode_matches_oracle(api, high_precision):
    """py/wavelet_cfg.py:713-715,737-738: use_1d_dwt flattens the latents to [B, C, H*W]; one detail band per level."""
    torch.manual_seed(4)
    shape = (2, 4, 24, 20)
    cond, uncond, x = (torch.randn(shape) for _ in range(3))
    params = dict(difference=dict(yl_scale=2.0, yh_scales=[3.0, 0.5, "fill"]), high_precision_mode=high_precision, use_1d_dwt=True, level=3)
    rules = api.wc.WCFGRules.build(**params)
    fn = api.wc.WaveletCFG(existing_cfg=None, rules=rules)
    args = {"input": x.cuda(), "cond_scale": 7.0, "cond": (x - cond).cuda(), "uncond": (x - uncond).cuda(), "cond_denoised": cond.cuda(),
            "uncond_denoised": uncond.cuda(), "sigma": torch.full((2,), 7.0, device="cuda"), "model": FakeModel(), "model_options": MODEL_OPTIONS}
    out = fn(args)
    assert out.is_contiguous() and out.shape == x.shape and out.dtype == torch.float32
    ws = rules[0].wavelet
    dt = np.float64 if high_precision else np.float32
    flat = lambda t: t.numpy().astype(dt).reshape(2, 4, -1)  # noqa: E731
    res = dwo.wavelet_cfg(flat(cond), flat(uncond), ws.wave, ws.padding_mode, ws.level, diff_scales=(2.0, [3.0, 0.5, "fill"]), one_d=True)
    want = x - torch.from_numpy(res[..., : 24 * 20].reshape(shape).astype(np.float32))
    wcfg_close(out, want, high_precision)
    # a 3-D latent needs the 1-D mode (py/wavelet_cfg.py:681-682)
    rules2 = api.wc.WCFGRules.build(difference=dict(yl_scale=2.0))
    with pytest.raises(RuntimeError):
        api.wc.WaveletCFG(existing_cfg=None, rules=rules2)({**args, "input": x.cuda().flatten(2), "cond_denoised": cond.cuda().flatten(2),
                                                            "uncond_denoised": uncond.cuda().flatten(2)})


def test_wavelet_cfg_rule_window_and_blend(api):
    torch.manual_seed(2)
    shape = (1, 4, 64, 64)
    cond, uncond, x = (torch.randn(shape, device="cuda") for _ in range(3))
    args = {"input": x, "cond_scale": 5.0, "cond": x - cond, "uncond": x - uncond, "cond_denoised": cond, "uncond_denoised": uncond,
            "sigma": torch.full((1,), 3.0, device="cuda"), "model": FakeModel(), "model_options": MODEL_OPTIONS}
    plain = x - (uncond + (cond - uncond) * 5.0)
    # outside the rule's sigma window -> plain CFG
    fn = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(start_sigma=14.0, end_sigma=5.0, **PLACEHOLDER_RULE))
    torch.testing.assert_close(fn(args), plain, rtol=1e-5, atol=1e-5)
    # blend_strength 0 -> plain CFG; 0.5 -> halfway between plain CFG and the wavelet result
    fn0 = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(blend_strength=0.0, **PLACEHOLDER_RULE))
    torch.testing.assert_close(fn0(args), plain, rtol=1e-5, atol=1e-5)
    full = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(**PLACEHOLDER_RULE))(args)
    half = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(blend_strength=0.5, **PLACEHOLDER_RULE))(args)
    torch.testing.assert_close(half, (plain + full) / 2, rtol=1e-4, atol=1e-4)
    # identity scales: wavelet CFG with all scales 1 and strength s == uncond + s (cond - uncond)
    ident = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(difference=dict(yl_scale=1.0, yh_scales=1.0), difference_blend_strength=5.0))
    torch.testing.assert_close(ident(args), plain, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("high_precision", [True, False])
@pytest.mark.parametrize("extra", [{}, {"wave": "haar", "level": 3, "padding_mode": "periodization"}, {"wave": "sym5", "level": 2, "padding_mode": "zero"},
                                   {"wave": "bior2.2", "level": 4, "padding_mode": "reflect", "difference_blend_mode": "lerp", "difference_blend_strength": 0.3,
                                    "difference": {"yl_scale": 0.5, "yh_scales": [[1.0, 2.0, 3.0], 0.25, "fill"]}}])
def test_wavelet_cfg_fused_equals_per_pass_path(api, monkeypatch, high_precision, extra):
    """The LDS-staged 2 * level launch path (fused multiply-adds, H-first analysis) against the per-pass kernels:
    same transform, different rounding -> fp64 1e-12, fp32 2e-5 on O(1..10) data."""
    torch.manual_seed(4)
    shape = (3, 4, 96, 80)  # not a power of two: odd coefficient sizes on the way down
    cond, uncond, x = (torch.randn(shape, device="cuda") for _ in range(3))
    params = dict(PLACEHOLDER_RULE, high_precision_mode=high_precision)
    params.update(extra)
    args = {"input": x, "cond_scale": 7.0, "cond": x - cond, "uncond": x - uncond, "cond_denoised": cond, "uncond_denoised": uncond,
            "sigma": torch.full((3,), 7.0, device="cuda"), "model": FakeModel(), "model_options": MODEL_OPTIONS}
    fn = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(**params))
    monkeypatch.setattr(api.wc.WaveletCFG, "_lowpass_launch", classmethod(lambda cls, **_k: None))  # this test is about the band kernels
    monkeypatch.setattr(api.wc.WaveletCFG, "single_launch_bands", False)  # ... of the tile route (fp32's default is the single-launch kernel)
    calls = []
    real = api.hl.FusedCall.__call__
    monkeypatch.setattr(api.hl.FusedCall, "__call__", lambda self, *a: calls.append(1) or real(self, *a))
    fused = fn(args)
    assert calls, "the fused entry point was not used"
    # fp64 mode stores level 1's detail bands in fp32 (sonar_wcfg_hi_storage, default on): with fp64 storage the fused route IS the
    # per-pass route to 1e-11; with fp32 storage to a fraction of the fp32 result's own rounding
    lib = api.hl.load()
    assert lib.sonar_wcfg_hi_storage(0) == 1
    try:
        fused_hi64 = fn(args)
    finally:
        lib.sonar_wcfg_hi_storage(1)
    monkeypatch.setattr(api.wc.WaveletCFG, "wavelet_cfg_fused", classmethod(lambda cls, **_k: None))
    per_pass = fn(args)
    torch.testing.assert_close(fused_hi64, per_pass, rtol=1e-5, atol=(1e-11 if high_precision else 4e-5))
    torch.testing.assert_close(fused, per_pass, rtol=1e-5, atol=(2e-6 if high_precision else 4e-5))
    if not high_precision:
        assert torch.equal(fused, fused_hi64)  # the switch concerns fp64 mode only


def test_wavelet_cfg_fused_full_batch_identity(api):
    """cfg4 size (256 x 4 x 128 x 128): unit scales and strength 1 give x - cond (perfect reconstruction), in 10 launches."""
    torch.manual_seed(5)
    shape = (256, 4, 128, 128)
    cond, uncond, x = (torch.randn(shape, device="cuda") for _ in range(3))
    args = {"input": x, "cond_scale": 7.0, "cond": x - cond, "uncond": x - uncond, "cond_denoised": cond, "uncond_denoised": uncond,
            "sigma": torch.full((256,), 7.0, device="cuda"), "model": FakeModel(), "model_options": MODEL_OPTIONS}
    fn = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(difference=dict(yl_scale=1.0, yh_scales=1.0), high_precision_mode=False))
    torch.testing.assert_close(fn(args), x - cond, rtol=0, atol=2e-5)


@pytest.mark.parametrize("high_precision", [True, False])
@pytest.mark.parametrize("wave,mode,level,shape", [
    ("haar", "zero", 3, (2, 3, 40, 56)), ("db2", "symmetric", 4, (1, 4, 37, 50)), ("db4", "reflect", 3, (2, 2, 64, 48)),
    ("sym5", "periodization", 2, (1, 4, 50, 38)), ("coif2", "periodic", 2, (1, 2, 72, 64)), ("db8", "constant", 2, (1, 2, 96, 80)),
    ("bior2.2", "symmetric", 3, (1, 4, 33, 47)), ("db10", "symmetric", 1, (1, 2, 64, 64)), ("db4", "symmetric", 5, (2, 4, 128, 128))])
def test_lowpass_path_equals_band_path(api, monkeypatch, wave, mode, level, shape, high_precision):
    """The one-launch low-pass pyramid (sonar_wcfg_lowpass_*) against the band-by-band kernels (sonar_wcfg_fused_*) on rules with one
    scale per level: every padding mode, odd sizes, filter lengths 2..20, lerp / inject / subtract blends.  Same transform, different
    algebra: fp64 arithmetic 3e-7 of the output peak (the outputs are fp32 tensors: a last-bit difference of the final rounding), fp32 5e-5."""
    torch.manual_seed(7)
    cond, uncond, x = (torch.randn(shape, device="cuda") for _ in range(3))
    args = {"input": x, "cond_scale": 7.0, "cond": x - cond, "uncond": x - uncond, "cond_denoised": cond, "uncond_denoised": uncond,
            "sigma": torch.full((shape[0],), 7.0, device="cuda"), "model": FakeModel(), "model_options": MODEL_OPTIONS}
    for blend, strength in (("inject", 1.0), ("lerp", 0.35), ("lerp", 0.8), ("subtract_b", 0.6)):
        params = dict(difference=dict(yl_scale=2.5, yh_scales=[3.0, 0.5, "fill", 1.5][: level + 1]), wave=wave, level=level, padding_mode=mode,
                      high_precision_mode=high_precision, difference_blend_mode=blend, difference_blend_strength=strength)
        fn = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(**params))
        calls = []
        real = api.hl.wcfg_lowpass_plan
        with monkeypatch.context() as m:
            m.setattr(api.hl, "wcfg_lowpass_plan", lambda *a, **k: wh.counted_launch(real(*a, **k), calls))
            low = fn(args)
        assert calls, "the low-pass entry point was not used"
        with monkeypatch.context() as m:
            m.setattr(api.wc.WaveletCFG, "_lowpass_launch", classmethod(lambda cls, **_k: None))
            m.setattr(api.wc, "_reconstructs", lambda w: False)  # cond and uncond both transformed (the general route)
            bands = fn(args)
        peak = float(bands.abs().max())
        torch.testing.assert_close(low, bands, rtol=0, atol=(3e-7 if high_precision else 5e-5) * max(1.0, peak))


@pytest.mark.parametrize("high_precision", [True, False])
@pytest.mark.parametrize("wave,mode,level,shape", [
    ("haar", "zero", 3, (2, 3, 40, 56)), ("db2", "symmetric", 4, (1, 4, 37, 50)), ("db4", "reflect", 3, (2, 2, 64, 48)),
    ("sym5", "periodization", 2, (1, 4, 50, 38)), ("coif2", "periodic", 2, (1, 2, 72, 64)), ("db8", "constant", 2, (1, 2, 96, 80)),
    ("bior2.2", "symmetric", 3, (1, 4, 33, 47)), ("db10", "symmetric", 1, (1, 2, 64, 64)), ("db4", "symmetric", 5, (2, 4, 128, 128)),
    ("db4", "zero", 5, (1, 4, 128, 128))])
def test_difference_route_equals_pair_route(api, monkeypatch, wave, mode, level, shape, high_precision):
    """Difference-only rules with per-orientation scales: ``sonar_wcfg_fused_*`` transforms cond - uncond alone when the wavelet pair
    reconstructs (perfect_reconstruction = 1) and must agree with the route that transforms cond and uncond side by side.  (The tile
    route forced: with fp32 arithmetic the default is the single-launch kernel, WaveletCFG.single_launch_bands.)"""
    monkeypatch.setattr(api.wc.WaveletCFG, "single_launch_bands", False)
    torch.manual_seed(11)
    cond, uncond, x = (torch.randn(shape, device="cuda") for _ in range(3))
    args = {"input": x, "cond_scale": 7.0, "cond": x - cond, "uncond": x - uncond, "cond_denoised": cond, "uncond_denoised": uncond,
            "sigma": torch.full((shape[0],), 7.0, device="cuda"), "model": FakeModel(), "model_options": MODEL_OPTIONS}
    for blend, strength in (("inject", 1.0), ("lerp", 0.35), ("subtract_b", 0.6)):
        params = dict(difference=dict(yl_scale=0.75, yh_scales=[[3.0, 1.0, 0.5], 0.5, "fill", [1.5, 2.0, 0.25]][: level + 1]), wave=wave, level=level,
                      padding_mode=mode, high_precision_mode=high_precision, difference_blend_mode=blend, difference_blend_strength=strength)
        fn = api.wc.WaveletCFG(existing_cfg=None, rules=api.wc.WCFGRules.build(**params))
        seen = []
        real = api.hl.FusedCall.__init__
        with monkeypatch.context() as m:
            m.setattr(api.hl.FusedCall, "__init__", lambda self, **k: seen.append(k.get("perfect_reconstruction")) or real(self, **k))
            single = fn(args)
        assert seen == [True], "the fused entry point was not used with the reconstruction flag"
        with monkeypatch.context() as m:
            m.setattr(api.wc, "_reconstructs", lambda w: False)
            pair = fn(args)
        peak = float(pair.abs().max())
        torch.testing.assert_close(single, pair, rtol=0, atol=(3e-7 if high_precision else 5e-5) * max(1.0, peak))


def test_max_to_host_matches_torch(api):
    """``sonar_max_to_host_f32`` == ``sigma.max().item()`` (py/wavelet_cfg.py:795-796), NaN rule included, on a side stream too."""
    g = torch.Generator(device="cuda").manual_seed(3)
    for n in (1, 2, 63, 64, 65, 256, 1000, 70001):
        t = torch.randn(n, device="cuda", generator=g) * 50
        assert api.hl.max_to_host(t) == t.max().item()
    t = torch.randn(4, 1, 1, 1, device="cuda", generator=g)
    assert api.hl.max_to_host(t) == t.max().item()
    t = torch.full((300,), -float("inf"), device="cuda")
    assert api.hl.max_to_host(t) == -float("inf")
    t = torch.randn(500, device="cuda", generator=g)
    t[321] = float("nan")
    assert math.isnan(api.hl.max_to_host(t)) and math.isnan(t.max().item())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        t = torch.arange(1000, device="cuda", dtype=torch.float32)
        assert api.hl.max_to_host(t) == 999.0


def test_max_to_host_split(api):
    """``sonar_max_to_host_begin_f32`` / ``_end_f32``: the launch and the wait as two calls; one request per thread at a time."""
    t = torch.randn(777, device="cuda")
    tok = api.hl.max_to_host_begin(t)
    other = torch.randn(1 << 20, device="cuda").square_().sum()  # unrelated work queued behind the request
    assert api.hl.max_to_host_end(tok) == t.max().item()
    assert math.isfinite(other.item())
    tok = api.hl.max_to_host_begin(t)
    with pytest.raises(api.hl.SonarHipError):
        api.hl.max_to_host_begin(t)
    assert api.hl.max_to_host_end(tok) == t.max().item()
    with pytest.raises(api.hl.SonarHipError):
        api.hl.max_to_host_end(tok)
    for k in range(200):  # tickets: every request returns ITS value
        v = torch.full((65,), float(k), device="cuda")
        assert api.hl.max_to_host(v) == float(k)
    t[5] = float("nan")
    assert math.isnan(api.hl.max_to_host_end(api.hl.max_to_host_begin(t)))
