"""-m gpu: the HIP wavelet rows (W scaling / blend, WC, WF) against fixtures produced by the REFERENCE's own Python run end to end
over PyWavelets 1.1.1 (tests/golden/make_wavelet_golden.py; cases in tests/golden/wavelet_cases.py).  No expected value in this
file comes from code written for this repository.

Tolerances: fp64 coefficient arithmetic 1e-12; WaveletCFG outputs are fp32 tensors: 2e-6 of the output scale when the rule
computes in fp64 (high_precision_mode, the reference default: only the final fp32 roundings differ), 5e-5 when it computes in fp32
(the 8-tap dot products of up to 5 levels, PyWavelets' fp32 summation order against fused multiply-adds; outputs reach O(30));
wavelet-filtered noise (fp32 transforms of O(1) noise) 3e-5."""
import importlib
import json
import os
import types

import numpy as np
import pytest
import torch

from tests import wavelet_helpers as wh
from tests.conftest import GOLDEN
from tests.golden import wavelet_cases as wc

pytestmark = pytest.mark.gpu

DWT = np.load(os.path.join(GOLDEN, "dwt.npz"), allow_pickle=False)
SCALING = np.load(os.path.join(GOLDEN, "wavelet_scaling.npz"), allow_pickle=False)
WCFG = np.load(os.path.join(GOLDEN, "wavelet_cfg.npz"), allow_pickle=False)
WF = np.load(os.path.join(GOLDEN, "wavelet_filtered.npz"), allow_pickle=False)
SIG = (torch.tensor(14.6), torch.tensor(10.0))


@pytest.fixture(scope="module")
def api(pkg):
    pkg.hip_lib.load()
    mods = {m: importlib.import_module(f"comfyui_sonar_amd.py.{m}") for m in ("utils", "noise_generation", "noise", "wavelet_functions", "wavelet_cfg")}
    mods["registry"] = importlib.import_module("comfyui_sonar_amd.py.nodes.registry")
    return types.SimpleNamespace(**mods, hl=pkg.hip_lib)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", list(wc.SCALING_CASES))
def test_wavelet_scaling(api, name, dtype):
    tag, yl_scale, yh_scales = wc.SCALING_CASES[name]
    _, _, yl, yh = wc.dwt_case(DWT, tag)
    yl, yh = yl.to(dtype).cuda(), [b.to(dtype).cuda() for b in yh]
    keep_l, keep_h = yl.clone(), [b.clone() for b in yh]
    tol = dict(rtol=1e-12, atol=1e-12) if dtype == torch.float64 else dict(rtol=2e-6, atol=1e-6)

    def check(rl, rh):
        torch.testing.assert_close(rl.cpu().double(), torch.from_numpy(SCALING[f"{name}__yl"]), **tol)
        assert len(rh) == len(yh)
        for j, band in enumerate(rh):
            torch.testing.assert_close(band.cpu().double(), torch.from_numpy(SCALING[f"{name}__yh{j}"]), **tol)

    rl, rh = api.wavelet_functions.wavelet_scaling(yl, yh, yl_scale, yh_scales)
    check(rl, rh)
    assert torch.equal(yl, keep_l) and all(torch.equal(a, b) for a, b in zip(yh, keep_h))  # out of place
    il, ih = api.wavelet_functions.wavelet_scaling(yl, yh, yl_scale, yh_scales, in_place=True)
    assert il is yl and all(a is b for a, b in zip(ih, yh))
    check(il, ih)


@pytest.mark.parametrize("name", list(wc.BLEND_CASES))
def test_wavelet_blend(api, name):
    """wavelet_blend with the package's BLENDING_MODES functions (fp32 kernels) on fp32 copies of the PyWavelets coefficients."""
    tag, fn_l, fn_h, fac_l, fac_h = wc.BLEND_CASES[name]
    _, _, yl, yh = wc.dwt_case(DWT, tag)
    bl, bh = wc.second_coeffs(yl, yh)
    dev = lambda t: t.float().cuda()  # noqa: E731
    modes = api.utils.BLENDING_MODES
    rl, rh = api.wavelet_functions.wavelet_blend((dev(yl), [dev(b) for b in yh]), (dev(bl), [dev(b) for b in bh]), yl_factor=fac_l, yh_factor=fac_h,
                                                 blend_function=modes[fn_l], yh_blend_function=None if fn_h is None else modes[fn_h])
    torch.testing.assert_close(rl.cpu().double(), torch.from_numpy(SCALING[f"blend_{name}__yl"]), rtol=1e-5, atol=2e-6)
    for j, band in enumerate(rh):
        torch.testing.assert_close(band.cpu().double(), torch.from_numpy(SCALING[f"blend_{name}__yh{j}"]), rtol=1e-5, atol=2e-6)


def _wcfg_tol(case, want):
    hp = True
    for params in (case["params"], *case["params"].get("rules", ())):
        hp = hp and params.get("high_precision_mode", True)
    return (2e-6 if hp else 5e-5) * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("name", list(wc.WCFG_CASES))
def test_wavelet_cfg_matches_reference(api, name):
    case = wc.WCFG_CASES[name]
    args = wh.wcfg_args(case, name, wc.FakeModel(), device="cuda")
    for k in ("input", "cond_denoised", "uncond_denoised", "sigma"):
        np.testing.assert_array_equal(args[k].cpu().numpy(), WCFG[f"{name}__{k}"])
    fn = wh.build_wcfg(api.wavelet_cfg, case)
    out = fn(args)
    want = WCFG[f"{name}__out"]
    assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == want.shape
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=_wcfg_tol(case, want))
    # a second call reuses the cached wavelet / workspace and must not depend on the first
    np.testing.assert_allclose(fn(args).cpu().numpy(), want, rtol=0, atol=_wcfg_tol(case, want))


FOUR_D = [n for n, c in wc.WCFG_CASES.items() if len(c["shape"]) == 4 and not c["params"].get("use_1d_dwt")]


def test_lowpass_path_is_taken_where_it_applies(api, monkeypatch):
    """Difference-only rules with one scale per level run as ONE launch (sonar_wcfg_lowpass_*); the others do not."""
    calls = []
    real = api.hl.wcfg_lowpass_plan
    monkeypatch.setattr(api.hl, "wcfg_lowpass_plan", lambda *a, **k: wh.counted_launch(real(*a, **k), calls))  # launches, not preparations
    used = {}
    for name in FOUR_D:
        case = wc.WCFG_CASES[name]
        calls.clear()
        wh.build_wcfg(api.wavelet_cfg, case)(wh.wcfg_args(case, name, wc.FakeModel(), device="cuda"))
        used[name] = bool(calls)
    for name in ("placeholder", "placeholder_f32", "placeholder_128", "identity_scales", "haar_per", "lerp_diff_small_t", "subtract_diff",
                 "two_levels_inv_wave", "second_rule"):
        assert used[name], name
    for name in ("odd_sizes", "lerp_diff", "all_scales", "all_scales_f32", "target_noise", "blend_half", "outside_window", "scheduled_scales"):
        assert not used[name], name


@pytest.mark.parametrize("name", FOUR_D)
def test_wavelet_cfg_band_path_matches_reference(api, name, monkeypatch):
    """The same cases with the low-pass shortcut disabled: the bands resident in LDS (sonar_wcfg_bands_*: one launch for difference-only
    rules, two for rules that scale cond / uncond / final) where the wavelet pair reconstructs, the tile kernels elsewhere."""
    monkeypatch.setattr(api.wavelet_cfg.WaveletCFG, "_lowpass_launch", classmethod(lambda cls, **_k: None))
    monkeypatch.setattr(api.wavelet_cfg.WaveletCFG, "single_launch_bands", True)
    calls = []
    real = api.hl.wcfg_bands
    monkeypatch.setattr(api.hl, "wcfg_bands", lambda *a, **k: calls.append(1) or real(*a, **k))
    case = wc.WCFG_CASES[name]
    args = wh.wcfg_args(case, name, wc.FakeModel(), device="cuda")
    want = WCFG[f"{name}__out"]
    np.testing.assert_allclose(wh.build_wcfg(api.wavelet_cfg, case)(args).cpu().numpy(), want, rtol=0, atol=_wcfg_tol(case, want))
    BANDS_USED[name] = len(calls)


BANDS_USED: dict = {}


def test_band_kernel_is_taken_where_it_applies():
    """One launch for difference-only rules, two when cond / uncond / final are scaled as well, none where the extension pair does not
    reconstruct, the target is not the denoised prediction or the call blends with the fallback CFG."""
    if len(BANDS_USED) < len(FOUR_D):
        pytest.skip("runs after test_wavelet_cfg_band_path_matches_reference")
    for name in ("placeholder", "placeholder_f32", "placeholder_128", "identity_scales", "haar_per", "lerp_diff_small_t", "subtract_diff", "lerp_diff"):
        assert BANDS_USED[name] == 1, (name, BANDS_USED[name])
    for name in ("all_scales", "all_scales_f32"):
        assert BANDS_USED[name] == 2, (name, BANDS_USED[name])
    for name in ("target_noise", "blend_half", "outside_window"):
        assert BANDS_USED[name] == 0, (name, BANDS_USED[name])


@pytest.mark.parametrize("name", FOUR_D)
def test_wavelet_cfg_tile_kernels_match_reference(api, name, monkeypatch):
    """Low-pass shortcut disabled, the tile route forced (the default route for rules that need the bands in fp64 arithmetic; fp32 takes
    the single-launch kernel of the test above unless told otherwise: WaveletCFG.single_launch_bands): level 1 by the tile kernels through
    the workspace, the deeper levels by the LDS-resident kernel where the wavelet pair reconstructs (sonar_wcfg_fused_*, 3 launches; 4
    for cond / uncond rules), by the old per-level walk elsewhere."""
    monkeypatch.setattr(api.wavelet_cfg.WaveletCFG, "_lowpass_launch", classmethod(lambda cls, **_k: None))
    monkeypatch.setattr(api.wavelet_cfg.WaveletCFG, "single_launch_bands", False)
    case = wc.WCFG_CASES[name]
    args = wh.wcfg_args(case, name, wc.FakeModel(), device="cuda")
    want = WCFG[f"{name}__out"]
    np.testing.assert_allclose(wh.build_wcfg(api.wavelet_cfg, case)(args).cpu().numpy(), want, rtol=0, atol=_wcfg_tol(case, want))


@pytest.mark.parametrize("name", FOUR_D)
def test_wavelet_cfg_per_pass_path_matches_reference(api, name, monkeypatch):
    """The same cases with both fast entry points disabled: the per-level kernels behind Wavelet.forward / inverse."""
    monkeypatch.setattr(api.wavelet_cfg.WaveletCFG, "_lowpass_launch", classmethod(lambda cls, **_k: None))
    monkeypatch.setattr(api.wavelet_cfg.WaveletCFG, "wavelet_cfg_bands", classmethod(lambda cls, **_k: None))
    monkeypatch.setattr(api.wavelet_cfg.WaveletCFG, "wavelet_cfg_fused", classmethod(lambda cls, **_k: None))
    case = wc.WCFG_CASES[name]
    args = wh.wcfg_args(case, name, wc.FakeModel(), device="cuda")
    want = WCFG[f"{name}__out"]
    np.testing.assert_allclose(wh.build_wcfg(api.wavelet_cfg, case)(args).cpu().numpy(), want, rtol=0, atol=_wcfg_tol(case, want))


def test_prepared_launch_equals_ordinary_path(api):
    """While the sigma read is in flight the call LAUNCHES the fast path for the rule the previous call matched (WaveletCFG._speculate)
    and keeps the result when its own sigma selects that rule.  Same bits as the ordinary order, across changes of rule and calls outside
    every window, and the launch made ahead is really the one returned."""
    case = wc.WCFG_CASES["second_rule"]
    fn = wh.build_wcfg(api.wavelet_cfg, case)
    plain = wh.build_wcfg(api.wavelet_cfg, case)
    seen = []
    real = fn._speculate
    fn._speculate = lambda a: seen.append(real(a)) or seen[-1]
    plain._speculate = lambda a: None
    base = wh.wcfg_args(case, "second_rule", wc.FakeModel(), device="cuda")
    for k, sigma in enumerate((3.0, 3.0, 9.0, 9.0, 3.0, 20.0, 3.0, 3.0)):
        args = dict(base, sigma=torch.full_like(base["sigma"], sigma))
        got, want = fn(args), plain(args)
        assert torch.equal(got, want), (k, sigma)
        rule = fn.rules.get_rule(sigma)
        # launched for the rule of the previous call (nothing after a call that matched no rule, or before the first): kept when this
        # call matches the same rule
        assert (seen[-1] is None) == (k in (0, 6)), (k, sigma)
        assert (seen[-1] is not None and seen[-1][0] is rule) == (k in (1, 3, 7)), (k, sigma)
        if k in (1, 3, 7):
            assert got.data_ptr() == seen[-1][2].data_ptr()
    np.testing.assert_allclose(fn(base).cpu().numpy(), WCFG["second_rule__out"], rtol=0, atol=_wcfg_tol(case, WCFG["second_rule__out"]))
    # the percentages are still built (under the kernel), so their errors still surface: no sample_sigmas -> the reference's error
    bad = dict(base, model_options={"transformer_options": {}})
    fn(base)
    with pytest.raises(UnboundLocalError):
        fn(bad)
    with pytest.raises(UnboundLocalError):
        plain(bad)
    # user operations on cond / uncond run inside get_context: nothing is prepared then
    op = wh.build_wcfg(api.wavelet_cfg, case)
    op.operation_cond = lambda latent, **_k: latent
    assert op._speculate(base) is None
    torch.testing.assert_close(op(base), plain(base), rtol=0, atol=0)


def test_wavelet_cfg_errors(api):
    errors = json.loads(str(WCFG["errors_json"]))
    for name, case in wc.WCFG_ERRORS.items():
        args = wh.wcfg_args(case, name, wc.FakeModel(), device="cuda")
        kind, msg = errors[name]
        with pytest.raises(Exception) as exc:
            wh.build_wcfg(api.wavelet_cfg, case)(args)
        assert type(exc.value).__name__ == kind, (name, exc.value)
        if kind == "RuntimeError":
            assert str(exc.value) == msg


@pytest.mark.parametrize("name", list(wc.WF_GEN_CASES))
def test_wavelet_filtered_generator(api, name):
    case = wc.WF_GEN_CASES[name]
    shape = tuple(case["shape"])
    torch.manual_seed(9)
    gen = api.noise_generation.WaveletFilteredNoiseGenerator(torch.zeros(shape, device="cuda"), sigma_min=0.03, sigma_max=14.6, seed=9, cpu=True,
                                                             normalized=False, **json.loads(json.dumps(case["kw"])))
    out = gen(*SIG)
    want = WF[f"gen_{name}__out"]
    assert tuple(out.shape) == want.shape and out.dtype == torch.float32
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=3e-5 * max(1.0, float(np.abs(want).max())))


def _gauss_chain(api):
    chain = api.noise.CustomNoiseChain()
    chain.add(api.noise.CustomNoiseItem(1.0, noise_type="gaussian"))
    return chain


@pytest.mark.parametrize("name", list(wc.WF_ITEM_CASES))
def test_wavelet_filtered_item(api, name):
    import yaml

    case = wc.WF_ITEM_CASES[name]
    shape = tuple(case["shape"])
    item = api.noise.WaveletFilteredNoise(1.0, noise=_gauss_chain(api), noise_high=_gauss_chain(api) if case["high"] else None, normalize=None,
                                          normalize_noise=case["normalize_noise"], yaml_parameters=yaml.safe_dump(case["yaml"]))
    torch.manual_seed(12)
    out = item.make_noise_sampler(torch.zeros(shape, device="cuda"), 0.03, 14.6, seed=12, cpu=True, normalized=case["normalized"])(*SIG)
    want = WF[f"item_{name}__out"]
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=5e-5 * max(1.0, float(np.abs(want).max())))


def test_wavelet_filtered_node(api):
    nc = wc.WF_NODE_CASE
    node = api.registry.NODE_CLASS_MAPPINGS["SonarWaveletFilteredNoise"]()
    (chain,) = node.go(factor=1.0, rescale=0.0, normalize="disabled", normalize_noise=False, custom_noise=_gauss_chain(api), yaml_parameters=nc["yaml"])
    torch.manual_seed(nc["seed"])
    out = chain.make_noise_sampler(torch.zeros(nc["shape"], device="cuda"), 0.03, 14.6, seed=nc["seed"], cpu=True, normalized=False)(*SIG)
    want = WF["node__out"]
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=3e-5 * float(np.abs(want).max()))
