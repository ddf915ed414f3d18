"""-m gpu, round 6: the look-ahead forms added this round -- a fused launch equals the launches it replaces bit for bit, whatever the
plan's hook finds or misses."""
import importlib
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

SIG = (torch.tensor(14.6), torch.tensor(10.0))


@pytest.fixture(scope="module")
def api(pkg):
    pkg.hip_lib.load()
    return types.SimpleNamespace(hl=pkg.hip_lib, nz=importlib.import_module("comfyui_sonar_amd.py.noise"),
                                 ng=importlib.import_module("comfyui_sonar_amd.py.noise_generation"))


@pytest.mark.parametrize("uniform,factor", [(True, 1.0), (True, 0.7), (False, 1.5), (False, 1.0)])
@pytest.mark.parametrize("n,offset", [(4 * 128 * 128 * 3, 0), (4096 * 5 + 36, 8), (1001, 3), (4 * 64 * 64 * 70, 4 * 64 * 64 * 9)])
def test_fill_ahead_entry_point_is_the_two_launches(api, uniform, factor, n, offset):
    """sonar_philox_noise_ahead_f32 (this call's final pass + the next call's statistics pass in one launch) against sonar_philox_noise_f32
    (statistics pass, final pass): the same output bits with and without statistics left by an earlier launch, and the partials it leaves
    for the next call are the ones that call's own statistics pass writes (reference: py/noise_generation.py:496-514 + py/utils.py:85-106)."""
    hl = api.hl
    lib = hl.load()
    st = hl._stream()
    kw = dict(sub=0.5, mul=3.46, add=0.1) if uniform else {}
    aff = (kw.get("sub", 0.0), kw.get("mul", 1.0), kw.get("add", 0.0))
    want0 = hl.philox_noise(uniform, (n,), "cuda", 5, 30, offset, factor, **kw)
    want1 = hl.philox_noise(uniform, (n,), "cuda", 5, 37, offset, factor, **kw)
    assert lib.sonar_philox_noise_ahead_ok(int(uniform), n, factor) == 1
    # the statistics pass alone: what call 37 would compute for itself
    own = hl.new_partials("cuda")
    dry = torch.empty(n, device="cuda")
    assert lib.sonar_philox_noise_f32(int(uniform), dry.data_ptr(), n, 5, 37, offset, *aff, factor, 2.5, own.data_ptr(), st) == 0
    p0, p1, p2 = hl.new_partials("cuda"), hl.new_partials("cuda"), hl.new_partials("cuda")
    out0, out1 = torch.empty(n + 4, device="cuda")[4:], torch.empty(n, device="cuda")
    # call 30 finds nothing (have_stats = 0), leaves call 37's statistics; call 37 uses them (have_stats = 1)
    assert lib.sonar_philox_noise_ahead_f32(int(uniform), out0.data_ptr(), n, 5, 30, offset, *aff, factor, 2.5, p0.data_ptr(), 0, 37, p1.data_ptr(), st) == 0
    assert torch.equal(out0, want0)
    assert torch.equal(p1, own)
    assert lib.sonar_philox_noise_ahead_f32(int(uniform), out1.data_ptr(), n, 5, 37, offset, *aff, factor, 2.5, p1.data_ptr(), 1, 44, p2.data_ptr(), st) == 0
    assert torch.equal(out1, want1)
    # refusal: the two statistics buffers must differ
    assert lib.sonar_philox_noise_ahead_f32(1, out1.data_ptr(), n, 5, 37, offset, *aff, factor, 2.5, p1.data_ptr(), 1, 44, p1.data_ptr(), st) == hl.ERR_ARG


def test_fill_ahead_of_unit_normals_corrects_like_scale_noise(api):
    """N(0,1) with factor 1: the ordinary route stores the raw draws and lets scale_noise's kernel correct them in place when a threshold
    fails (1.3 % of tensors at 2.5 standard errors).  With a threshold of 0 every tensor fails both: the look-ahead form must apply that
    kernel's subtract / IEEE-divide sequence, not the reciprocal multiply of the other shapes."""
    hl = api.hl
    lib = hl.load()
    st = hl._stream()
    for n, offset in ((4096 * 6, 4096), (4096 * 3 + 20, 0), (999, 1)):
        want = hl.philox_noise(False, (n,), "cuda", 7, 12, offset, 1.0, threshold_std_devs=0.0)
        p0, p1 = hl.new_partials("cuda"), hl.new_partials("cuda")
        out = torch.empty(n, device="cuda")
        assert lib.sonar_philox_noise_ahead_f32(0, out.data_ptr(), n, 7, 12, offset, 0.0, 1.0, 0.0, 1.0, 0.0, p0.data_ptr(), 0, 13, p1.data_ptr(), st) == 0
        assert torch.equal(out, want)
        assert abs(out.double().mean().item()) < 1e-6 and abs(out.double().std().item() - 1.0) < 1e-5


@pytest.mark.parametrize("name", ["uniform", "gaussian", "gaussian_scaled"])
@pytest.mark.parametrize("shape", [(1, 4, 128, 128), (64, 4, 128, 128), (3, 4, 104, 152), (2, 4, 3, 64, 64)])
def test_a_planned_normalised_fill_runs_its_statistics_a_call_ahead(api, name, shape):
    """Inside a plan a normalised uniform fill (and a Gaussian one with factor != 1) is ONE launch per call in the steady state: same bits as
    the ordinary path whatever the hook finds -- a reseed and a foreign draw in the middle of the run cost shortcuts, not values."""
    hl, nz = api.hl, api.nz
    x = torch.zeros(shape, device="cuda")
    factor = 1.0 if name != "gaussian_scaled" else 0.8
    name = "gaussian" if name == "gaussian_scaled" else name
    make = lambda: nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True, factor=factor)  # noqa: E731
    a, b = make(), make()

    def script(ns, plans):
        old = hl.PLANS_ENABLED
        hl.PLANS_ENABLED = plans
        try:
            torch.manual_seed(31)
            got = []
            for i in range(14):
                if i == 8:
                    torch.manual_seed(32)
                if i == 11:
                    torch.randn(5, device="cuda")
                got.append(ns(*SIG).clone())
            return got
        finally:
            hl.PLANS_ENABLED = old

    ra, rb = script(a, True), script(b, False)
    assert all(torch.equal(p, q) for p, q in zip(ra, rb))
    planned = a if isinstance(a, hl.Planned) else getattr(a, "_planned", None)
    plan = planned.plan
    assert plan is not None, getattr(planned, "reason", None)
    hooks = [h for h in plan.hooks if isinstance(h, hl._FillAheadHook)]
    assert len(hooks) == 1 and hooks[0].hits >= 6 and hooks[0].misses >= 3  # first run, the reseed, the foreign draw
    # the same sampler with the look-ahead switched off plans the two-launch form: same bits again
    old = hl.FILL_AHEAD
    hl.FILL_AHEAD = False
    try:
        c = make()
        rc = script(c, True)
    finally:
        hl.FILL_AHEAD = old
    assert all(torch.equal(p, q) for p, q in zip(rc, rb))
