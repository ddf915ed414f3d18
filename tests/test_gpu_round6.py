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


# ------------------------------------------------------------------------------------------------ Brownian noise, round 6
def test_brownian_node_bursts_are_independent_unit_normals(api):
    """Round 6 seeds a node's burst by hashing the sub-tile's Philox-drawn base state with the node id (fmix32 of base ^ splitmix64(node))
    instead of a Philox block per (node, sub-tile, lane), and draws with 23 radius / 16 angle bits.  What a Brownian path needs of them:
    every node's field is N(0, 1) (moments, a 64-bin chi-square against the normal law, the tail), fields of different nodes -- siblings,
    parent and child, consecutive ids, ids that differ in one high bit -- are uncorrelated, and so are a field's neighbours at the lags the
    generator's layout could couple (the next value of a lane, the next lane, the next step, the next sub-tile)."""
    import math

    hl = api.hl
    shape = (32, 16, 128, 128)
    n = math.prod(shape)
    nodes = [1, 2, 3, 4, 5, 1 << 23, (1 << 23) + 1, (1 << 24) - 1, 12345677, 12345678, (1 << 40) + 7]
    fields = {k: hl.brownian(shape, "cuda", [k], [1.0], 20240607) for k in nodes}
    tol = 5.0 / math.sqrt(n)
    edges = torch.linspace(-4.0, 4.0, 65, device="cuda", dtype=torch.float64)
    cdf = 0.5 * (1.0 + torch.erf(edges / math.sqrt(2.0)))
    expect = torch.cat([cdf[:1], cdf[1:] - cdf[:-1], 1.0 - cdf[-1:]]) * n
    for k, z in fields.items():
        zd = z.double().flatten()
        assert abs(zd.mean().item()) < tol and abs(zd.var().item() - 1.0) < 3 * tol, k
        assert abs((zd**4).mean().item() - 3.0) < 40 * tol and abs((zd**3).mean().item()) < 20 * tol, k
        counts = torch.bincount(torch.bucketize(zd, edges), minlength=66).double()
        chi2 = (((counts - expect) ** 2) / expect).sum().item()
        assert chi2 < 66 + 6 * math.sqrt(2 * 66), (k, chi2)  # 65 degrees of freedom
        assert 4.0 < zd.abs().max().item() < 5.7  # 23 radius bits: |z| <= sqrt(2 ln 2^23) = 5.65
        for lag in (1, 2, 4, 256, 1024, 4096, 65536):
            assert abs((zd[:-lag] * zd[lag:]).mean().item()) < tol, (k, lag)
        # squares too: a shared radius (the two values of a Box-Muller pair) must not show as correlated magnitudes beyond the pair itself
        sq = zd * zd - 1.0
        for lag in (2, 4, 256, 1024):
            assert abs((sq[:-lag] * sq[lag:]).mean().item()) < 3 * tol, (k, lag)
    keys = list(fields)
    for i, a in enumerate(keys):
        for b in keys[i + 1:]:
            za, zb = fields[a].double().flatten(), fields[b].double().flatten()
            assert abs((za * zb).mean().item()) < tol, (a, b)
            assert abs(((za * za - 1.0) * (zb * zb - 1.0)).mean().item()) < 3 * tol, (a, b)
    # another seed: another field; the same seed: the same bits; a shard: its part of the batch
    again = hl.brownian(shape, "cuda", [5], [1.0], 20240607)
    other = hl.brownian(shape, "cuda", [5], [1.0], 20240608)
    assert torch.equal(again, fields[5]) and abs((other.double() * fields[5].double()).mean().item()) < tol
    part = hl.brownian((8, *shape[1:]), "cuda", [5], [1.0], 20240607, elem_offset=8 * math.prod(shape[1:]))
    assert torch.equal(part, fields[5][8:16])


def test_brownian_times_inside_one_tree_cell_give_a_zero_increment(api):
    """Two distinct sigmas closer than the tree's grid (6e-8 of the range at depth 24) are the same point of the path: the increment is 0, as
    torchsde's tree with its tolerance returns (it used to raise ValueError); a chain folds it as zeros."""
    ng = api.ng
    x = torch.zeros(2, 4, 64, 64, device="cuda")
    ns = ng.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=3, tree_depth=24)
    cell = (14.6 - 0.03) / (1 << 24)
    out = ns(torch.tensor(7.0, dtype=torch.float64), torch.tensor(7.0 + cell / 16, dtype=torch.float64))
    assert out.shape == x.shape and float(out.abs().max()) == 0.0
    acc = torch.full_like(x, 2.0)
    assert ns.accumulate(acc, 0.5, 3.0, None, torch.tensor(7.0, dtype=torch.float64), torch.tensor(7.0 + cell / 16, dtype=torch.float64))
    assert torch.equal(acc, torch.full_like(x, 1.0))
    # a narrow range at a large offset cannot resolve 24 levels: the tree is clamped to what float times can tell apart
    assert ng.BrownianPath(1.0e6, 1.0e6 + 1e-3, 24).tree_depth < 24
    assert ng.BrownianPath(0.03, 14.6, 24).tree_depth == 24


def test_power_spectrum_angle_and_radius_are_independent_at_1e8_draws(api):
    """The power-law draw takes 16 angle bits per value (two values per angle word) and 23 radius bits.  Over 10^8 drawn
    spectrum values: a 64 x 32 chi-square of (angle, radius^2 quantile) against independence with uniform margins, the angle's own 4096-bin
    histogram (65 536 directions: sixteen per bin), lag correlations of the angle itself (next column, next row, next plane, the
    partner row that shares the angle word), and the discreteness of the directions -- a measured choice, not an assumed one."""
    import math

    hl = api.hl
    tot = 0
    joint = torch.zeros(64 * 32, dtype=torch.float64, device="cuda")
    fine = torch.zeros(4096, dtype=torch.float64, device="cuda")
    lag_sums = {name: 0.0 for name in ("column", "row", "plane", "partner")}
    lag_counts = dict.fromkeys(lag_sums, 0)
    for call in range(13):  # 13 x 256 latents x 4 planes x 128 x 63 interior values = 1.07e8
        z = hl.power_spectrum((256, 4, 128, 128), "cuda", seed=777, stream_id=10 + 3 * call)[..., 1:64]  # the interior columns (edge columns: own stream)
        re, im = z.real.double(), z.imag.double()
        # direction in revolutions, [0, 1): the drawn directions are multiples of 2^-16, and v_sin / v_cos leave them ~1e-3 of such a step
        # off -- half a step is added so that a direction never sits on a bin edge
        rev = (torch.atan2(im, re) / (2 * math.pi) + 0.5 / 65536) % 1.0
        q = 1.0 - torch.exp(-(re * re + im * im))                 # radius^2 ~ Exp(1): its CDF value is uniform
        a_bin = (rev * 64).long().clamp_(0, 63)
        r_bin = (q * 32).long().clamp_(0, 31)
        joint += torch.bincount((a_bin * 32 + r_bin).flatten(), minlength=64 * 32).double()
        fine += torch.bincount((rev * 4096).long().clamp_(0, 4095).flatten(), minlength=4096).double()
        tot += rev.numel()
        c = torch.cos(2 * math.pi * rev)                          # (a circular statistic: mean zero, variance 1/2)
        half = c.shape[-2] // 2
        for name, a, b in (("column", c[..., :-1], c[..., 1:]), ("row", c[..., :-1, :], c[..., 1:, :]), ("plane", c[:, :-1], c[:, 1:]),
                           ("partner", c[..., :half, :], c[..., half:, :])):
            lag_sums[name] += float((a * b).sum())
            lag_counts[name] += a.numel()
        # every direction is a multiple of 2^-16 revolutions (sampled: the rounding of atan2 in fp64 leaves ~1e-12)
        sample = rev.flatten()[:: 4099][:200000] * 65536.0 - 0.5
        assert float((sample - sample.round()).abs().max()) < 0.05
    assert tot >= 10**8
    expect = tot / (64 * 32)
    chi2 = float((((joint - expect) ** 2) / expect).sum())
    dof = 64 * 32 - 1
    assert abs(chi2 - dof) < 6 * math.sqrt(2 * dof), chi2
    exp_fine = tot / 4096
    chi2f = float((((fine - exp_fine) ** 2) / exp_fine).sum())
    assert abs(chi2f - 4095) < 6 * math.sqrt(2 * 4095), chi2f
    for name, total in lag_sums.items():
        assert abs(total / lag_counts[name]) < 5 * 0.5 / math.sqrt(lag_counts[name]), (name, total / lag_counts[name])


# ------------------------------------------------------------------------------------------------ pyramid look-ahead
@pytest.mark.parametrize("shape", [(1, 4, 128, 128), (64, 4, 128, 128), (3, 4, 104, 152), (2, 4, 3, 64, 64), (300, 4, 64, 64)])
def test_a_planned_normalised_pyramid_call_is_one_launch(api, shape):
    """Inside a plan a normalised pyramid call (py/noise_generation.py:609-649 + py/utils.py:85-106) is ONE launch in the steady state
    (sonar_pyramid_noise_ahead_f32: this call's planes stored normalised, the next call's planes evaluated for their statistics only): the
    same bits as the generating launch + the in-place scale_noise pass, whatever the hook finds -- a reseed and a foreign draw cost
    shortcuts, not values -- and the same bits with the look-ahead switched off."""
    hl, nz = api.hl, api.nz
    x = torch.zeros(shape, device="cuda")
    make = lambda: nz.get_noise_sampler("pyramid", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)  # noqa: E731

    def script(ns, plans):
        old = hl.PLANS_ENABLED
        hl.PLANS_ENABLED = plans
        try:
            torch.manual_seed(41)
            got = []
            for i in range(16):
                if i == 9:
                    torch.manual_seed(42)
                if i == 12:
                    torch.randn(5, device="cuda")
                got.append(ns(*SIG).clone())
            return got
        finally:
            hl.PLANS_ENABLED = old

    a, b = make(), make()
    ra, rb = script(a, True), script(b, False)
    assert all(torch.equal(p, q) for p, q in zip(ra, rb)), [i for i, (p, q) in enumerate(zip(ra, rb)) if not torch.equal(p, q)]
    planned = a if isinstance(a, hl.Planned) else getattr(a, "_planned", None)
    plan = planned.plan
    assert plan is not None, getattr(planned, "reason", None)
    assert hl.load().sonar_plan_length(plan.handle) == 1
    hooks = [h for h in plan.hooks if isinstance(h, hl._FillAheadHook)]
    assert len(hooks) == 1 and hooks[0].hits >= 5 and hooks[0].misses >= 2, (hooks[0].hits, hooks[0].misses)
    old = hl.PYRAMID_AHEAD
    hl.PYRAMID_AHEAD = False
    try:
        rc = script(make(), True)
    finally:
        hl.PYRAMID_AHEAD = old
    assert all(torch.equal(p, q) for p, q in zip(rc, rb))


def test_pyramid_ahead_entry_point_is_the_two_launches(api):
    """The entry point itself against sonar_pyramid_noise_f32: with and without statistics left by an earlier launch, and the partials it
    leaves for the next call against the ones that call's own generating launch writes."""
    import ctypes as C

    hl = api.hl
    lib = hl.load()
    st = hl._stream()
    shape, planes, h, w = (6, 4, 128, 128), 24, 128, 128
    lv0, lv1 = hl.AutoLevels(h, w, 10, 0.7, 9, 50), hl.AutoLevels(h, w, 10, 0.7, 9, 57)
    want0 = hl.pyramid_noise(shape, "cuda", lv0, "bilinear", 9, 50, 0, 0.9)
    want1 = hl.pyramid_noise(shape, "cuda", lv1, "bilinear", 9, 57, 0, 0.9)
    assert want0 is not None and want1 is not None
    own = hl.new_partials("cuda")
    assert hl.pyramid_generate(shape, "cuda", lv1, "bilinear", 9, 57, 0, partials=own) is not None  # call 57's own statistics

    def tables(lv):
        n = len(lv)
        return (n, (C.c_int64 * n)(*[v[1] for v in lv]), (C.c_int64 * n)(*[v[2] for v in lv]), (C.c_float * n)(*[v[3] for v in lv]))

    t0, t1 = tables(lv0), tables(lv1)
    p0, p1, p2 = hl.new_partials("cuda"), hl.new_partials("cuda"), hl.new_partials("cuda")
    out0, out1 = torch.empty(shape, device="cuda"), torch.empty(shape, device="cuda")
    rc = lib.sonar_pyramid_noise_ahead_f32(out0.data_ptr(), planes, h, w, *t0, 0, 9, 50, 0, 0.9, 2.5, p0.data_ptr(), 0, 57, *t1, p1.data_ptr(), st)
    assert rc == 0, hl.load().sonar_last_error()
    assert torch.equal(out0, want0)
    g = min(planes, 1024)
    assert torch.equal(p1[:2 * g], own[:2 * g]) and float(p1[2 * g:].abs().max()) == 0.0
    rc = lib.sonar_pyramid_noise_ahead_f32(out1.data_ptr(), planes, h, w, *t1, 0, 9, 57, 0, 0.9, 2.5, p1.data_ptr(), 1, 64, *t0, p2.data_ptr(), st)
    assert rc == 0 and torch.equal(out1, want1)
    # refusals: one workspace for both calls' statistics; a resampling mode the look-ahead form does not take
    assert lib.sonar_pyramid_noise_ahead_f32(out1.data_ptr(), planes, h, w, *t1, 0, 9, 57, 0, 0.9, 2.5, p1.data_ptr(), 1, 64, *t0, p1.data_ptr(), st) == hl.ERR_ARG
    assert lib.sonar_pyramid_noise_ahead_f32(out1.data_ptr(), planes, h, w, *t1, 1, 9, 57, 0, 0.9, 2.5, p1.data_ptr(), 1, 64, *t0, p2.data_ptr(), st) == hl.ERR_UNSUPPORTED


@pytest.mark.parametrize("shape,what", [((2, 3, 33, 17), "pyramid"), ((3, 4, 18, 30), "pyramid"), ((2, 4, 3, 7, 9), "chain"), ((4, 4, 64, 64), "pyramid")])
def test_pyramid_levels_sized_on_the_host_are_never_frozen_into_a_plan(api, shape, what):
    """A pyramid draw on a plane the plane kernel does not take (a width that is not a multiple of four) makes its level grids as tensors
    whose sizes the call computes from (seed, stream) -- a plan that recorded those launches replayed ONE call's sizes ever after
    (found by scratch/fuzz_plans_r6.py in round 6: values off by O(1) from the fourth call on).  Such a trace yields no plan; the replayed
    and the ordinary run agree bit for bit on every shape."""
    hl, nz = api.hl, api.nz
    x = torch.zeros(shape, device="cuda")

    def make():
        if what == "chain":
            chain = nz.CustomNoiseChain()
            chain.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
            chain.add(nz.CustomNoiseItem(0.5, noise_type="uniform"))
            return chain.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        return nz.get_noise_sampler("pyramid", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)

    def run(plans):
        old = hl.PLANS_ENABLED
        hl.PLANS_ENABLED = plans
        try:
            torch.manual_seed(77)
            ns = make()
            return ns, [ns(*SIG).clone() for _ in range(10)]
        finally:
            hl.PLANS_ENABLED = old

    na, a = run(True)
    _, b = run(False)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    planned = na if isinstance(na, hl.Planned) else getattr(na, "_planned", None)
    if shape[-1] % 4:
        assert planned is None or planned.plan is None
    else:
        assert planned is not None and planned.plan is not None


def test_brownian_tree_values_do_not_depend_on_what_is_kept(api):
    """The tree's two-stage rule defines W(t); CACHE_POINTS / COARSE_POINTS only say how many tensors a sampler keeps.  A sampler that keeps
    nothing gives the bits of one that keeps everything."""
    ng = api.ng
    x = torch.zeros(2, 4, 64, 64, device="cuda")
    T = lambda v: torch.tensor(v)  # noqa: E731
    keep, bare = (ng.BrownianTreeNoiseSampler(x, 0.03, 14.6, seed=21, tree_depth=24) for _ in range(2))
    bare.CACHE_POINTS = 0
    bare.COARSE_POINTS = 0
    sig = [14.6 * 0.85**k for k in range(12)]
    for k in range(11):
        mid = (sig[k] * sig[k + 1]) ** 0.5
        for a, b in ((sig[k], mid), (sig[k], sig[k + 1])):
            assert torch.equal(keep(T(a), T(b)), bare(T(a), T(b))), (k, a, b)
    assert not bare._points and len(bare._coarse_pts) == 0 and 0 < len(keep._coarse_pts) <= keep.COARSE_POINTS
