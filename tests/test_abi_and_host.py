"""CPU: the C-ABI library builds, loads and exports every symbol include/sonar_hip.h declares (no
compute calls without a GPU); host-side logic that needs no device."""
import importlib
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "sonar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sonar_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_header_symbol(pkg):
    import __graft_entry__

    __graft_entry__.build()
    lib = pkg.hip_lib.load()
    syms = header_symbols()
    assert len(syms) >= 35
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in include/sonar_hip.h but not exported"
    assert set(syms) == set(pkg.hip_lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.sonar_abi_version() == 1
    assert lib.sonar_noise_stream_version() == 6  # generate-mode values: changes with every seed break (include/sonar_hip.h lists them)
    assert lib.sonar_last_error() is not None


def test_only_the_catch_all_plane_kernel_spills(pkg):
    """Round 5: no kernel of the library spills registers / uses a scratch segment, with ONE named exception -- the general-size plane kernel
    instantiated with every codelet length in its run-time switches (power_any_all.hip: plane sizes with a factor of 13 .. 19 that are no
    SDXL bucket; direct sums instead would cost them 25-60 % more time than the spills).  Round 4 had 37 spilling kernels, the ones serving
    every non-square SDXL latent among them (62 vector registers).  The built library's code objects say what every kernel needs."""
    import __graft_entry__
    import importlib.util

    __graft_entry__.build()
    spec = importlib.util.spec_from_file_location("so_kernels", os.path.join(ROOT, "tools", "so_kernels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.kernels(os.path.join(ROOT, "comfyui-sonar_amd", "libsonar_hip.so"))
    assert len(rows) > 900, "the library's code objects were not found"
    catch_all = re.compile(r"power_irfft2_any_kernel<\d+, \d, (true|false), (true|false), 0, 0, 0, 0, 2>")
    bad = [f"{r['pretty']}: {r['scratch']} B ({r['spill_v']} vector registers spilled)" for r in rows
           if r["scratch"] > 0 and not catch_all.search(r["pretty"])]
    assert not bad, "kernels with a scratch segment:\n" + "\n".join(bad)
    # the SDXL buckets' general-size kernels exist with compile-time factor pairs (104 x 152: 13 x 8 rows, 19 x 4 half-columns)
    assert any("power_irfft2_any_kernel<512, 1, false, true, 13, 8, 19, 4, 0>" in r["pretty"] for r in rows)


def test_product_never_touches_the_oracle():
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "comfyui-sonar_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} references the oracle"
                assert "/root/reference" not in src


def test_cpu_tensors_fail_loudly(pkg):
    hl = pkg.hip_lib
    with pytest.raises(hl.SonarHipError):
        hl.stats(torch.zeros(8))
    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
    with pytest.raises(hl.SonarHipError):
        ng.GaussianNoiseGenerator(torch.zeros(1, 4, 8, 8))


def test_noise_type_enum_and_registry(pkg):
    nz = importlib.import_module("comfyui_sonar_amd.py.noise")
    names = [n.name for n in nz.NoiseType]
    assert len(names) == 38 and names[0] == "BROWNIAN" and names[-1] == "WHITE"
    listed = list(nz.NoiseType.get_names())
    assert listed[0] == "gaussian" and len(listed) == 38 and len(set(listed)) == 38
    assert "brownian" not in list(nz.NoiseType.get_names(skip=(nz.NoiseType.BROWNIAN,)))
    assert set(nz.NOISE_SAMPLERS) == set(nz.NoiseType)


def test_chain_factor_algebra(pkg):
    nz = importlib.import_module("comfyui_sonar_amd.py.noise")
    chain = nz.CustomNoiseChain()
    for f, t in ((0.6, "GAUSSIAN"), (-0.3, "UNIFORM"), (0.5, "PERLIN")):
        chain.add(nz.CustomNoiseItem(f, noise_type=nz.NoiseType[t]))
    assert chain.factor == pytest.approx(1.4)
    r = chain.rescaled(2.0)
    assert r.factor == pytest.approx(2.0) and [i.factor < 0 for i in r.items] == [False, True, False]
    assert chain.factor == pytest.approx(1.4)  # rescaled() clones
    with pytest.raises(ValueError):
        chain.add(None)
    with pytest.raises(ValueError):
        nz.CustomNoiseItem(1.0)
    item = nz.CustomNoiseItem(1.0, noise_type=nz.NoiseType.PERLIN, yaml_parameters="blend_mode: inject\niterations: 3")
    assert item.ns_kwargs == {"blend_mode": "inject", "iterations": 3}
    with pytest.raises(ValueError):
        nz.CustomNoiseItem(1.0, noise_type=nz.NoiseType.PERLIN, yaml_parameters="- a\n- b")


def test_sonar_config_and_ratios(pkg):
    S = importlib.import_module("comfyui_sonar_amd.py.sonar")
    assert len(S.SonarConfig._fields) == 17
    sb = S.SonarBase(S.SonarConfig())
    assert sb.history_ratios == (0.75, 1.0, 1.0)
    assert S.SonarBase(S.SonarConfig(direction=-0.5)).history_ratios == (0.75, 1.0 + 0.5 * 0.25, -0.5)
    assert S.SonarBase(S.SonarConfig(direction=1.5)).history_ratios[1] == 0.5
    gated = S.SonarBase(S.SonarConfig(momentum_start_step=2, momentum_end_step=4, always_update_history=False))
    assert [gated.check_step(i) for i in (1, 2, 4, 5)] == [False, True, True, False]
    assert gated.check_step(0, is_history=True) is False and sb.check_step(0, is_history=True) is True
    kc = gated.kernel_cfg(0)
    assert (kc.use_momentum, kc.update_hist) == (0, 0)
    down, up = S.get_ancestral_step(torch.tensor(10.0), torch.tensor(6.0), 1.0)
    assert abs(down.item() ** 2 + up.item() ** 2 - 36.0) < 1e-4


def test_power_filter_host_build_matches_golden(pkg, golden):
    pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
    g = golden("power_filter")
    f = pn.PowerFilter.normalize(pn.PowerFilter(alpha=1.0, max_freq=0.7071).build((1, 4, 128, 128)), (1, 4, 128, 128))
    torch.testing.assert_close(f, g["cfg2_128x128"], rtol=2e-6, atol=1e-30)
    a = pn.PowerFilter(alpha=1.0).build((1, 4, 32, 32))
    b = pn.PowerFilter(alpha=0.0, min_freq=0.2, max_freq=0.3).build((1, 4, 32, 32))
    for mode in ("max", "min", "add", "sub", "mul"):
        torch.testing.assert_close(pn.PowerFilter.compose(a.clone(), b.clone(), mode), g[f"compose_{mode}"], rtol=2e-6, atol=1e-30)
    composed = pn.PowerFilter(alpha=1.0, compose_with=pn.PowerFilter(alpha=0.0, min_freq=0.2, max_freq=0.3)).build((1, 4, 32, 32))
    torch.testing.assert_close(composed, g["compose_max"], rtol=2e-6, atol=1e-30)
    clone = pn.PowerFilter(alpha=0.5, rotate=10.0, compose_with=pn.PowerFilter(alpha=2.0)).clone()
    assert clone.alpha == 0.5 and clone.compose_with.alpha == 2.0
    assert pn.ChannelMixer(4, 0.0, torch.ones(6)).is_identity and not pn.ChannelMixer(4, 0.25, torch.ones(6)).is_identity


def test_crop_samples(pkg):
    u = importlib.import_module("comfyui_sonar_amd.py.utils")
    t = torch.arange(6 * 8).reshape(1, 6, 8)
    assert torch.equal(u.crop_samples(t, 4, 2), t[..., 2:4, 2:6])
    assert torch.equal(u.crop_samples(t, 4, 2, mode="top_left"), t[..., 0:2, 0:4])
    assert torch.equal(u.crop_samples(t, 4, 2, mode="bottom_right"), t[..., 4:6, 4:8])
    assert torch.equal(u.crop_samples(t, 4, 2, mode="center", offset_width=10), t[..., 2:4, 4:8])
    with pytest.raises(ValueError):
        u.crop_samples(t, 16, 2)


def test_brownian_path_coefficients_are_a_brownian_motion(pkg):
    """Host side of sonar_brownian_f32: W(t) = sum coef * z(node) must have Var W(t) = t - t_lo and Cov(W(a), W(b)) =
    min(a, b) - t_lo exactly (any set of i.i.d. N(0,1) node normals then gives ONE consistent Brownian path), and every
    increment divided by sqrt(dt) must have unit variance."""
    import importlib
    import random

    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
    bp = ng.BrownianPath(0.03, 14.6)
    rnd = random.Random(1)
    var = lambda c: sum(v * v for v in c.values())  # noqa: E731
    cov = lambda c1, c2: sum(v * c2.get(k, 0.0) for k, v in c1.items())  # noqa: E731
    for _ in range(300):
        a, b = sorted(rnd.uniform(0.03, 14.6) for _ in range(2))
        ca, cb = bp.coefficients(a), bp.coefficients(b)
        assert abs(var(ca) - (a - 0.03)) < 1e-9 and abs(cov(ca, cb) - (a - 0.03)) < 1e-9
        ids, co = bp.increment(b, a)  # order does not matter here (the sampler applies the sign)
        assert abs(sum(c * c for c in co) - 1.0) < 1e-6 and len(ids) <= 96 and ids == sorted(ids)
    assert bp.coefficients(0.03) == {} and set(bp.coefficients(14.6)) == {bp.ROOT}
    # outside [t_lo, t_hi] the two-sided motion continues: independent increments from the outermost known time, bridges in between
    out = bp.coefficients(20.0)
    far = bp.coefficients(25.0)
    mid = bp.coefficients(22.0)
    assert abs(var(out) - (20.0 - 0.03)) < 1e-9 and abs(cov(out, far) - (20.0 - 0.03)) < 1e-9 and abs(cov(mid, far) - (22.0 - 0.03)) < 1e-9
    below, lower = bp.coefficients(0.02), bp.coefficients(0.005)
    assert abs(var(below) - 0.01) < 1e-12 and abs(cov(below, lower) - 0.01) < 1e-12 and abs(var(lower) - 0.025) < 1e-12
    assert abs(cov(below, out)) < 1e-12  # increments on either side of t_lo are independent
    with pytest.raises(ValueError):
        bp.increment(1.0, 1.0)
    # a sampler's pattern -- times walking down from the top, two queries per step -- keeps every new point a bridge between the previous
    # one and an end of the interval; its expansion then holds every earlier normal (the device evaluates it from the kept neighbours)
    bp = ng.BrownianPath(0.03, 14.6)
    ts = [14.6 * 0.97**k for k in range(1, 150)]
    for k, t in enumerate(ts):
        c = bp.coefficients(t)
        a, b, fa, fb, sd, node = bp.bridge[t]
        assert (a, b) == (0.03, ts[k - 1] if k else 14.6) and abs(fa + fb - 1.0) < 1e-15 and c[node] == sd and len(c) == k + 2
        assert abs(var(c) - (t - 0.03)) < 1e-9
        if k:
            assert abs(cov(c, bp.coefficients(ts[k - 1])) - (t - 0.03)) < 1e-9
    assert bp.coefficients(ts[40]) == bp.coefficients(ts[40]) and bp.bridge[ts[40]][5] == 41  # a known time is never redefined
    # points cost O(1) to define; expansions are made on demand and only a bounded number is remembered
    long = ng.BrownianPath(0.0, 1.0)
    for k in range(1, 3000):
        long.define(1.0 - k / 3001.0)
    c = long.coefficients(1.0 - 2999 / 3001.0)
    assert len(c) == 3000 and abs(var(c) - (1.0 - 2999 / 3001.0)) < 1e-9 and len(long._memo) <= long.MEMO + 3000


def test_brownian_tree_mode_is_a_function_of_the_time_alone(pkg):
    """BrownianPath(tree_depth = D), the virtual Brownian tree samplers use by default (ComfyUI's BrownianTree up to its tolerance): a time is snapped to
    the grid of 2**D cells and defined through its dyadic ancestors, whose node ids are their places in the tree -- the expansion of
    W(t) is the same whatever was asked before (another order, other step counts, a fresh instance), still has a Brownian motion's
    variances and covariances exactly, and holds at most D + 1 normals."""
    import importlib
    import random

    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
    lo, hi, D = 0.03, 14.6, 24
    var = lambda c: sum(v * v for v in c.values())  # noqa: E731
    cov = lambda c1, c2: sum(v * c2.get(k, 0.0) for k, v in c1.items())  # noqa: E731
    rnd = random.Random(5)
    times = [rnd.uniform(lo, hi) for _ in range(60)] + [hi * 0.93**k for k in range(1, 40)]
    a, b = ng.BrownianPath(lo, hi, D), ng.BrownianPath(lo, hi, D)
    ca = {t: dict(a.coefficients(t)) for t in times}
    cb = {t: dict(b.coefficients(t)) for t in sorted(times)}          # another order ...
    fresh = {t: dict(ng.BrownianPath(lo, hi, D).coefficients(t)) for t in times[:10]}  # ... and no history at all
    assert all(ca[t] == cb[t] for t in times) and all(fresh[t] == ca[t] for t in times[:10])
    cells = 1 << D
    for t in times:
        ts = a.resolve(t)
        assert abs(ts - t) <= (hi - lo) / cells and a.resolve(ts) == ts
        assert 1 <= len(ca[t]) <= D + 1 and all(0 <= k < cells for k in ca[t])
        assert abs(var(ca[t]) - (ts - lo)) < 1e-9
    for s_, t in zip(times[:-1], times[1:]):
        assert abs(cov(ca[s_], ca[t]) - (min(a.resolve(s_), a.resolve(t)) - lo)) < 1e-9
        if a.resolve(s_) != a.resolve(t):
            ids, co = a.increment(s_, t)
            assert abs(sum(c * c for c in co) - 1.0) < 1e-6 and ids == sorted(ids) and len(ids) <= 2 * D + 2
    # the ends and their normals are the default's; a point's node id is its place in the tree
    assert a.coefficients(lo) == {} and set(a.coefficients(hi)) == {a.ROOT}
    mid = a.resolve((lo + hi) / 2)
    assert a.bridge[mid][5] == 1 and a.bridge[a.resolve(lo + (hi - lo) / 4)][5] == 2 and a.bridge[a.resolve(lo + 3 * (hi - lo) / 4)][5] == 3
    # outside the range: the default's extensions, with ids above the tree's
    out = a.coefficients(20.0)
    assert abs(var(out) - (20.0 - lo)) < 1e-9 and max(out) >= cells
    with pytest.raises(ValueError):
        a.increment(1.0, 1.0 + (hi - lo) / cells / 8)  # two times in one cell
    with pytest.raises(ValueError):
        ng.BrownianPath(lo, hi, 99)
    assert ng.BrownianPath(lo, hi).tree_depth == 0 and ng.BrownianPath(lo, hi).resolve(1.2345) == 1.2345
    # samplers are trees of depth 24 unless the environment says otherwise
    import os
    assert ng._env_tree_depth() == (24 if "SONAR_BROWNIAN_TREE" not in os.environ else ng._env_tree_depth())
    assert ng.BROWNIAN_TREE_DEPTH == ng._env_tree_depth()
    # a malformed or out-of-range value cannot make the package fail to import: the default applies
    saved = os.environ.get("SONAR_BROWNIAN_TREE")
    try:
        for text, want in (("banana", 24), ("99", 24), ("-3", 24), ("16", 16), ("off", 0), ("", 24)):
            os.environ["SONAR_BROWNIAN_TREE"] = text
            assert ng._env_tree_depth() == want, text
    finally:
        if saved is None:
            os.environ.pop("SONAR_BROWNIAN_TREE", None)
        else:
            os.environ["SONAR_BROWNIAN_TREE"] = saved
    # a time within half a grid cell beyond an end IS that end (a sigma_max handed over as float32 and queried as float64 differs in the
    # eighth digit); further out it stays an extension point of its own
    tree = ng.BrownianPath(0.03, 14.6, 24)
    cell = (14.6 - 0.03) / (1 << 24)
    assert tree.resolve(14.6 + 0.4 * cell) == 14.6 and tree.resolve(0.03 - 0.4 * cell) == 0.03
    assert tree.resolve(14.6 + 0.6 * cell) == 14.6 + 0.6 * cell and tree.resolve(0.03 - 2 * cell) == 0.03 - 2 * cell
    assert tree.grid_index(0.03) == 0 and tree.grid_index(14.6) == 1 << 24 and tree.grid_index((0.03 + 14.6) / 2) == 1 << 23
    assert ng.BrownianPath(0.03, 14.6, 0).resolve(14.6 + 0.4 * cell) == 14.6 + 0.4 * cell  # (the path of bridges snaps nothing)
    # grid times are floats: a range too narrow for its offset gets a shallower tree, and its grid points stay distinct
    narrow = ng.BrownianPath(1.0e6, 1.0e6 + 1e-3, 24)
    assert 1 <= narrow.tree_depth < 24
    pts = [narrow._grid_time(g) for g in range((1 << narrow.tree_depth) + 1)]
    assert all(b > a for a, b in zip(pts, pts[1:]))


def test_every_tagged_view_of_a_storage_loses_its_tag(pkg):
    """Two tensors on one storage may each carry a statistics tag; a kernel handed either of them (or any third view) must drop BOTH:
    raw-pointer kernels do not bump torch's version counter, so a surviving tag would normalise with statistics of other contents."""
    import importlib

    hl = pkg.hip_lib
    utils = importlib.import_module("comfyui_sonar_amd.py.utils")
    base = torch.zeros(4, 8)
    a, b = base[:2], base[2:]
    utils.attach_stats(a, torch.zeros(4, dtype=torch.float64))
    utils.attach_stats(b, torch.ones(4, dtype=torch.float64))
    assert len(hl.TAGGED[base.untyped_storage().data_ptr()]) == 2
    with pytest.raises(hl.SonarHipError):  # a CPU tensor is refused -- after the tags of its storage are gone
        hl._dev(base.view(-1), "x")
    assert utils.pop_stats(a) is None and utils.pop_stats(b) is None and not hl.TAGGED
    # a tensor that gives up its own tag leaves its sibling's alone; dead tensors leave the table
    utils.attach_stats(a, torch.zeros(4, dtype=torch.float64))
    utils.attach_stats(b, torch.ones(4, dtype=torch.float64))
    assert utils.pop_stats(a) is not None and utils.pop_stats(b) is not None and not hl.TAGGED
    utils.attach_stats(a, torch.zeros(4, dtype=torch.float64))
    del a
    assert not hl.TAGGED


def test_per_latent_sigma_must_have_one_value_or_one_per_latent(pkg):
    import importlib

    wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
    with pytest.raises(RuntimeError, match=r"size of tensor a \(4\) must match the size of tensor b \(3\)"):
        wc._per_latent(torch.zeros(4, 2, 8, 8), torch.ones(3), divide=True)


def test_brownian_path_depends_on_the_query_order(pkg):
    """The Brownian path is built point by point in query order (BrownianPath's docstring): the same times in the same order give the
    same expansion, another order gives another -- equally a Brownian motion (the variance / covariance test above holds for every
    order) -- so W(t) is a function of (seed, history), not of (seed, t) alone.  Pinned here so that a change of that contract is a
    decision, not an accident."""
    import importlib

    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
    times = [10.0, 7.0, 4.0, 8.5]
    a, b, c = ng.BrownianPath(0.03, 14.6), ng.BrownianPath(0.03, 14.6), ng.BrownianPath(0.03, 14.6)
    for t in times:
        a.coefficients(t)
        b.coefficients(t)
    for t in reversed(times):
        c.coefficients(t)
    assert all(a.coefficients(t) == b.coefficients(t) for t in times)          # same history: the same path
    assert any(a.coefficients(t) != c.coefficients(t) for t in times)          # another history: another path ...
    var = lambda co: sum(v * v for v in co.values())  # noqa: E731
    assert all(abs(var(p.coefficients(t)) - (t - 0.03)) < 1e-9 for p in (a, c) for t in times)  # ... of the same law
