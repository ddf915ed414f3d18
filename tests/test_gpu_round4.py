"""-m gpu, round 4: prepared call plans (include/sonar_hip.h "prepared call plans") -- a replayed step is the ordinary step bit for bit,
and every condition under which a plan must stand aside hands the call back to the ordinary path."""
import importlib
import math
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

SIG = (torch.tensor(14.6), torch.tensor(10.0))


@pytest.fixture(scope="module")
def api(pkg):
    pkg.hip_lib.load()
    return types.SimpleNamespace(hl=pkg.hip_lib, pn=importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise"),
                                 nz=importlib.import_module("comfyui_sonar_amd.py.noise"),
                                 ng=importlib.import_module("comfyui_sonar_amd.py.noise_generation"))


def _power(api, factor=1.0):
    return api.pn.PowerNoiseItem(factor, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0, mix=1.0,
                                 common_mode=0.0, channel_correlation="1,1,1,1,1,1")


def _maker(api, x, what, normalized=True):
    nz = api.nz
    if what == "power":
        return lambda: _power(api).make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=normalized)
    if what.startswith("chain:"):
        def make():
            chain = nz.CustomNoiseChain()
            for part in what[6:].split("+"):
                name, factor = part.split("*")
                chain.add(_power(api, float(factor)) if name == "power" else nz.CustomNoiseItem(float(factor), noise_type=name))
            return chain.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=normalized)
        return make
    return lambda: nz.get_noise_sampler(what, x, 0.03, 14.6, seed=None, cpu=False, normalized=normalized, factor=1.0 if normalized else 0.7)


def _planned(api, ns):
    return ns if isinstance(ns, api.hl.Planned) else getattr(ns, "_planned", None)


def _run(api, ns, calls, plans, seed=4321, fn=None):
    hl = api.hl
    old = hl.PLANS_ENABLED
    hl.PLANS_ENABLED = plans
    try:
        torch.manual_seed(seed)
        out = []
        for _ in range(calls):
            res = (fn or ns)(*SIG)
            res = res if isinstance(res, tuple) else (res,)
            out.append(tuple(None if t is None else t.clone() for t in res) + (getattr(res[0], hl.STATS_ATTR, None) is not None,))
        return out
    finally:
        hl.PLANS_ENABLED = old


def _same(a, b):
    return len(a) == len(b) and all(
        len(p) == len(q) and all((u is None and v is None) or (isinstance(u, bool) and u == v) or (torch.is_tensor(u) and torch.equal(u, v))
                                 for u, v in zip(p, q)) for p, q in zip(a, b))


CASES = ["gaussian", "uniform", "perlin", "pyramid", "pyramid_area", "power", "chain:perlin*0.5+pyramid*0.5", "chain:gaussian*0.6+perlin*0.4",
         "chain:power*0.5+perlin*0.3+gaussian*0.2", "chain:pyramid*1.0", "chain:uniform*0.3+pyramid*0.7"]


@pytest.mark.parametrize("shape", [(1, 4, 128, 128), (3, 4, 128, 128), (64, 4, 128, 128), (2, 16, 64, 64), (2, 4, 3, 64, 64)])
@pytest.mark.parametrize("what", CASES)
@pytest.mark.parametrize("normalized", [True, False])
def test_a_replayed_step_is_the_ordinary_step(api, shape, what, normalized):
    """Nine calls with plans (two ordinary, one traced, six replayed) == nine ordinary calls from the same RNG position: tensors, the
    statistics tags on them, and the (tensor, decision) pairs of the chains' deferred form."""
    if len(shape) == 5 and "power" in what:
        pytest.skip("video latents: the power item's planes are not [B, C*F] planes of the other generators")
    x = torch.zeros(shape, device="cuda")
    make = _maker(api, x, what, normalized)
    a, b = make(), make()
    assert _same(_run(api, a, 9, True), _run(api, b, 9, False))
    planned = _planned(api, a)
    assert planned is not None and planned.plan is not None, getattr(planned, "reason", "no Planned wrapper")
    assert planned.plan.runs == 6
    if isinstance(getattr(a, "deferred", None), api.hl.Planned):
        assert _same(_run(api, a, 7, True, fn=a.deferred), _run(api, b, 7, False, fn=b.deferred))
        assert a.deferred.plan is not None and a.deferred.plan.runs == 4


def test_plans_follow_the_rng_position(api):
    """A reseed, a draw by somebody else and torch.cuda.set_rng_state move the stream ids of the next call: the replay takes its ids from
    the same generator state as the ordinary path."""
    x = torch.zeros((4, 4, 128, 128), device="cuda")
    for what in ("chain:perlin*0.5+pyramid*0.5", "power", "pyramid", "perlin"):
        make = _maker(api, x, what)

        def script(ns, plans):
            api.hl.PLANS_ENABLED = plans
            try:
                got = []
                torch.manual_seed(9)
                for i in range(12):
                    if i == 5:
                        torch.manual_seed(10)
                    if i == 7:
                        torch.randn(3, device="cuda")  # another consumer of the device generator
                    if i == 9:
                        state = torch.cuda.get_rng_state()
                    if i == 11:
                        torch.cuda.set_rng_state(state)
                    got.append(ns(*SIG).clone())
                return got
            finally:
                api.hl.PLANS_ENABLED = True

        a, b = make(), make()
        ra, rb = script(a, True), script(b, False)
        assert all(torch.equal(p, q) for p, q in zip(ra, rb)), what
        assert torch.equal(ra[9], ra[11])
        assert _planned(api, a).plan.runs >= 6


def test_a_plan_stands_aside_when_its_guards_change(api):
    """Another shard position, another factor, plans switched off: the call is the ordinary path's (and still right)."""
    hl, ng, nz = api.hl, api.ng, api.nz
    x = torch.zeros((4, 4, 128, 128), device="cuda")
    ns = nz.get_noise_sampler("perlin", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    ref = nz.get_noise_sampler("perlin", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    torch.manual_seed(1)
    for _ in range(5):
        ns(*SIG)
    plan = ns._planned.plan
    assert plan is not None and plan.runs == 2
    with ng.shard_offset(8):
        torch.manual_seed(2)
        got = ns(*SIG).clone()
        assert plan.runs == 2  # guard: the traced call sat at shard offset 0
        old, hl.PLANS_ENABLED = hl.PLANS_ENABLED, False
        torch.manual_seed(2)
        want = ref(*SIG).clone()
        hl.PLANS_ENABLED = old
    assert torch.equal(got, want)
    ns(*SIG)
    assert plan.runs == 3
    ns.factor = 0.5
    torch.manual_seed(3)
    got = ns(*SIG).clone()
    assert plan.runs == 3
    ref.factor = 0.5
    hl.PLANS_ENABLED = False
    torch.manual_seed(3)
    want = ref(*SIG).clone()
    hl.PLANS_ENABLED = True
    assert torch.equal(got, want)


def test_a_refused_replay_rewinds_the_generator_and_takes_the_ordinary_path(api, monkeypatch):
    """An entry point that answers SONAR_ERR_UNSUPPORTED inside sonar_plan_run (a level table the plane kernel cannot hold, say): the step
    is issued again by the ordinary path from the SAME stream ids."""
    hl = api.hl
    x = torch.zeros((2, 4, 128, 128), device="cuda")
    make = _maker(api, x, "chain:perlin*0.5+pyramid*0.5")
    a, b = make(), make()
    want = _run(api, b, 8, False)
    real = hl._lib
    calls = {"n": 0}

    class Lib:
        def __getattr__(self, name):
            return getattr(real, name)

        def sonar_plan_run(self, *args):
            calls["n"] += 1
            if calls["n"] in (2, 3):
                return hl.ERR_UNSUPPORTED
            return real.sonar_plan_run(*args)

    monkeypatch.setattr(hl, "_lib", Lib())
    got = _run(api, a, 8, True)
    assert calls["n"] == 5 and _same(got, want)


def test_a_failing_replay_raises(api, monkeypatch):
    hl = api.hl
    x = torch.zeros((2, 4, 128, 128), device="cuda")
    ns = _maker(api, x, "gaussian")()
    for _ in range(4):
        ns(*SIG)
    real = hl._lib

    class Lib:
        def __getattr__(self, name):
            return getattr(real, name)

        def sonar_plan_run(self, *args):
            return hl.ERR_ARG

    monkeypatch.setattr(hl, "_lib", Lib())
    with pytest.raises(hl.SonarHipError, match="sonar_plan_run"):
        ns(*SIG)


def test_plans_keep_their_scratch_per_stream(api):
    """The same sampler called from two HIP streams in turn: each stream has its own scratch tensors (partials, lattice), the power item's
    look-ahead statistics are only trusted on the stream that wrote them -- and the values do not care."""
    x = torch.zeros((8, 4, 128, 128), device="cuda")
    side = torch.cuda.Stream()
    for what in ("chain:perlin*0.5+pyramid*0.5", "power"):
        make = _maker(api, x, what)
        a, b = make(), make()
        want = [t[0] for t in _run(api, b, 10, False)]
        torch.manual_seed(4321)
        got = []
        for i in range(10):
            if i % 3 == 2:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    got.append(a(*SIG).clone())
                torch.cuda.current_stream().wait_stream(side)
            else:
                got.append(a(*SIG).clone())
        torch.cuda.synchronize()
        assert all(torch.equal(p, q) for p, q in zip(got, want)), what
        plan = _planned(api, a).plan
        assert plan is not None and len(plan.by_stream) == 2


def test_steps_that_depend_on_their_arguments_are_not_planned(api):
    """Brownian noise (a function of the two sigmas), replay-mode generators (host draws) and scheduled noise keep the ordinary path."""
    nz, hl = api.nz, api.hl
    x = torch.zeros((2, 4, 64, 64), device="cuda")
    chain = nz.CustomNoiseChain()
    chain.add(nz.CustomNoiseItem(0.5, noise_type="perlin"))
    chain.add(nz.CustomNoiseItem(0.5, noise_type="brownian"))
    ns = chain.make_noise_sampler(x, 0.03, 14.6, seed=3, cpu=False, normalized=True)
    assert not isinstance(ns, hl.Planned)
    replay = nz.get_noise_sampler("perlin", x, 0.03, 14.6, seed=None, cpu=True, normalized=True)
    assert replay._planned is None
    sched = nz.CustomNoiseChain()
    inner = nz.CustomNoiseChain()
    inner.add(nz.CustomNoiseItem(1.0, noise_type="gaussian"))
    sched.add(nz.ScheduledNoise(1.0, noise=inner, start_sigma=20.0, end_sigma=0.0, normalize=None, fallback_noise=None))
    assert not isinstance(sched.make_noise_sampler(x, 0.03, 14.6, seed=None, cpu=False, normalized=True), hl.Planned)


def test_trace_refuses_what_it_cannot_replay(api):
    """A step that calls an entry point outside the replayable set, or hands a kernel an address nobody accounts for, gets no plan (and
    its traced call is still an ordinary, correct call)."""
    hl, ng = api.hl, api.ng
    x = torch.zeros((2, 4, 64, 64), device="cuda")

    def with_readback(_s, _sn):
        t = hl.philox_normal((2, 4, 64, 64), "cuda", *ng.DeviceRNG.take())
        hl.max_to_host(t)
        return t

    res, plan = hl.trace_plan(with_readback, SIG, take=ng.DeviceRNG.take, rewind=ng.DeviceRNG.rewind, guards=())
    assert plan is None and "not replayable" in hl.trace_plan.last_reason and res.shape == x.shape
    hidden = torch.zeros(2 * 4 * 64 * 64, device="cuda")

    def with_hidden_buffer(_s, _sn):
        lib = hl.load()
        seed, stream = ng.DeviceRNG.take()
        hl._check(lib.sonar_philox_normal_f32(hidden.data_ptr(), hidden.numel(), seed, stream, 0, None, hl._stream()), "fill")
        return hidden

    res, plan = hl.trace_plan(with_hidden_buffer, SIG, take=ng.DeviceRNG.take, rewind=ng.DeviceRNG.rewind, guards=())
    assert plan is None and "no tensor the trace knows" in hl.trace_plan.last_reason
    keep = []

    def with_escaping_scratch(_s, _sn):
        t = hl.philox_normal((2, 4, 64, 64), "cuda", *ng.DeviceRNG.take())
        keep.append(hl.stats(t))  # the partials outlive the call without being its result
        return t

    res, plan = hl.trace_plan(with_escaping_scratch, SIG, take=ng.DeviceRNG.take, rewind=ng.DeviceRNG.rewind, guards=())
    assert plan is None and "outlives" in hl.trace_plan.last_reason


def test_wavelet_kernels_take_tensors_at_odd_storage_offsets(api):
    """cond / uncond / x / out as contiguous views one float into their storage (4-byte aligned only): the low-pass and band kernels
    fall back from their 16- and 8-byte accesses instead of faulting or reading shifted data (advisor finding of round 3)."""
    hl = api.hl
    wf = importlib.import_module("comfyui_sonar_amd.py.wavelet_functions")
    w = wf.Wavelet(wave="db4", level=3, mode="symmetric")
    shape = (3, 4, 32, 32)
    n = 3 * 4 * 32 * 32
    torch.manual_seed(3)
    base = [torch.randn(n + 4, device="cuda") for _ in range(3)]
    aligned = [b[:n].clone().view(shape) for b in base]
    for off in (1, 2):
        odd = []
        for b in base:
            t = torch.empty(n + 4, device="cuda")
            t[off:off + n] = b[:n]
            odd.append(t[off:off + n].view(shape))
        assert all(t.data_ptr() % 16 == 4 * off for t in odd)
        for hp in (True, False):
            kw = dict(levels=3, dec_lo=w.dec_lo, rec_lo=w.rec_lo, mode="symmetric", inv_mode="symmetric", g=[3.0, 0.5, -1.0, 2.0], ku=1.0, kt=0.7,
                      subtract_from_x=True, high_precision=hp)
            assert torch.equal(hl.wcfg_lowpass(*odd, **kw), hl.wcfg_lowpass(*aligned, **kw))
            kb = dict(levels=3, dec_lo=w.dec_lo, dec_hi=w.dec_hi, rec_lo=w.rec_lo, rec_hi=w.rec_hi, mode="symmetric", inv_mode="symmetric",
                      yh_scales=[[3.0, 2.5, 2.0], [1.0, 0.5, 0.25], [2.0, 2.0, 1.0]], yl_scale=1.5, ku=1.0, kt=0.7, subtract_from_x=True, high_precision=hp)
            out_odd = torch.empty(n + 4, device="cuda")[off:off + n].view(shape)
            assert torch.equal(hl.wcfg_bands(odd[0], odd[1], odd[2], out_odd, **kb), hl.wcfg_bands(*aligned, **kb))


def test_folded_multiplies_are_the_multiply_passes(api):
    """y * a + x * b in the accumulation kernel == multiply pass, multiply pass, add pass, bit for bit, for multipliers that are not powers
    of two (MixedNoiseGenerator and the chains fold their factors this way; advisor finding of round 3) -- and the mixed presets draw
    one part at a time."""
    hl, nz = api.hl, api.nz
    torch.manual_seed(8)
    y, x = torch.randn(2, 4, 128, 128, device="cuda") * 3, torch.randn(2, 4, 128, 128, device="cuda")
    for a, b in ((0.3, 0.7), (1.0, -0.8), (0.55, 1.0), (1.15, 0.2), (-0.3, 1e-3)):
        fused = hl.axpby_(y.clone(), a, x, b)
        seq = hl.axpby_(hl.scale_noise_(y.clone(), a, False, None) if a != 1.0 else y.clone(), 1.0,
                        hl.scale_noise_(x.clone(), b, False, None) if b != 1.0 else x.clone(), 1.0)
        assert torch.equal(fused, seq), (a, b)
        with_stats, part = hl.axpby_stats_(y.clone(), a, x, b)
        assert torch.equal(with_stats, seq)
        torch.testing.assert_close(part.view(-1, 2).sum(0)[0], seq.double().sum(), rtol=1e-9, atol=1e-6)
    # pyramid_mix = 0.2 * pyramid - 0.8 * pyramid, rainbow_mild = (0.55 g + 0.7 g) * 1.15: the same values from explicit passes
    lat = torch.zeros((2, 4, 64, 64), device="cuda")
    for name in ("pyramid_mix", "rainbow_mild", "onef_pinkishgreenish"):
        torch.manual_seed(21)
        got = nz.get_noise_sampler(name, lat, 0.03, 14.6, seed=None, cpu=False, normalized=False)(*SIG)
        torch.manual_seed(21)
        sampler = nz.get_noise_sampler(name, lat, 0.03, 14.6, seed=None, cpu=False, normalized=False)
        gen = sampler.noise_sampler
        total = None
        for sub, transform in gen.ng_list:
            part = sub(*SIG)
            part = transform(part) if transform is not None else part
            total = part if total is None else hl.axpby_(total, 1.0, part, 1.0)
        total = gen.output_fun(total) if gen.output_fun is not None else total
        torch.testing.assert_close(got, total, rtol=0, atol=0)


def test_a_trace_only_sees_its_own_thread(api):
    """While one thread traces a step into a plan another thread (ComfyUI's preview thread, say) keeps calling the library: its calls
    and allocations are not the trace's, and both get their usual results."""
    import threading

    hl, nz = api.hl, api.nz
    x = torch.zeros((2, 4, 64, 64), device="cuda")
    ns = nz.get_noise_sampler("perlin", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    ref = nz.get_noise_sampler("perlin", x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    stop, errors, side = threading.Event(), [], []

    def other():
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                t = torch.randn(4096, device="cuda")
                while not stop.is_set():
                    side.append(float(hl.stats_finalize(hl.stats(t), t.numel())[0].item()))
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    th = threading.Thread(target=other)
    th.start()
    try:
        while len(side) < 3 and not errors:
            pass
        got = _run(api, ns, 8, True)
    finally:
        stop.set()
        th.join()
    assert not errors and len(set(side)) == 1
    assert ns._planned.plan is not None and hl.load().sonar_plan_length(ns._planned.plan.handle) == 1  # lattice + call fused (look-ahead)
    assert _same(got, _run(api, ref, 8, False))


@pytest.mark.parametrize("what", ["chain:perlin*0.5+pyramid*0.5", "power", "pyramid", "pyramid+ahead"])
def test_two_threads_replay_the_same_sampler(api, what):
    """Round 5: two threads call the SAME planned sampler at once, each on its own HIP stream (a preview thread beside the sampling thread).
    sonar_plan_run patches a per-call copy of the argument words and level tables (it used to patch the shared records), and a sampler
    serialises its callers: every call gets its own RNG position and the values that position gives -- the 2 x 12 results are, as a
    multiset, the 24 results of one thread calling 24 times from the same position."""
    import hashlib
    import threading

    hl = api.hl
    x = torch.zeros((4, 4, 128, 128), device="cuda")
    # "pyramid": the plan without hooks (round 6 gave the normalised pyramid call a look-ahead hook: switched off for this case, which is
    # about sonar_plan_run itself running on two threads at once); "pyramid+ahead": the same sampler as it is planned by default
    ahead_before = hl.PYRAMID_AHEAD
    hl.PYRAMID_AHEAD = what != "pyramid"
    what = "pyramid" if what == "pyramid+ahead" else what
    hookless = not hl.PYRAMID_AHEAD
    make = _maker(api, x, what, True)
    ns, ref = make(), make()
    old = hl.PLANS_ENABLED
    hl.PLANS_ENABLED = True
    try:
        torch.manual_seed(99)
        for _ in range(6):  # warm calls + the traced one: the plan exists before the threads start
            ns(*SIG)
        planned = _planned(api, ns)
        assert planned is not None and planned.plan is not None, getattr(planned, "reason", None)
        runs_before = planned.plan.runs
        direct = not planned.plan.hooks
        assert direct == hookless
        state = torch.cuda.get_rng_state()
        results, errors = {0: [], 1: []}, []
        barrier = threading.Barrier(2)

        def worker(k):
            try:
                torch.cuda.set_device(0)
                with torch.cuda.stream(torch.cuda.Stream()):
                    barrier.wait()
                    for _ in range(12):
                        # a plan without hooks keeps nothing between calls: its replay needs no lock at all -- sonar_plan_run itself runs
                        # concurrently on the two threads then (the pyramid's plan patches level tables in its records' blobs)
                        out = planned.plan.run() if direct else ns(*SIG)
                        assert out is not hl.NOT_RUN
                        out = out[0] if isinstance(out, tuple) else out
                        torch.cuda.current_stream().synchronize()
                        results[k].append(hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest())
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        if what != "power":  # (the power item's look-ahead statistics are tied to ONE HIP stream: two streams taking turns send every call down
            # the ordinary path -- right values, no replay)
            assert planned.plan.runs - runs_before >= 20  # replays, not fall-backs
        torch.cuda.set_rng_state(state)
        hl.PLANS_ENABLED = False
        single = []
        for _ in range(24):
            out = ref(*SIG)
            out = out[0] if isinstance(out, tuple) else out
            single.append(hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest())
        assert len(set(single)) == 24
        assert sorted(results[0] + results[1]) == sorted(single)
    finally:
        hl.PLANS_ENABLED = old
        hl.PYRAMID_AHEAD = ahead_before


REGISTRY_TYPES = ["gaussian", "uniform", "perlin", "pyramid", "pyramid_area", "pyramid_discount5", "pyramid_mix", "pyramid_mix_area", "pyramid_old",
                  "pyramid_old_area", "highres_pyramid", "laplacian", "power_old", "pink_old", "white", "grey", "velvet", "violet", "onef_pinkish",
                  "onef_greenish", "onef_pinkishgreenish", "onef_pinkish_mix", "onef_greenish_mix", "green_test", "rainbow_mild", "rainbow_intense",
                  "studentt", "wavelet", "brownian"]
NOT_STATIC = {"highres_pyramid", "studentt", "wavelet", "brownian"}  # host draws per call / torch arithmetic / functions of the sigmas


@pytest.mark.parametrize("normalized", [True, False])
@pytest.mark.parametrize("name", REGISTRY_TYPES)
def test_every_registry_type_replays_or_declines(api, name, normalized):
    """Every registry type the path runs, at 2 latents: the types whose step is a function of the RNG position get a plan and the replayed
    step is the ordinary step bit for bit (NaN where the reference's own arithmetic gives NaN); the others keep the ordinary path."""
    x = torch.zeros((2, 4, 64, 64), device="cuda")
    make = lambda: api.nz.get_noise_sampler(name, x, 0.03, 14.6, seed=5, cpu=False, normalized=normalized)  # noqa: E731
    a, b = make(), make()
    ra, rb = _run(api, a, 7, True), _run(api, b, 7, False)
    for p, q in zip(ra, rb):
        assert torch.equal(torch.isnan(p[0]), torch.isnan(q[0])) and torch.equal(torch.nan_to_num(p[0]), torch.nan_to_num(q[0]))
        assert p[-1] == q[-1]
    planned = _planned(api, a)
    if name in NOT_STATIC:
        assert planned is None
    else:
        assert planned is not None and planned.plan is not None, getattr(planned, "reason", None)
        assert planned.plan.runs == 4


def test_torch_work_inside_a_step_means_no_plan(api):
    """A step that runs a torch kernel of its own on a device tensor (here: an in-place multiply between two entry points) cannot be
    replayed from the recorded entry points alone: the trace sees the operation and declines."""
    hl, ng = api.hl, api.ng

    def step(_s, _sn):
        t = hl.philox_normal((2, 4, 64, 64), "cuda", *ng.DeviceRNG.take())
        t.mul_(2.0)
        return hl.scale_noise_(t, 0.5, False, None)

    res, plan = hl.trace_plan(step, SIG, take=ng.DeviceRNG.take, rewind=ng.DeviceRNG.rewind, guards=())
    assert plan is None and "torch operation aten.mul_" in hl.trace_plan.last_reason and res.shape == (2, 4, 64, 64)

    def with_item(_s, _sn):
        t = hl.philox_normal((2, 4, 64, 64), "cuda", *ng.DeviceRNG.take())
        return hl.scale_noise_(t, float(t[0, 0, 0, 0].item()), False, None)

    _res, plan = hl.trace_plan(with_item, SIG, take=ng.DeviceRNG.take, rewind=ng.DeviceRNG.rewind, guards=())
    assert plan is None and "torch operation" in hl.trace_plan.last_reason


@pytest.mark.parametrize("shape", [(1, 4, 128, 128), (4, 4, 128, 128), (64, 4, 128, 128), (2, 16, 64, 64), (2, 4, 3, 64, 64)])
def test_a_planned_perlin_call_is_one_launch(api, shape):
    """Inside a plan a normalised Perlin call is ONE launch in the steady state (sonar_perlin_noise_ahead_f32: this call's final pass, the
    next call's statistics pass, the lattice of the call after it): same bits as lattice + statistics + final launch, whatever the hook
    finds or misses -- a reseed and a foreign draw in the middle of the run cost shortcuts, not values."""
    hl = api.hl
    x = torch.zeros(shape, device="cuda")
    make = _maker(api, x, "perlin")
    a, b = make(), make()

    def script(ns, plans):
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
            hl.PLANS_ENABLED = True

    ra, rb = script(a, True), script(b, False)
    assert all(torch.equal(p, q) for p, q in zip(ra, rb))
    plan = a._planned.plan
    assert plan is not None and hl.load().sonar_plan_length(plan.handle) == 1
    hook = plan.hooks[0]
    assert isinstance(hook, hl._PerlinAheadHook) and hook.hits >= 6 and hook.misses >= 3  # first run, the reseed, the foreign draw
    # the entry point itself: every combination of what a call may find, against the three launches it replaces
    lib = hl.load()
    b_, c, h, w = (shape[0], math.prod(shape[1:-2]), shape[-2], shape[-1])
    chw = c * h * w
    assert lib.sonar_perlin_noise_ahead_ok(b_, chw, 0) == 1
    st = hl._stream()
    lat = lambda stream: hl.perlin_lattice(2, c, h, w, "cuda", "lerp", 9, stream)  # noqa: E731
    t0, t1 = lat(101), lat(103)
    want0 = hl.perlin_noise((b_, c, h, w), t0, 2.0, 9, 100, 0, 0.9)
    want1 = hl.perlin_noise((b_, c, h, w), t1, 2.0, 9, 102, 0, 0.9)
    for have_stats in (0, 1):
        for with_next in (0, 1):
            out0, out1 = torch.empty((b_, c, h, w), device="cuda"), torch.empty((b_, c, h, w), device="cuda")
            p0, p1, p2 = hl.new_partials("cuda"), hl.new_partials("cuda"), hl.new_partials("cuda")
            if have_stats:
                hl.perlin_generate((b_, c, h, w), t0, 2.0, 9, 100, 0, partials=None)  # (no statistics-only entry point: take them from a dry call)
                ws = hl.new_partials("cuda")
                assert lib.sonar_perlin_noise_ahead_f32(t0.data_ptr(), out0.data_ptr(), b_, chw, 2.0, 9, 100, 0, 0.9, 2.5, ws.data_ptr(), 0, 100, t0.data_ptr(),
                                                        p0.data_ptr(), None, 0, 0, 0, 0, 0, 0, st) == 0  # leaves call 100's statistics in p0
            t_out = torch.empty_like(t1)
            rc = lib.sonar_perlin_noise_ahead_f32(t0.data_ptr(), out0.data_ptr(), b_, chw, 2.0, 9, 100, 0, 0.9, 2.5, p0.data_ptr(), have_stats, 102,
                                                  t1.data_ptr() if with_next else None, p1.data_ptr() if with_next else None, t_out.data_ptr(), 2, c, h, w, 0,
                                                  103, st)
            assert rc == 0 and torch.equal(out0, want0) and torch.equal(t_out, t1)
            rc = lib.sonar_perlin_noise_ahead_f32(t1.data_ptr(), out1.data_ptr(), b_, chw, 2.0, 9, 102, 0, 0.9, 2.5, (p1 if with_next else p2).data_ptr(),
                                                  with_next, 104, None, None, None, 0, 0, 0, 0, 0, 0, st)
            assert rc == 0 and torch.equal(out1, want1)
    # (round 5: no upper size limit any more -- at 512 latents the next call's statistics ride under the final pass's stores)
    assert lib.sonar_perlin_noise_ahead_ok(b_, chw + 4, 0) == 0 and lib.sonar_perlin_noise_ahead_ok(8192, 65536, 0) == 1


# ------------------------------------------------------------------------------------------------ planes beyond LDS, generated (kind 4)
@pytest.mark.parametrize("shape", [(8, 4, 256, 256), (2, 4, 256, 256), (1, 4, 512, 512), (3, 1, 384, 512), (2, 3, 320, 256), (1, 2, 64, 2048), (1, 4, 512, 4),
                                   (70, 4, 256, 256)])
def test_planes_beyond_lds_are_generated_in_column_blocks(api, shape):
    """Kind-4 planes (2048 px latents: 256 x 256): the spectrum is drawn in blocks of columns, filtered and column-transformed into a complex
    workspace, the rows come out of it (sonar_power_block_f32).  The tensor is irfft2(drawn spectrum x filter) -- the spectrum itself is mode
    2's dump --, the normalised call (Parseval statistics first, no pass over the tensor) is the raw tensor normalised, the statistics of
    mode 0 are the tensor's, and two shards of a batch are the batch."""
    hl = api.hl
    lib = hl.load()
    b, c, H, W = shape
    K = W // 2 + 1
    planes = b * c
    if hl.power_plane_kind(H, W) != 4:
        assert (H, W) == (512, 4)  # fits LDS: the general-size kernels' plane
        return
    assert lib.sonar_power_block_ws_bytes(planes, H, W) == planes * H * K * 8 and lib.sonar_power_block_ws_bytes(1, 128, 128) == -1
    g = torch.Generator(device="cuda").manual_seed(5)
    filt = torch.rand(H, K, device="cuda", generator=g) + 0.25
    spec = hl.power_spectrum(shape, "cuda", seed=7, stream_id=3)
    assert tuple(spec.shape) == (b, c, H, K)
    mag = (spec.real.double() ** 2 + spec.imag.double() ** 2)
    # every column of every block is drawn: an undrawn column would be H zeros; a drawn value is zero with probability 2^-23 (u = 1)
    assert int((mag == 0).sum().item()) <= 2 + 10 * mag.numel() * 2.0 ** -23
    n = spec.numel()
    assert abs(mag.mean().item() - 1.0) < 6.0 / n ** 0.5 and abs(spec.real.double().mean().item()) < 4.0 / n ** 0.5  # unit complex normals
    part = hl.new_partials("cuda")
    raw = hl.power_irfft2(None, filt, shape, seed=7, stream_id=3, plane_offset=0, partials=part)
    want = torch.fft.irfft2(spec * filt, s=(H, W), norm="ortho")
    torch.testing.assert_close(raw, want, rtol=0, atol=2e-5 * float(want.abs().max()))
    d = raw.double()
    torch.testing.assert_close(part.view(-1, 2).sum(0), torch.stack([d.sum(), (d * d).sum()]), rtol=1e-9, atol=1e-6)
    assert torch.equal(raw, hl.power_irfft2(None, filt, shape, seed=7, stream_id=3, plane_offset=0))
    norm = hl.power_noise(filt, shape, seed=7, stream_id=3, plane_offset=0, factor=0.75)
    # scale_noise's rule (a mean inside its band is left alone), by the tensor's own statistics in place of the draw's Parseval sums
    torch.testing.assert_close(norm, hl.scale_noise_(raw.clone(), 0.75, True, part), rtol=2e-5, atol=4e-6)
    assert abs(norm.double().std().item() - 0.75) < 1e-4
    if b % 2 == 0:
        h = b // 2
        lo = hl.power_irfft2(None, filt, (h, c, H, W), seed=7, stream_id=3, plane_offset=0)
        hi = hl.power_irfft2(None, filt, (h, c, H, W), seed=7, stream_id=3, plane_offset=h * c)
        assert torch.equal(torch.cat([lo, hi]), raw)
    other = hl.power_irfft2(None, filt, shape, seed=7, stream_id=4, plane_offset=0)
    assert abs(torch.corrcoef(torch.stack([raw.flatten(), other.flatten()]))[0, 1].item()) < 5.0 / raw.numel() ** 0.5 + 1e-3


def test_block_planes_refusals(api):
    hl = api.hl
    lib = hl.load()
    st = hl._stream()
    filt = torch.ones(256, 129, device="cuda")
    ws = torch.empty(4, 256, 129, dtype=torch.complex64, device="cuda")
    out = torch.full((4, 256, 256), 7.0, device="cuda")
    part = hl.new_partials("cuda")
    args = (filt.data_ptr(), ws.data_ptr(), out.data_ptr(), 4, 256, 256, 1, 2, 0, 4)
    assert lib.sonar_power_block_f32(*args, 1, 1.0, 2.5, None, st) == hl.ERR_ARG  # the normalised mode needs its statistics workspace
    assert lib.sonar_power_block_f32(*args, 3, 1.0, 2.5, part.data_ptr(), st) == hl.ERR_ARG
    assert lib.sonar_power_block_f32(filt.data_ptr(), ws.data_ptr(), out.data_ptr(), 4, 256, 256, 1, 2, 0, 3, 0, 1.0, 2.5, None, st) == hl.ERR_ARG
    assert lib.sonar_power_block_f32(filt.data_ptr(), ws.data_ptr(), out.data_ptr(), 4, 128, 128, 1, 2, 0, 4, 0, 1.0, 2.5, None, st) == hl.ERR_UNSUPPORTED
    assert lib.sonar_power_noise_f32(filt.data_ptr(), out.data_ptr(), 4, 256, 256, 1, 2, 0, 4, 1.0, 2.5, part.data_ptr(), st) == hl.ERR_UNSUPPORTED
    torch.cuda.synchronize()
    assert bool((out == 7.0).all())


def test_power_sampler_on_a_2048px_latent(api):
    """The power-law item on a 256 x 256 latent (kind 4) from the node down: unit statistics, a plan after the warm calls with the same bits
    as the ordinary path, and an isotropic 1/f spectrum (the filter reaches the draw: low radii carry more power than high ones)."""
    hl = api.hl
    x = torch.zeros(2, 4, 256, 256, device="cuda")
    item = _power(api)

    def run(plans):
        hl.PLANS_ENABLED = plans
        try:
            torch.manual_seed(77)
            ns = item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)
            return [ns(None, None).clone() for _ in range(6)]
        finally:
            hl.PLANS_ENABLED = True

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    t = a[-1].double()
    assert abs(t.std().item() - 1.0) < 1e-3 and abs(t.mean().item()) < 1e-3 and not torch.equal(a[0], a[1])
    p = torch.fft.rfft2(a[-1], norm="ortho").abs().pow(2).mean((0, 1))
    assert p[2:6, 2:6].mean().item() > 4.0 * p[60:100, 60:100].mean().item()
