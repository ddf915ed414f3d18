"""CPU: the host side of the prepared call plans (include/sonar_hip.h "prepared call plans", csrc/plan.hip) -- the level table of a
device-mode pyramid draw against the Python sequence it replaces, and the plan object's argument validation.  No kernel is launched."""
import ctypes as C
import importlib
import random

import pytest


def test_pyramid_levels_match_the_python_sequence(pkg):
    """sonar_pyramid_levels == PyramidNoiseGenerator._plan with _level_ratios(seed, stream) (py/noise_generation.py:609-649 semantics):
    sizes as integers, weights as the float32 of discount ** i."""
    hl = pkg.hip_lib
    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")

    class Gen:
        pass

    rnd = random.Random(7)
    for trial in range(3000):
        h, w = rnd.choice([1, 3, 8, 64, 96, 104, 128, 135, 256]), rnd.choice([2, 5, 8, 64, 96, 128, 152, 240, 256])
        g = Gen()
        g.iterations, g.discount = rnd.choice([0, 1, 2, 5, 10, 16]), rnd.choice([0.7, 0.5, 0.6, 0.8, 1.0, 0.33])
        seed = rnd.getrandbits(64) if trial % 2 else rnd.randrange(100)
        stream = rnd.randrange(1 << 40)
        want = [(None, ch, cw, C.c_float(g.discount**i).value) for i, ch, cw in ng.PyramidNoiseGenerator._plan(g, h, w, ng._level_ratios(seed, stream))]
        got = hl.AutoLevels(h, w, g.iterations, g.discount, seed, stream)
        assert list(got) == want
        assert got.rule == (h, w, g.iterations, g.discount, stream)


def test_plan_object_validates_its_records(pkg):
    hl = pkg.hip_lib
    lib = hl.load()
    assert lib.sonar_plan_fn_id(b"sonar_perlin_noise_f32") >= 0
    assert lib.sonar_plan_fn_id(b"sonar_max_to_host_f32") == -1  # hands a value to the host: not replayable
    assert lib.sonar_plan_fn_id(b"sonar_momentum_euler_f32") == -1  # writes `history_present` on the host
    fid = lib.sonar_plan_fn_id(b"sonar_axpby_f32")
    nargs = lib.sonar_plan_fn_nargs(fid)
    assert nargs == len(hl.SIGNATURES["sonar_axpby_f32"][1])
    plan = lib.sonar_plan_create(2)
    assert plan
    try:
        words = (C.c_uint64 * nargs)()
        ok = (hl.PlanPatch * 1)(hl.PlanPatch(hl.PATCH_SLOT, 0, 1, 8, 16))
        assert lib.sonar_plan_add(plan, fid, words, nargs, None, 0, C.cast(ok, C.c_void_p), 1) == 0
        assert lib.sonar_plan_length(plan) == 1
        for bad in (hl.PlanPatch(hl.PATCH_SLOT, 0, 2, 8, 0),          # slot 2 of 2
                    hl.PlanPatch(hl.PATCH_SLOT, nargs - 1, 0, 8, 0),  # the stream argument is the run's
                    hl.PlanPatch(hl.PATCH_STREAM, -1, 0, 8, 0),       # blob target without a blob
                    hl.PlanPatch(9, 0, 0, 8, 0)):                     # unknown source
            arr = (hl.PlanPatch * 1)(bad)
            assert lib.sonar_plan_add(plan, fid, words, nargs, None, 0, C.cast(arr, C.c_void_p), 1) == hl.ERR_ARG
        assert lib.sonar_plan_add(plan, fid, words, nargs - 1, None, 0, None, 0) == hl.ERR_ARG
        assert lib.sonar_plan_add(plan, 10_000, words, nargs, None, 0, None, 0) == hl.ERR_ARG
        assert lib.sonar_plan_length(plan) == 1
        failed = C.c_int(5)
        assert lib.sonar_plan_run(plan, (C.c_uint64 * 3)(), 3, 0, 0, None, failed) == hl.ERR_ARG  # slot table of the wrong size
    finally:
        lib.sonar_plan_destroy(plan)
    empty = lib.sonar_plan_create(0)
    failed = C.c_int(5)
    assert lib.sonar_plan_run(empty, None, 0, 1, 2, None, failed) == 0 and failed.value == -1  # nothing to issue
    lib.sonar_plan_destroy(empty)


def test_every_launch_only_entry_point_the_plans_name_exists(pkg):
    """The replayable set is a subset of the header's entry points whose last parameter is the stream."""
    hl = pkg.hip_lib
    lib = hl.load()
    names = [n for n in hl.SIGNATURES if lib.sonar_plan_fn_id(n.encode()) >= 0]
    assert len(names) >= 25
    for n in names:
        restype, argtypes = hl.SIGNATURES[n]
        assert restype is C.c_int and argtypes[-1] is C.c_void_p, n
        assert lib.sonar_plan_fn_nargs(lib.sonar_plan_fn_id(n.encode())) == len(argtypes), n


def test_blob_words(pkg):
    hl = pkg.hip_lib
    assert hl._float_word(1.0) == 0x3F800000 and hl._double_word(1.0) == 0x3FF0000000000000
    assert hl._float_word(0.1) == 0x3DCCCCCD  # the rounded float, as ctypes passes it
    blob = bytearray(b"abc")
    off = hl._blob_reserve(blob, 8, b"\x01\x02")
    assert off == 16 and len(blob) == 24 and blob[16:18] == b"\x01\x02" and blob[18:24] == bytes(6)


def test_a_trace_ending_on_one_thread_does_not_break_callers_on_another(pkg):
    """Round 6 (review item 7): `load()`, `_dev()` and `Planned._call` read the module's `_recorder` ONCE.  One thread starts and ends traces in
    a loop (the global flips between a recorder and None); the other keeps asking for the library and registering tags.  With two reads
    (`_recorder is None or _recorder.thread != ...`) the second could meet None: AttributeError.  Here the switch interval is set to its
    minimum so that the interleaving is exercised thousands of times."""
    import sys
    import threading

    import torch

    hl = pkg.hip_lib
    raw = hl.load()
    stop, errors, seen = threading.Event(), [], [0, 0]
    old = sys.getswitchinterval()
    sys.setswitchinterval(1e-6)

    def tracer():
        try:
            while not stop.is_set():
                result, plan = hl.trace_plan(lambda: 7, (), take=None, rewind=None, guards=())
                assert result == 7 and plan is None  # "the call launched nothing"
                seen[0] += 1
        except BaseException as exc:  # noqa: BLE001
            errors.append(exc)

    def caller():
        try:
            t = torch.zeros(4)
            planned_check = hl.Planned._call
            assert planned_check is not None
            while not stop.is_set():
                assert hl.load() is raw  # never the recording proxy: the trace belongs to the other thread
                hl.tag_register(t)
                hl.tag_forget(t)
                with pytest.raises(hl.SonarHipError):  # CPU tensor: refused, after the recorder check's code path is compiled in
                    hl._dev(t, "t")
                seen[1] += 1
        except BaseException as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=tracer), threading.Thread(target=caller)]
    try:
        for th in threads:
            th.start()
        import time

        time.sleep(1.5)
    finally:
        stop.set()
        for th in threads:
            th.join()
        sys.setswitchinterval(old)
    assert not errors, errors
    assert seen[0] > 50 and seen[1] > 50, seen


def test_every_reader_takes_the_recorder_global_once(pkg):
    """The deterministic half of the test above: no function of hip_lib other than trace_plan (which owns the global) loads `_recorder`
    more than once per call, so no reader can see a recorder in its test and None in its use."""
    import dis
    import types

    hl = pkg.hip_lib

    def functions(ns):
        for obj in ns.values():
            if isinstance(obj, types.FunctionType) and obj.__module__ == hl.__name__:
                yield obj
            elif isinstance(obj, type) and obj.__module__ == hl.__name__:
                yield from functions(vars(obj))

    def code_objects(code):
        yield code
        for const in code.co_consts:
            if isinstance(const, types.CodeType):
                yield from code_objects(const)

    readers = 0
    for fn in functions(vars(hl)):
        if fn.__name__ == "trace_plan":
            continue
        for code in code_objects(fn.__code__):
            loads = [i for i in dis.get_instructions(code) if i.opname in ("LOAD_GLOBAL", "LOAD_NAME") and i.argval == "_recorder"]
            assert len(loads) <= 1, f"{fn.__qualname__} reads _recorder {len(loads)} times"
            readers += len(loads)
    assert readers >= 4  # load, _dev, the power look-ahead wrapper, Planned._call
