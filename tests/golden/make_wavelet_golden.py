"""Golden vectors for the wavelet rows (W scaling / blend, WC, WF -- SURVEY.md §8a) from the REAL reference.

Container-only tool (like make_golden.py).  The reference's own Python is imported from /root/reference and run
end to end: ``expand_yh_scales / wavelet_scaling / wavelet_blend`` (py/wavelet_functions.py:148-238), the schedule /
percentage / rule logic and ``WaveletCFG.__call__`` (py/wavelet_cfg.py:33-842), ``Wavelet.forward / inverse``
(py/wavelet_functions.py:81-105) and ``WaveletFilteredNoiseGenerator`` / ``WaveletFilteredNoise``
(py/noise_generation.py:1908-2032, py/noise.py:1521-1593).  The one thing the image lacks, ``pytorch_wavelets``, is
supplied by tests/golden/pywt_bridge.py: constructor-compatible transform objects whose arithmetic is the real
PyWavelets 1.1.1 (run in the interpreter that has it).  Nothing from oracle/ or the product takes part.

    python tests/golden/make_wavelet_golden.py

Writes wavelet_scaling.npz, wcfg_host.json, wavelet_cfg.npz, wavelet_filtered.npz.  fp64 cases are exact PyWavelets
arithmetic; ``high_precision_mode=False`` / noise cases are PyWavelets' fp32 arithmetic (its own summation order).
"""
from __future__ import annotations

import importlib
import json
import math
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import pywt_bridge  # noqa: E402
import wavelet_cases as wc  # noqa: E402
from oracle.ref_import import load_reference  # noqa: E402

ref = load_reference()
taps = json.load(open(os.path.join(ROOT, "comfyui-sonar_amd", "wavelet_taps.json")))["wavelets"]
pywt_bridge.install(ref.wavelet_functions, wavelist=tuple(taps))
rwf = ref.wavelet_functions
rcfg = importlib.import_module("sonar_ref.wavelet_cfg")
assert pywt_bridge.pywt_version() == "1.1.1"
G = np.load(os.path.join(HERE, "dwt.npz"), allow_pickle=False)
SIG = (torch.tensor(14.6), torch.tensor(10.0))


def save(name, **arrays):
    conv = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def jsonable(v):
    if isinstance(v, float):
        if math.isnan(v):
            return "nan"
        if math.isinf(v):
            return "inf" if v > 0 else "-inf"
        return v
    if hasattr(v, "_asdict"):  # NamedTuple: by field name
        return {k: jsonable(i) for k, i in v._asdict().items()}
    if isinstance(v, (tuple, list)):
        return [jsonable(i) for i in v]
    if isinstance(v, dict):
        return {k: jsonable(i) for k, i in v.items()}
    if hasattr(v, "name") and hasattr(v, "value"):  # Enum
        return v.name
    return v


# ------------------------------------------------------------------------------------------------ (a) scaling + blend
def gen_scaling():
    out = {}
    tables = {}
    for name, (tag, yl_scale, yh_scales) in wc.SCALING_CASES.items():
        _, _, yl, yh = wc.dwt_case(G, tag)
        tables[name] = jsonable(rwf.expand_yh_scales(yh, yh_scales=1.0 if yh_scales is None else yh_scales))
        keep_l, keep_h = yl.clone(), [b.clone() for b in yh]
        rl, rh = rwf.wavelet_scaling(yl, yh, yl_scale, yh_scales)                    # out of place: inputs untouched
        assert torch.equal(yl, keep_l) and all(torch.equal(a, b) for a, b in zip(yh, keep_h))
        il, ih = rwf.wavelet_scaling(yl, yh, yl_scale, yh_scales, in_place=True)     # in place: same storage, same values
        assert il is yl and all(a is b for a, b in zip(ih, yh))
        assert torch.equal(il, rl) and all(torch.equal(a, b) for a, b in zip(ih, rh))
        out[f"{name}__yl"] = rl
        for j, band in enumerate(rh):
            out[f"{name}__yh{j}"] = band
    errors = {}
    for name, (tag, yh_scales) in wc.SCALING_ERRORS.items():
        _, _, yl, yh = wc.dwt_case(G, tag)
        try:
            rwf.expand_yh_scales(yh, yh_scales=yh_scales)
            raise SystemExit(f"{name}: expected an error")
        except ValueError as exc:
            errors[name] = [type(exc).__name__, str(exc)]
    for name, (tag, fn_l, fn_h, fac_l, fac_h) in wc.BLEND_CASES.items():
        _, _, yl, yh = wc.dwt_case(G, tag)
        bl, bh = wc.second_coeffs(yl, yh)
        modes = ref.utils.BLENDING_MODES
        rl, rh = rwf.wavelet_blend((yl, yh), (bl, bh), yl_factor=fac_l, yh_factor=fac_h, blend_function=modes[fn_l],
                                   yh_blend_function=None if fn_h is None else modes[fn_h])
        out[f"blend_{name}__yl"] = rl
        for j, band in enumerate(rh):
            out[f"blend_{name}__yh{j}"] = band
    out["tables_json"] = np.array(json.dumps(tables))
    out["errors_json"] = np.array(json.dumps(errors))
    save("wavelet_scaling", **out)


# ------------------------------------------------------------------------------------------------ (b) host schedule logic
def sample_sigmas(key):
    if key == "ascending":
        return torch.tensor([1.0, 2.0, 3.0, 0.0])
    if key == "bad_ndim":
        return torch.zeros(2, 2, 2)
    return wc.SAMPLE_SIGMAS[key]


class FakeYh:
    def __init__(self, norient):
        self.shape = (1, 1, norient, 1, 1) if norient > 1 else (1, 1, 1)


def gen_host():
    res = {}
    res["interp"] = {name: [getattr(rcfg.WCFGSchedule, name.upper()).interp(v) for v in wc.INTERP_GRID] for name in wc.SCHEDULES}
    ms = wc.DiscreteSampling()
    pcts = []
    for start, end, sigma, key in wc.PCT_CASES:
        try:
            pcts.append(rcfg.WCFGPercentages.build(ms=ms, start_sigma=start, end_sigma=end, sigma=sigma, sigmas=sample_sigmas(key)))
        except Exception as exc:  # noqa: BLE001 -- the reference leaves step_first / pct_enabled_steps unbound on some branches (:181-187)
            pcts.append({"error": [type(exc).__name__, str(exc)]})
            print("  pct row", len(pcts) - 1, (start, end, sigma, key), "->", type(exc).__name__, exc)
    res["pcts"] = [jsonable(p) for p in pcts]
    res["pct_inverted"] = [None if isinstance(p, dict) else jsonable(p.invert()) for p in pcts]
    errs = []
    for start, end, sigma, key in wc.PCT_ERRORS:
        try:
            rcfg.WCFGPercentages.build(ms=ms, start_sigma=start, end_sigma=end, sigma=sigma, sigmas=sample_sigmas(key))
            raise SystemExit("expected an error")
        except (ValueError, RuntimeError) as exc:
            errs.append([type(exc).__name__, str(exc)])
    res["pct_errors"] = errs
    good = [i for i, p in enumerate(pcts) if not isinstance(p, dict)]
    res["pct_good_rows"] = good
    rows = []
    for kw in wc.SCHED_SCALE_CASES:
        sched = rcfg.WCFGScheduledScale.build(**dict(kw))
        rows.append({"built": jsonable(sched), "values": [sched.get_b_scale(pcts[i]) for i in good]})
    res["sched_scale"] = rows
    # modes without a value: step mode with no sample sigmas -> RuntimeError; sigma modes -> "Couldn't get percentage"
    none_errs = {}
    no_sigmas = pcts[good[0]]._replace(pct_sigmas=None, pct_enabled_sigmas=None, pct_steps=None, pct_enabled_steps=None)
    for mode in ("step", "sigmas", "enabled_sigmas"):
        try:
            rcfg.WCFGScheduledScale.build(schedule_mode=mode).get_b_scale(no_sigmas)
            raise SystemExit("expected an error")
        except RuntimeError as exc:
            none_errs[mode] = str(exc)
    res["sched_scale_errors"] = none_errs
    nb, no = wc.SCALES_RANGE_LEVELS
    yh = [FakeYh(no)] * nb
    sr = {}
    for name, kw in wc.SCALES_RANGE_CASES.items():
        built = rcfg.WCFGScalesRange.build(**json.loads(json.dumps(kw)))
        vals = []
        for i in good:
            sc = built.get_scales(pcts[i], yh)
            vals.append({"yl_scale": sc.yl_scale, "yh_scales": jsonable(sc.yh_scales)})
        sr[name] = {"type": type(built).__name__, "values": vals}
    res["scales_range"] = sr
    sf = []
    for spec in wc.SCHED_FLOAT_CASES:
        built = rcfg.WCFGScheduledFloat.build(spec)
        entry = {"built": jsonable(built)}
        if not isinstance(built.value_start, dict):
            entry["values"] = [built.get_value(pcts[i]) for i in good]
        sf.append(entry)
    res["sched_float"] = sf
    try:
        rcfg.WCFGScheduledFloat.build("0.5")
        raise SystemExit("expected an error")
    except TypeError as exc:
        res["sched_float_error"] = str(exc)
    rules = {}
    for name, spec in wc.RULES_CASES.items():
        built = rcfg.WCFGRules.build(**json.loads(json.dumps(spec["params"])))
        picks = []
        for s in spec["sigmas"]:
            rule = built.get_rule(s)
            picks.append(None if rule is None else built.rules.index(rule))
        rules[name] = {"n": len(built), "rules": [jsonable(r) for r in built.rules], "picks": picks}
    res["rules"] = rules
    for bad in ("blend_strength", "difference_blend_strength"):
        try:
            rcfg.WCFGRule.build(**{bad: "1.0"})
            raise SystemExit("expected an error")
        except TypeError as exc:
            res[f"rule_error_{bad}"] = str(exc)
    path = os.path.join(HERE, "wcfg_host.json")
    with open(path, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(f"wcfg_host.json  {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------------------------------------ (c) WaveletCFG end to end
def run_wcfg(mod, name, case, model):
    """One ``WaveletCFG.__call__`` of ``mod`` (the reference's module here; the tests pass the product's)."""
    args = wc.wcfg_inputs(case, name)
    args["model"] = model
    key = case.get("sample_sigmas", "karras12")  # without sample sigmas the reference cannot get past WCFGPercentages.build
    args["model_options"] = {} if key is None else {"transformer_options": {"sample_sigmas": wc.SAMPLE_SIGMAS[key]}}
    rules = mod.WCFGRules.build(**json.loads(json.dumps(case["params"])))
    ops = wc.wcfg_ops() if case.get("ops") else {}
    fn = mod.WaveletCFG(existing_cfg=wc.existing_cfg if case.get("existing") else None, rules=rules, **ops)
    return args, fn(args)


def gen_wcfg():
    out = {}
    model = wc.FakeModel()
    for name, case in wc.WCFG_CASES.items():
        args, res = run_wcfg(rcfg, name, case, model)
        assert res.dtype == torch.float32 and res.shape == args["input"].shape and res.is_contiguous(), name
        for k in ("input", "cond_denoised", "uncond_denoised", "sigma"):
            out[f"{name}__{k}"] = args[k]
        out[f"{name}__out"] = res
    errors = {}
    for name, case in wc.WCFG_ERRORS.items():
        try:
            run_wcfg(rcfg, name, case, model)
            raise SystemExit(f"{name}: expected an error")
        except (RuntimeError, UnboundLocalError) as exc:
            errors[name] = [type(exc).__name__, str(exc)]
    out["errors_json"] = np.array(json.dumps(errors))
    save("wavelet_cfg", **out)


# ------------------------------------------------------------------------------------------------ (d) wavelet-filtered noise
def gauss_chain(noise_mod):
    chain = noise_mod.CustomNoiseChain()
    chain.add(noise_mod.CustomNoiseItem(1.0, noise_type="gaussian"))
    return chain


def gen_wf():
    out = {}
    for name, case in wc.WF_GEN_CASES.items():
        shape = tuple(case["shape"])
        torch.manual_seed(9)
        gen = ref.noise_generation.WaveletFilteredNoiseGenerator(torch.zeros(shape), sigma_min=0.03, sigma_max=14.6, seed=9, cpu=True, normalized=False,
                                                                 **json.loads(json.dumps(case["kw"])))
        res = gen(*SIG)
        assert res.shape == shape and res.dtype == torch.float32, (name, res.shape, res.dtype)
        torch.manual_seed(9)
        out[f"gen_{name}__low"] = torch.randn(shape)
        out[f"gen_{name}__out"] = res
    for name, case in wc.WF_ITEM_CASES.items():
        shape = tuple(case["shape"])
        item = ref.noise.WaveletFilteredNoise(1.0, noise=gauss_chain(ref.noise), noise_high=gauss_chain(ref.noise) if case["high"] else None,
                                              normalize=None, normalize_noise=case["normalize_noise"], yaml_parameters=yaml.safe_dump(case["yaml"]))
        torch.manual_seed(12)
        res = item.make_noise_sampler(torch.zeros(shape), 0.03, 14.6, seed=12, cpu=True, normalized=case["normalized"])(*SIG)
        assert res.shape == shape
        out[f"item_{name}__out"] = res
    nodes = importlib.import_module("sonar_ref.nodes.noise_filters")
    node = nodes.SonarWaveletFilteredNoiseNode()
    nc = wc.WF_NODE_CASE
    (chain,) = node.go(factor=1.0, rescale=0.0, normalize="disabled", normalize_noise=False, custom_noise=gauss_chain(ref.noise),
                       yaml_parameters=nc["yaml"])
    torch.manual_seed(nc["seed"])
    out["node__out"] = chain.make_noise_sampler(torch.zeros(nc["shape"]), 0.03, 14.6, seed=nc["seed"], cpu=True, normalized=False)(*SIG)
    save("wavelet_filtered", **out)


def main():
    gen_scaling()
    gen_host()
    gen_wcfg()
    gen_wf()


if __name__ == "__main__":
    if "--only" in sys.argv:
        globals()["gen_" + sys.argv[sys.argv.index("--only") + 1]]()
    else:
        main()
