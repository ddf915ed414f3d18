"""Shared by the CPU oracle tests and the GPU parity tests of the wavelet rows: builds the inputs of a WCFG / WF case from
tests/golden/wavelet_cases.py and resolves a rule into the plain numbers oracle/dwt_oracle.py takes (through the product's
host logic, which tests/test_wavelet_host_cpu.py pins to the reference's own outputs)."""
import json

import numpy as np
import torch

from tests.golden import wavelet_cases as wc


class _Band:
    def __init__(self, norient):
        self.shape = (1, 1, norient, 1, 1) if norient > 1 else (1, 1, 1)


def wcfg_args(case, name, model, device=None):
    args = wc.wcfg_inputs(case, name)
    if device is not None:
        args = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in args.items()}
    args["model"] = model
    key = case.get("sample_sigmas", "karras12")
    args["model_options"] = {} if key is None else {"transformer_options": {"sample_sigmas": wc.SAMPLE_SIGMAS[key]}}
    return args


def build_wcfg(mod, case):
    rules = mod.WCFGRules.build(**json.loads(json.dumps(case["params"])))
    ops = wc.wcfg_ops() if case.get("ops") else {}
    return mod.WaveletCFG(existing_cfg=wc.existing_cfg if case.get("existing") else None, rules=rules, **ops)


def resolve_for_oracle(mod, case, args):
    """Keyword arguments of ``dwt_oracle.wavelet_cfg_call`` for the rule that matches the case's sigma, or None when no rule
    matches / the rule blends to plain CFG (the oracle's caller then checks the fallback)."""
    rules = mod.WCFGRules.build(**json.loads(json.dumps(case["params"])))
    sigma_f = float(args["sigma"].max())
    rule = rules.get_rule(sigma_f)
    if rule is None:
        return None
    pcts = mod.WCFGPercentages.build(ms=args["model"].model_sampling, start_sigma=rule.start_sigma, end_sigma=rule.end_sigma, sigma=sigma_f,
                                     sigmas=args["model_options"].get("transformer_options", {}).get("sample_sigmas"))
    wcfg_blend = rule.blend_strength.get_value(pcts)
    if rule.blend_mode == "lerp" and wcfg_blend == 0:
        return None
    ws = rule.wavelet
    yh = [_Band(1 if ws.use_1d_dwt else 3)] * ws.level

    def pair(spec):
        if spec is None:
            return None
        sc = spec.get_scales(pcts, yh)
        return (sc.yl_scale, sc.yh_scales)

    return dict(target=rule.target_mode.name.lower(), high_precision=rule.high_precision_mode, use_1d=ws.use_1d_dwt, wcfg_blend=wcfg_blend,
                blend_mode=rule.blend_mode, wave=ws.wave, mode=ws.padding_mode, level=ws.level, inv_wave=ws.inv_wave, inv_mode=ws.inv_padding_mode,
                cond_scales=pair(rule.cond), uncond_scales=pair(rule.uncond), diff_scales=pair(rule.diff), final_scales=pair(rule.final),
                strength=rule.difference_blend_strength.get_value(pcts), blend=rule.difference_blend_mode)


def numpy_args(args):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in args.items() if k not in ("model", "model_options")}


def numpy_ops(ops):
    """The torch hooks of wavelet_cases.wcfg_ops as numpy callables for the oracle."""
    def wrap(fn):
        def call(t, **kw):
            ext = getattr(fn, "EXTENDED_LATENT_OPERATION", None)
            kw = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in kw.items()} if ext else {}
            return fn(latent=torch.from_numpy(np.ascontiguousarray(t)), **kw).numpy()
        return call
    return {k: wrap(v) for k, v in ops.items()}


def counted_launch(launch, calls):
    """Wrap what ``hip_lib.wcfg_lowpass_plan`` returned so that ``calls`` records every LAUNCH (a prepared launch that is dropped because
    another rule matched does not count)."""
    if launch is None:
        return None
    return lambda: calls.append(1) or launch()
