"""CPU: the ComfyUI node ABI — all 54 keys, socket names, order, types, enum members and
defaults equal to the reference's (tests/golden/node_abi.json, captured from the reference's INPUT_TYPES())."""
import importlib
import json
import os

import pytest
import torch

from tests.conftest import GOLDEN

ABI = json.load(open(os.path.join(GOLDEN, "node_abi.json")))


def test_all_54_keys_in_reference_order(pkg):
    assert list(pkg.NODE_CLASS_MAPPINGS) == list(ABI) and len(ABI) == 54
    assert {"SonarCustomNoise", "SonarPowerNoise", "NoisyLatentLike", "SamplerSonarEuler", "SamplerSonarEulerA", "SamplerSonarDPMPPSDE",
            "SonarWaveletCFG", "SONAR_CUSTOM_NOISE to NOISE"} <= pkg.nodes.IMPLEMENTED_KEYS


@pytest.mark.parametrize("key", list(ABI))
def test_node_sockets_match_reference(pkg, key):
    cls = pkg.NODE_CLASS_MAPPINGS[key]
    want = ABI[key]
    assert tuple(cls.RETURN_TYPES) == tuple(want["returns"]) and cls.FUNCTION == want["function"] and cls.CATEGORY == want["category"]
    assert callable(getattr(cls, cls.FUNCTION))
    got = cls.INPUT_TYPES()
    flat = [(sec, name) for sec in ("required", "optional") for name in got[sec]]
    assert [n for _s, n in flat] == [n for n, v in want["inputs"].items() if v["section"] == "required"] + \
        [n for n, v in want["inputs"].items() if v["section"] == "optional"]
    for sec, name in flat:
        spec, ref = got[sec][name], want["inputs"][name]
        assert sec == ref["section"]
        typ = list(spec[0]) if isinstance(spec[0], tuple) else spec[0]
        assert typ == ref["type"]
        opts = spec[1] if len(spec) > 1 else {}
        for k in ("default", "min", "max"):
            if k in ref and not (isinstance(ref[k], str) and len(ref[k]) > 200):  # long help-text defaults are replaced by a short placeholder
                assert opts[k] == ref[k], (name, k)


def test_off_path_nodes_fail_loudly(pkg):
    cls = pkg.NODE_CLASS_MAPPINGS["SonarAdvancedCollatzNoise"]
    with pytest.raises(NotImplementedError):
        getattr(cls(), cls.FUNCTION)()


def test_chain_building_nodes(pkg):
    M = pkg.NODE_CLASS_MAPPINGS
    nz = importlib.import_module("comfyui_sonar_amd.py.noise")
    (chain,) = M["SonarCustomNoise"]().go(factor=0.6, rescale=0.0, noise_type="gaussian")
    (chain,) = M["SonarCustomNoise"]().go(factor=-0.3, rescale=0.0, noise_type="perlin", sonar_custom_noise_opt=chain)
    assert isinstance(chain, nz.CustomNoiseChain) and [i.factor for i in chain.items] == [0.6, -0.3]
    (scaled,) = M["SonarCustomNoise"]().go(factor=0.1, rescale=2.0, noise_type="uniform", sonar_custom_noise_opt=chain)
    assert scaled.factor == pytest.approx(2.0) and len(chain.items) == 2  # the incoming chain is cloned, not mutated
    (same,) = M["SonarCustomNoise"]().go(factor=0.0, rescale=0.0, noise_type="uniform", sonar_custom_noise_opt=chain)
    assert len(same.items) == 2  # factor 0 adds nothing
    (pw,) = M["SonarPowerNoise"]().go(factor=1.0, rescale=0.0, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0,
                                     rotate=0.0, pnorm=2.0, mix=1.0, common_mode=0.0, channel_correlation="1, 1, 1, 1, 1, 1", preview="none")
    assert pw.items[0].power_filter.alpha == 1.0 and torch.equal(pw.items[0].channel_correlation, torch.ones(6))
    (filt,) = M["SonarPowerFilter"].go(alpha=0.5, max_freq=0.5, min_freq=0.1, stretch=1.0, rotate=0.0, pnorm=2.0, oversample=4, blur=0.2,
                                       scale=1.0, compose_mode="max")
    assert filt.rel_bw == 0.2 and filt.alpha == 0.5
    # the composite node swaps normalize_dst / normalize_src exactly like the reference (noise_filters.py:246-247)
    (comp,) = M["SonarCompositeNoise"]().go(factor=1.0, sonar_custom_noise_dst=chain, sonar_custom_noise_src=chain, normalize_src="forced",
                                            normalize_dst="disabled", normalize_result="default", mask=torch.ones(1, 8, 8))
    item = comp.items[0]
    assert item.normalize_dst is True and item.normalize_src is False and item.normalize_result is None


def test_sampler_nodes_build_ksamplers(pkg):
    M = pkg.NODE_CLASS_MAPPINGS
    S = importlib.import_module("comfyui_sonar_amd.py.sonar")
    (k,) = M["SamplerSonarEuler"].get_sampler(momentum=0.9, momentum_hist=0.7, momentum_init="SAMPLE", direction=-0.5, rand_init_noise_type="perlin")
    cfg = k.extra_options["sonar_config"]
    assert k.sampler_function == S.SonarEuler.sampler and cfg.init == S.HistoryType.SAMPLE and cfg.direction == -0.5
    (k,) = M["SamplerSonarDPMPPSDE"].get_sampler(momentum=0.95, momentum_hist=0.75, momentum_init="ZERO", direction=1.0, rand_init_noise_type="gaussian",
                                                 noise_type="brownian", eta=0.8, s_noise=1.1)
    assert k.sampler_function == S.SonarDPMPPSDE.sampler and k.extra_options["eta"] == 0.8 and k.extra_options["s_noise"] == 1.1
    (g,) = M["SonarGuidanceConfig"].make_guidance_cfg("euler", 0.02, 1, 5, {"samples": torch.zeros(1, 4, 8, 8)})
    assert g.guidance_type == S.GuidanceType.EULER and g.end_step == 5
