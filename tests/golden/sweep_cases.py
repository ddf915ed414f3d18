"""Case table of the wrapper sweep (shared by tests/golden/make_golden.py gen_wrapper_sweep and tests/test_gpu_round2.py): noise-item
wrappers of py/noise.py built generically on both sides (the reference's module in the build container, the product's on the GPU box)
from these plain specs, on shapes the per-row fixtures do not have."""

SHAPES = [(1, 3, 10, 14), (2, 4, 9, 7), (1, 4, 2, 12, 8), (3, 1, 18, 6), (1, 16, 8, 8), (2, 5, 12, 20)]
SIGMAS = [(9.0, 6.0), (6.0, 3.5), (12.0, 11.0)]

# name -> (class, factor, kwargs, chains {kwarg: [(noise_type, factor), ...]}, shape index, seed, calls)
WRAPPERS = {
    "random_mix1": ("RandomNoise", 0.7, dict(mix_count=1, normalize=None), dict(noise=[("gaussian", 1.0), ("uniform", 0.5), ("perlin", 0.8)]), 0, 141, 3),
    "random_mix2_video": ("RandomNoise", 1.0, dict(mix_count=2, normalize=True), dict(noise=[("gaussian", 1.0), ("pyramid", 0.5), ("laplacian", 0.8)]), 2, 142, 3),
    "repeated_odd": ("RepeatedNoise", 0.9, dict(repeat_length=2, max_recycle=3, normalize=None, permute="enabled"), dict(noise=[("gaussian", 1.0)]), 1, 143, 5),
    "repeated_always": ("RepeatedNoise", 1.0, dict(repeat_length=3, max_recycle=2, normalize=True, permute="always"), dict(noise=[("perlin", 1.0)]), 3, 144, 5),
    "channel_wrap_16": ("ChannelNoise", 1.1, dict(insufficient_channels_mode="wrap", normalize=None), dict(noise=[("gaussian", 1.0), ("uniform", 0.5), ("perlin", 1.0)]), 4, 145, 2),
    "channel_zero_5": ("ChannelNoise", 1.0, dict(insufficient_channels_mode="zero", normalize=True), dict(noise=[("gaussian", 1.0), ("pyramid", 0.5)]), 5, 146, 2),
    "channel_repeat_1": ("ChannelNoise", 1.0, dict(insufficient_channels_mode="repeat", normalize=None), dict(noise=[("uniform", 1.0), ("gaussian", 0.5)]), 3, 147, 2),
    "ripple_cos": ("RippleFilteredNoise", 0.8, dict(offset=0.3, roll=1.5, amplitude_high=0.25, amplitude_low=1.6, period=3.0, normalize_noise=False,
                                                  normalize=None, mode="cos", dim=-1, flatten=False), dict(noise=[("gaussian", 1.0)]), 0, 148, 2),
    "ripple_flat_video": ("RippleFilteredNoise", 1.0, dict(offset=0.0, roll=0.5, amplitude_high=0.5, amplitude_low=1.2, period=2.0, normalize_noise=True,
                                                         normalize=True, mode="sin_copysign", dim=2, flatten=True), dict(noise=[("perlin", 1.0)]), 2, 149, 2),
    "perdim_dim1": ("PerDimNoise", 0.6, dict(dim=1, shrink_dim=False, chunk_size=1, offset=0, normalize_noise=False, normalize=None), dict(noise=[("gaussian", 1.0)]), 5, 150, 2),
    "perdim_dim2_chunk": ("PerDimNoise", 1.0, dict(dim=2, shrink_dim=False, chunk_size=4, offset=1, normalize_noise=False, normalize=True), dict(noise=[("uniform", 1.0)]), 1, 151, 2),
    "perdim_shrink": ("PerDimNoise", 1.0, dict(dim=1, shrink_dim=True, chunk_size=2, offset=0, normalize_noise=True, normalize=None), dict(noise=[("gaussian", 1.0)]), 4, 152, 2),
    "to_scale_simple": ("NormalizeToScaleNoise", 1.0, dict(min_negative_value=-4.5, max_negative_value=0.0, min_positive_value=0.0, max_positive_value=4.5,
                                                          mode="simple", dims=(-3, -2, -1), std_dims=(-3, -2, -1), std_multiplier=1.0, mean_dims=(-3, -2, -1),
                                                          mean_multiplier=1.0, normalize_noise=False, normalize=None), dict(noise=[("gaussian", 1.0)]), 0, 157, 2),
    "to_scale_simple_hw": ("NormalizeToScaleNoise", 0.8, dict(min_negative_value=-1.5, max_negative_value=0.0, min_positive_value=0.0, max_positive_value=2.0,
                                                             mode="simple", dims=(-2, -1), std_dims=(-2, -1), std_multiplier=0.5, mean_dims=(-1,),
                                                             mean_multiplier=0.25, normalize_noise=True, normalize=True), dict(noise=[("perlin", 1.0)]), 5, 158, 2),
    "to_scale_simple_plain": ("NormalizeToScaleNoise", 1.0, dict(min_negative_value=0.25, max_negative_value=0.0, min_positive_value=0.0, max_positive_value=0.75,
                                                                mode="simple", dims=(), std_dims=(), std_multiplier=0.0, mean_dims=(), mean_multiplier=0.0,
                                                                normalize_noise=False, normalize=False), dict(noise=[("uniform", 1.0)]), 2, 159, 2),
    "to_scale_adv_global": ("NormalizeToScaleNoise", 1.0, dict(min_negative_value=-3.0, max_negative_value=-0.5, min_positive_value=0.25, max_positive_value=2.0,
                                                              mode="advanced", dims=(), std_dims=(), std_multiplier=0.0, mean_dims=(), mean_multiplier=0.0,
                                                              normalize_noise=False, normalize=False), dict(noise=[("gaussian", 1.0)]), 1, 160, 2),
    "to_scale_adv_auto": ("NormalizeToScaleNoise", 1.0, dict(min_negative_value=-4.3, max_negative_value=0.0, min_positive_value=-1.0, max_positive_value=3.7,
                                                            mode="advanced", dims=(-3, -2, -1), std_dims=(-3, -2, -1), std_multiplier=1.0, mean_dims=(-3, -2, -1),
                                                            mean_multiplier=1.0, normalize_noise=False, normalize=None), dict(noise=[("gaussian", 1.0)]), 5, 161, 2),
    "to_scale_adv_skip_neg": ("NormalizeToScaleNoise", 0.9, dict(min_negative_value=0.5, max_negative_value=1.0, min_positive_value=0.1, max_positive_value=1.2,
                                                                mode="advanced", dims=(-3, -2, -1), std_dims=(), std_multiplier=0.0, mean_dims=(), mean_multiplier=0.0,
                                                                normalize_noise=False, normalize=None), dict(noise=[("laplacian", 1.0)]), 0, 162, 2),
    "resized_up_crop": ("ResizedNoise", 1.0, dict(width=192.0, height=160.0, spatial_compression=8, spatial_mode="absolute", downscale_strategy="crop",
                                                  initial_reference="prefer_crop", crop_offset_horizontal=16, crop_offset_vertical=-8, crop_mode="top_left",
                                                  upscale_mode="bilinear", downscale_mode="area", normalize=None), dict(custom_noise=[("gaussian", 1.0)]), 0, 163, 2),
    "resized_up_scale": ("ResizedNoise", 0.8, dict(width=176.0, height=144.0, spatial_compression=8, spatial_mode="absolute", downscale_strategy="scale",
                                                   initial_reference="prefer_crop", crop_offset_horizontal=0, crop_offset_vertical=0, crop_mode="center",
                                                   upscale_mode="nearest-exact", downscale_mode="area", normalize="default"), dict(custom_noise=[("perlin", 1.0)]), 5, 164, 2),
    "resized_down_crop": ("ResizedNoise", 1.0, dict(width=64.0, height=48.0, spatial_compression=8, spatial_mode="absolute", downscale_strategy="crop",
                                                    initial_reference="prefer_crop", crop_offset_horizontal=8, crop_offset_vertical=8, crop_mode="bottom_right",
                                                    upscale_mode="bicubic", downscale_mode="bilinear", normalize=True), dict(custom_noise=[("gaussian", 1.0)]), 5, 165, 2),
    "resized_down_scale": ("ResizedNoise", 1.0, dict(width=64.0, height=48.0, spatial_compression=8, spatial_mode="absolute", downscale_strategy="scale",
                                                     initial_reference="prefer_scale", crop_offset_horizontal=0, crop_offset_vertical=0, crop_mode="center",
                                                     upscale_mode="bilinear", downscale_mode="bilinear", normalize=False), dict(custom_noise=[("uniform", 1.0)]), 5, 166, 2),
    "resized_mixed": ("ResizedNoise", 1.0, dict(width=-32.0, height=24.0, spatial_compression=8, spatial_mode="relative", downscale_strategy="crop",
                                                      initial_reference="prefer_crop", crop_offset_horizontal=0, crop_offset_vertical=0, crop_mode="center",
                                                      upscale_mode="nearest", downscale_mode="nearest-exact", normalize=None), dict(custom_noise=[("gaussian", 1.0)]), 3, 167, 2),
    "resized_percentage": ("ResizedNoise", 1.0, dict(width=1.5, height=2.25, spatial_compression=8, spatial_mode="percentage", downscale_strategy="crop",
                                                     initial_reference="prefer_crop", crop_offset_horizontal=24, crop_offset_vertical=0, crop_mode="center_left",
                                                     upscale_mode="area", downscale_mode="area", normalize=None), dict(custom_noise=[("pyramid", 1.0)]), 1, 168, 2),
    "resized_same": ("ResizedNoise", 0.6, dict(width=112.0, height=80.0, spatial_compression=8, spatial_mode="absolute", downscale_strategy="crop",
                                               initial_reference="prefer_crop", crop_offset_horizontal=0, crop_offset_vertical=0, crop_mode="center",
                                               upscale_mode="bilinear", downscale_mode="bilinear", normalize=None), dict(custom_noise=[("gaussian", 1.0)]), 0, 169, 2),
    "scheduled": ("ScheduledNoise", 1.0, dict(start_sigma=10.0, end_sigma=5.0, normalize=True),
                  dict(noise=[("perlin", 1.0), ("gaussian", 0.5)], fallback_noise=[("uniform", 1.0)]), 0, 153, 3),
    "scheduled_video": ("ScheduledNoise", 0.8, dict(start_sigma=20.0, end_sigma=4.0, normalize=None),
                        dict(noise=[("pyramid", 1.0)], fallback_noise=[("gaussian", 1.0)]), 2, 154, 3),
    "blended_lerp": ("BlendedNoise", 1.2, dict(normalize=True, blend_function="lerp", noise_2_percent=0.3),
                     dict(custom_noise_1=[("gaussian", 1.0)], custom_noise_2=[("uniform", 1.0)]), 1, 155, 2),
    "blended_mask": ("BlendedNoise", 1.0, dict(normalize=True, blend_function="inject", noise_2_percent=0.1),
                     dict(custom_noise_1=[("perlin", 1.0)], custom_noise_2=[("gaussian", 1.0)], custom_noise_mask=[("gaussian", 1.0)]), 5, 156, 2),
}


def build(noise_mod, utils_mod, name):
    """(item, shape, seed, calls) from the spec, with ``noise_mod`` / ``utils_mod`` the reference's or the product's modules."""
    cls, factor, kwargs, chains, shape_idx, seed, calls = WRAPPERS[name]
    kw = dict(kwargs)
    if "blend_function" in kw:
        kw["blend_function"] = utils_mod.BLENDING_MODES[kw["blend_function"]]
    for key, specs in chains.items():
        chain = noise_mod.CustomNoiseChain()
        for noise_type, f in specs:
            chain.add(noise_mod.CustomNoiseItem(f, noise_type=noise_type))
        kw[key] = chain
    return getattr(noise_mod, cls)(factor, **kw), SHAPES[shape_idx], seed, calls


# sampler sweep: name -> (sampler kind, config overrides, noise type, shape index, seed); 5 steps from sigma 14.6 down to 0, noise drawn by
# the registry sampler in replay mode (cpu=True) on the global generator, exactly as ComfyUI would run it
SAMPLERS = {
    "euler_odd": ("euler", dict(), "gaussian", 0, 171),
    "euler_video_denoised": ("euler", dict(momentum_mode="DENOISED", init="SAMPLE"), "gaussian", 2, 172),
    "ancestral_odd_perlin": ("ancestral", dict(momentum=0.8, momentum_hist=0.5), "perlin", 1, 173),
    "ancestral_video_pyramid": ("ancestral", dict(momentum_mode="CLASSIC"), "pyramid", 2, 174),
    "ancestral_16ch_uniform": ("ancestral", dict(direction=-0.5), "uniform", 4, 175),
    "dpmpp_odd_gaussian": ("dpmpp", dict(), "gaussian", 3, 176),
    "dpmpp_video_perlin": ("dpmpp", dict(momentum=0.7, init="SAMPLE_NORM"), "perlin", 2, 177),
    "dpmpp_5ch_laplacian": ("dpmpp", dict(blend_mode="inject", momentum=0.3, momentum_hist=0.4), "laplacian", 5, 178),
}


def fake_model(x, sigma, **_kw):
    import torch

    s = sigma.reshape(-1, *([1] * (x.ndim - 1)))
    return x * 0.5 + torch.tanh(x) * (0.1 * s / (1.0 + s))


def run_sampler(sonar_mod, noise_mod, name, device):
    """Per-step x of the named case (list of tensors) with ``sonar_mod`` / ``noise_mod`` the reference's or the product's modules."""
    import torch

    kind, cfg, noise_type, shape_idx, seed = SAMPLERS[name]
    shape = SHAPES[shape_idx]
    torch.manual_seed(seed)
    x0 = (torch.randn(shape) * 14.6).to(device)
    sigmas = torch.cat((torch.linspace(14.6, 0.03, 5), torch.zeros(1)))
    ns = noise_mod.get_noise_sampler(noise_type, x0, 0.03, 14.6, seed=seed, cpu=True, normalized=True)
    trace = []
    cb = lambda d: trace.append(d["x"].clone())  # noqa: E731
    extra = {"seed": seed}
    if kind == "euler":
        sonar_mod.SonarEuler.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, ns, None, dict(cfg))
    elif kind == "ancestral":
        sonar_mod.SonarEulerAncestral.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, None, dict(cfg), 0.8, 1.1, ns)
    else:
        sonar_mod.SonarDPMPPSDE.sampler(fake_model, x0.clone(), sigmas, extra, cb, True, None, dict(cfg), 0.9, 1.05, ns)
    return trace


# advanced items: name -> builder(noise_mod, powernoise_mod, utils_mod, torch, device) returning (item, x tensor on CPU); odd planes send the
# spectral ones through the direct DFT passes, the resamplers through non-integer ratios
def _chain(noise_mod, *specs):
    chain = noise_mod.CustomNoiseChain()
    for noise_type, f in specs:
        chain.add(noise_mod.CustomNoiseItem(f, noise_type=noise_type))
    return chain


def _latent(torch, shape, seed, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale + shift


WAVELET_KW = dict(octave_scale_mode="adaptive_avg_pool2d", octave_rescale_mode="bilinear", post_octave_rescale_mode="bilinear", initial_amplitude=1.0,
                  persistence=0.5, octaves=3, octave_height_factor=0.5, octave_width_factor=0.5, height_factor=2.0, width_factor=2.0, update_blend=1.0)

ADVANCED = {
    "modulated_intensity_odd": lambda N, P, U, t, dev: (N.ModulatedNoise(0.8, noise=_chain(N, ("gaussian", 1.0)), normalize_result=None, normalize_noise=None,
                                                                   normalize_ref=True, modulation_type="intensity", modulation_strength=1.5, modulation_dims=2,
                                                                   ref_latent_opt=_latent(t, (2, 4, 9, 7), 1, 1.3, 0.1)), _latent(t, (2, 4, 9, 7), 2, 2.0, 0.3)),
    "modulated_frequency_odd": lambda N, P, U, t, dev: (N.ModulatedNoise(1.0, noise=_chain(N, ("gaussian", 1.0)), normalize_result=None, normalize_noise=None,
                                                                   normalize_ref=True, modulation_type="frequency", modulation_strength=1.2, modulation_dims=3,
                                                                   ref_latent_opt=_latent(t, (1, 3, 15, 21), 3, 1.0, 0.0)), _latent(t, (1, 3, 15, 21), 4, 2.0, 0.0)),
    "modulated_frequency_self": lambda N, P, U, t, dev: (N.ModulatedNoise(0.9, noise=_chain(N, ("perlin", 1.0)), normalize_result=False, normalize_noise=None,
                                                                    normalize_ref=True, modulation_type="frequency", modulation_strength=-0.6, modulation_dims=1,
                                                                    ref_latent_opt=None), _latent(t, (2, 5, 12, 20), 5, 2.0, 0.3)),
    "guided_linear_odd": lambda N, P, U, t, dev: (N.GuidedNoise(1.0, guidance_factor=0.4, ref_latent=U.scale_noise(_latent(t, (1, 4, 11, 9), 6, 0.8, 0.2).to(dev), normalized=True),
                                                           method="linear", normalize_noise=None, normalize_result=None, noise=_chain(N, ("gaussian", 1.0))),
                                             _latent(t, (2, 4, 13, 17), 7, 3.0)),
    "guided_euler_odd": lambda N, P, U, t, dev: (N.GuidedNoise(0.7, guidance_factor=-0.3, ref_latent=U.scale_noise(_latent(t, (2, 3, 20, 12), 8, 0.8, 0.2).to(dev), normalized=True),
                                                          method="euler", normalize_noise=None, normalize_result=None, noise=None), _latent(t, (2, 3, 10, 14), 9, 3.0)),
    "wavelet_noise_odd": lambda N, P, U, t, dev: (N.AdvancedWaveletNoise(1.0, custom_noise=None, normalize_noise=False, normalize=None, update_blend_function=t.lerp,
                                                                   **WAVELET_KW), t.zeros(1, 3, 30, 22)),
    "wavelet_noise_custom": lambda N, P, U, t, dev: (N.AdvancedWaveletNoise(1.0, custom_noise=_chain(N, ("uniform", 1.0)), normalize_noise=True, normalize=None,
                                                                      update_blend_function=t.lerp, **WAVELET_KW), t.zeros(2, 4, 18, 26)),
    "power_filter_noise_odd": lambda N, P, U, t, dev: (P.PowerFilterNoiseItem(1.0, noise=_chain(N, ("gaussian", 1.0)), normalize_noise=None, normalize_result=None,
                                                                        time_brownian=True, power_filter=P.PowerFilter(alpha=1.0, max_freq=0.5),
                                                                        filter_norm_factor=1.0, mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1"),
                                                  t.zeros(2, 4, 15, 21)),
    "power_filter_noise_mixed": lambda N, P, U, t, dev: (P.PowerFilterNoiseItem(0.8, noise=_chain(N, ("uniform", 1.0), ("perlin", 0.5)), normalize_noise=None,
                                                                          normalize_result=None, time_brownian=True,
                                                                          power_filter=P.PowerFilter(alpha=-0.5, min_freq=0.1, max_freq=0.7071, rotate=20.0, stretch=1.5),
                                                                          filter_norm_factor=1.0, mix=0.7, common_mode=0.25, channel_correlation="1,0.5,0.2,1,0.3,0.1"),
                                                    t.zeros(1, 4, 24, 40)),
    "power_noise_odd": lambda N, P, U, t, dev: (P.PowerNoiseItem(1.0, time_brownian=False, alpha=1.5, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                                                            mix=1.0, common_mode=0.0, channel_correlation="1,1,1,1,1,1"), t.zeros(2, 4, 17, 33)),
}
ADVANCED_SEEDS = {name: 200 + k for k, name in enumerate(sorted(ADVANCED))}


def run_advanced(noise_mod, powernoise_mod, utils_mod, name, device):
    import torch

    item, x = ADVANCED[name](noise_mod, powernoise_mod, utils_mod, torch, device)
    seed = ADVANCED_SEEDS[name]
    for attr in ("ref_latent", "ref_latent_opt"):  # reference latents travel with the item
        v = getattr(item, attr, None)
        if torch.is_tensor(v):
            setattr(item, attr, v.to(device))
    torch.manual_seed(seed)
    ns = item.make_noise_sampler(x.to(device), 0.03, 14.6, seed=seed, cpu=True, normalized=True)
    return [ns(torch.tensor(s), torch.tensor(sn)) for s, sn in SIGMAS[:2]]


# node sweep: every node that returns a SONAR_CUSTOM_NOISE chain, built with the defaults of its sockets (tests/golden/node_abi.json, captured from
# the reference) over a gaussian base chain, then sampled twice in replay mode on an odd latent
NODE_SHAPE = (2, 4, 10, 14)
NODE_SKIP = {"SonarBlehOpsNoise", "SonarBlendFilterNoise"}  # integrations with other custom-node packs (not in the image)


def node_defaults(abi_entry, provide):
    kw = {}
    for name, spec in abi_entry["inputs"].items():
        t = spec["type"]
        if isinstance(t, list):
            kw[name] = spec.get("default", t[0])
        elif t in ("FLOAT", "INT", "BOOLEAN", "STRING"):
            if "default" not in spec and t != "STRING":
                return None
            kw[name] = spec.get("default", "")  # (yaml_parameters sockets have no default: empty text)
        elif spec["section"] == "optional":
            continue
        elif t in provide:
            kw[name] = provide[t]()
        elif t == "*" and "custom_noise" in name:
            kw[name] = provide["SONAR_CUSTOM_NOISE"]()
        else:
            return None
    return kw


def run_node(mappings, abi, key, device):
    """Outputs of two sampler calls through the chain node ``key`` builds from its default sockets; None when a required socket has no default
    and no provider here."""
    import torch

    def base_chain():
        node = mappings["SonarCustomNoise"]()
        return getattr(node, abi["SonarCustomNoise"]["function"])(factor=1.0, rescale=0.0, noise_type="gaussian")[0]

    def power_filter():
        node = mappings["SonarPowerFilter"]()
        kw = node_defaults(abi["SonarPowerFilter"], {})
        return getattr(node, abi["SonarPowerFilter"]["function"])(**kw)[0]

    def latent():
        g = torch.Generator().manual_seed(31)
        return {"samples": (torch.randn(NODE_SHAPE, generator=g) * 0.8 + 0.2).to(device)}

    def mask():
        g = torch.Generator().manual_seed(32)
        return (torch.rand(1, NODE_SHAPE[-2], NODE_SHAPE[-1], generator=g) > 0.5).float()

    provide = {"SONAR_CUSTOM_NOISE": base_chain, "SONAR_POWER_FILTER": power_filter, "LATENT": latent, "MASK": mask}
    kw = node_defaults(abi[key], provide)
    if kw is None:
        return None
    node = mappings[key]()
    chain = getattr(node, abi[key]["function"])(**kw)[0]
    x = torch.zeros(NODE_SHAPE, device=device)
    seed = 900 + sum(key.encode()) % 97
    torch.manual_seed(seed)
    ns = chain.make_noise_sampler(x, 0.03, 14.6, seed=seed, cpu=True, normalized=True)
    return [ns(torch.tensor(s), torch.tensor(sn)) for s, sn in SIGMAS[:2]]
