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
