"""Container-only loader for the upstream reference (TEST INFRASTRUCTURE, never shipped).

Imports /root/reference/py under the alias ``sonar_ref`` behind a throw-away stand-in
for the ComfyUI host modules it expects.  Used ONLY by
``tests/golden/make_golden.py`` to capture golden vectors in the build container;
``/root/reference`` does not exist on the GPU box and nothing at run time imports this.

The stand-ins restate the upstream ComfyUI helpers the hot path touches (from the
published k-diffusion formulas; ComfyUI itself is not vendored by the reference):
  to_d(x, sigma, den)            = (x - den) / sigma
  get_ancestral_step(s, s', eta) = (sigma_down, sigma_up)
  common_upscale(t, w, h, mode)  = F.interpolate(t, size=(h, w), mode=mode)
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SONAR_REFERENCE_ROOT", "/root/reference")
ALIAS = "sonar_ref"


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_host_standins() -> None:
    import torch
    import torch.nn.functional as F

    if "comfy" in sys.modules and getattr(sys.modules["comfy"], "_sonar_standin", False):
        return

    def common_upscale(samples, width, height, upscale_method, crop):
        if upscale_method in {"bislerp", "lanczos"}:
            raise NotImplementedError(f"{upscale_method} is not available in the stand-in")
        return F.interpolate(samples, size=(height, width), mode=upscale_method)

    def repeat_to_batch_size(tensor, batch_size, dim=0):
        n = tensor.shape[dim]
        if n > batch_size:
            return tensor.narrow(dim, 0, batch_size)
        if n < batch_size:
            reps = [1] * tensor.ndim
            reps[dim] = -(-batch_size // n)
            return tensor.repeat(*reps).narrow(dim, 0, batch_size)
        return tensor

    def to_d(x, sigma, denoised):
        while sigma.ndim < x.ndim:
            sigma = sigma[..., None]
        return (x - denoised) / sigma

    def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
        if not eta:
            return sigma_to, 0.0
        sigma_up = min(
            sigma_to,
            eta * (sigma_to**2 * (sigma_from**2 - sigma_to**2) / sigma_from**2) ** 0.5,
        )
        sigma_down = (sigma_to**2 - sigma_up**2) ** 0.5
        return sigma_down, sigma_up

    class BrownianTreeNoiseSampler:
        def __init__(self, *a, **k):
            raise NotImplementedError("torchsde is not installed in this container")

    class KSampler:
        SAMPLERS: list = []

    class KSAMPLER:
        def __init__(self, sampler_function, extra_options=None, inpaint_options=None):
            self.sampler_function = sampler_function
            self.extra_options = extra_options or {}
            self.inpaint_options = inpaint_options or {}

    class SD15:
        latent_channels = 4

    sampling = _mod(
        "comfy.k_diffusion.sampling",
        to_d=to_d,
        get_ancestral_step=get_ancestral_step,
        BrownianTreeNoiseSampler=BrownianTreeNoiseSampler,
    )
    kdiff = _mod("comfy.k_diffusion", sampling=sampling)
    samplers = _mod(
        "comfy.samplers", KSampler=KSampler, KSAMPLER=KSAMPLER, k_diffusion_sampling=sampling
    )
    mm = _mod(
        "comfy.model_management",
        device_supports_non_blocking=lambda _d: False,
        get_torch_device=lambda: torch.device("cpu"),
        throw_exception_if_processing_interrupted=lambda: None,
    )
    cutils = _mod(
        "comfy.utils", common_upscale=common_upscale, repeat_to_batch_size=repeat_to_batch_size
    )
    lf = _mod("comfy.latent_formats", SD15=SD15)
    _mod(
        "comfy",
        _sonar_standin=True,
        k_diffusion=kdiff,
        samplers=samplers,
        model_management=mm,
        utils=cutils,
        latent_formats=lf,
    )
    _mod(
        "folder_paths",
        get_temp_directory=lambda: "/tmp",
        get_save_image_path=lambda p, d: (d, p, 0, "", p),
    )
    _mod("latent_preview", get_previewer=lambda *a, **k: None)


def load_reference() -> types.SimpleNamespace:
    """Import the reference's ``py`` package (NOT its root __init__, which patches ComfyUI)."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT} (container-only tool)")
    _install_host_standins()
    if ALIAS not in sys.modules:
        pkg_dir = os.path.join(REFERENCE_ROOT, "py")
        spec = importlib.util.spec_from_file_location(
            ALIAS, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir]
        )
        pkg = importlib.util.module_from_spec(spec)
        sys.modules[ALIAS] = pkg
        sys.dont_write_bytecode = True
        spec.loader.exec_module(pkg)
    names = {}
    # `noise` must be imported before `sonar` (circular import in the reference).
    for sub in ("noise", "sonar", "utils", "noise_generation", "wavelet_functions", "latent_ops"):
        names[sub] = importlib.import_module(f"{ALIAS}.{sub}")
    names["powernoise"] = importlib.import_module(f"{ALIAS}.nodes.powernoise")
    names["nodes"] = importlib.import_module(f"{ALIAS}.nodes")
    return types.SimpleNamespace(**names)


if __name__ == "__main__":
    ref = load_reference()
    print("nodes:", len(ref.nodes.NODE_CLASS_MAPPINGS))
    print("noise types:", len(ref.noise_generation.NoiseType))
