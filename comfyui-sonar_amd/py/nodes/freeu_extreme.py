"""FreeU-Extreme's spectral filter on MI355X (``ffilter`` of the reference's ``py/nodes/freeu_extreme.py:10-29``).

Only the filter is on the hot path this package rebuilds: ``irfft2(rfft2(x) * filter)`` on an arbitrary
feature map is ONE call of the LDS-resident forward + filter + inverse kernel (``sonar_spectral_filter_f32``: one read and one write
of the tensor).  The UNet block patching around it (FreeUExtreme / FreeUExtremeConfig nodes) is ComfyUI model plumbing and stays
outside (their registry keys exist and raise).
"""
from __future__ import annotations

import torch

from ... import hip_lib
from ..utils import host_setup_threads
from .powernoise import PowerFilter


def ffilter(x: torch.Tensor, pfilter: PowerFilter, normalization_factor: float = 1.0, cfg_idx=None, filter_cache=None) -> torch.Tensor:
    """py/nodes/freeu_extreme.py:10-29.  The normalised half-spectrum filter is built once per (cfg_idx, plane size) and kept on the
    device in ``filter_cache``; a hit is used whatever ``pfilter`` says, like the reference.  fp16 / bf16 inputs are filtered in fp32
    and cast back."""
    cache_key = None
    if filter_cache is not None and cfg_idx is not None:
        cache_key = (cfg_idx, x.shape[-2:])
        filter_rfft = filter_cache.get(cache_key)
    else:
        # the reference reads `filter_rfft` before binding it when there is no cache key (:12-16): same failure, not a silent rebuild
        raise UnboundLocalError("local variable 'filter_rfft' referenced before assignment")
    if not x.is_cuda:
        raise hip_lib.SonarHipError(f"ffilter: got a {x.device} tensor; this implementation only runs on a ROCm device")
    if filter_rfft is None:
        with host_setup_threads():
            filter_rfft = PowerFilter.normalize(pfilter.build(x.shape), x.shape, normalization_factor=normalization_factor)
        filter_rfft = filter_rfft.to(x.device, torch.float32)
    filter_cache[cache_key] = filter_rfft
    h, w = x.shape[-2:]
    if not hip_lib.power_supported(h, w):
        raise hip_lib.SonarHipError(f"ffilter: plane {h}x{w} is beyond the spectral kernels (sides of at most 2048)")
    x32 = x if x.dtype == torch.float32 else x.to(torch.float32)
    out = hip_lib.spectral_filter(x32.contiguous(), filter_rfft.reshape(h, w // 2 + 1).contiguous())
    return out if x.dtype == torch.float32 else out.to(x.dtype)
