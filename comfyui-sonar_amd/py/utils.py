"""Tensor utilities of the Sonar hot path on MI355X (mirrors the reference's ``py/utils.py`` API).

Every arithmetic function launches hand-written HIP kernels (``hip_lib``) on ROCm tensors.  The
normaliser keeps the reference's whole-tensor, data-dependent semantics (py/utils.py:85-106) but
evaluates the thresholds on device, so there is no ``.item()`` host sync.
"""
from __future__ import annotations

import contextlib
import threading
from typing import Callable, Optional, Sequence

import torch

from .. import hip_lib

Tensor = torch.Tensor

UPSCALE_METHODS = ("bilinear", "nearest-exact", "nearest", "area", "bicubic", "bislerp", "adaptive_avg_pool2d")

_STATS_ATTR = hip_lib.STATS_ATTR


_HOST_SETUP_THREADS = 4
_host_setup_lock = threading.Lock()
_host_setup_users = 0
_host_setup_before = None


@contextlib.contextmanager
def host_setup_threads():
    """The host-side, once-per-sampler tensor work of the path (the power filter's oversampled grid, py/nodes/powernoise.py:156-266) on a
    few CPU threads instead of torch's default of one per logical CPU.  Round 5 traced an intermittent 50-90 ms freeze of launch-bound
    steps (round 4's bench read 434 us for a 27 us call) to this: a 256-thread intra-op pool spins for a while after every parallel
    region, a container with a CPU quota (the GPU pool's: 16 CPUs per 100 ms) runs out of quota, and the kernel's bandwidth control
    throttles the WHOLE process -- the thread that launches kernels included -- until the period ends (cpu.stat nr_throttled 0 -> 100
    over three 4-second runs; none with 4 threads; scratch/stall_fresh.py).  The tensors are a few hundred kilobytes:
    four threads lose nothing, and elementwise results do not depend on the thread count.
    The count is process-global, so concurrent users share ONE lowering: the first to enter lowers it and remembers the old value, the
    last to leave puts it back (a lock and a count of users; interleaved enter / exit pairs used to be able to leave the pool at four
    threads, or restore a stale value)."""
    global _host_setup_users, _host_setup_before
    with _host_setup_lock:
        if _host_setup_users == 0:
            before = torch.get_num_threads()
            _host_setup_before = before if before > _HOST_SETUP_THREADS else None
            if _host_setup_before is not None:
                torch.set_num_threads(_HOST_SETUP_THREADS)
        _host_setup_users += 1
    try:
        yield
    finally:
        with _host_setup_lock:
            _host_setup_users -= 1
            if _host_setup_users == 0 and _host_setup_before is not None:
                torch.set_num_threads(_host_setup_before)
                _host_setup_before = None


def fallback(val, default=None):
    return val if val is not None else default


# --------------------------------------------------------------------------------------------------
# fused statistics hand-off: a producer kernel that already reduced (sum, sumsq) of the tensor it
# wrote tags the tensor; the next scale_noise() on that very tensor consumes the tag instead of
# re-reading the tensor.  Any in-package op that changes the values must drop the tag first.
def attach_stats(t: Tensor, partials: Optional[Tensor]) -> Tensor:
    if partials is not None:
        setattr(t, _STATS_ATTR, (partials, t._version))
        hip_lib.tag_register(t)
    return t


def pop_stats(t: Tensor) -> Optional[Tensor]:
    """The tag is honoured only if nothing wrote to the tensor since: torch's in-place ops bump ``_version`` (shared by all views of
    a storage); every kernel of this library that is handed ANY view of the tensor's storage drops the tag (hip_lib._dev)."""
    p = getattr(t, _STATS_ATTR, None)
    if p is None:
        return None
    delattr(t, _STATS_ATTR)
    hip_lib.tag_forget(t)
    partials, version = p
    return partials if version == t._version else None


def _require_device(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise hip_lib.SonarHipError(f"{what}: got a {t.device} tensor; this implementation only runs on a ROCm device")


def as_f32(t: Tensor) -> Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


# --------------------------------------------------------------------------------------------------
def _tiles(t: Tensor, a: Tensor) -> bool:
    """True when ``t`` broadcast against ``a`` is a plain tiling of t's flat data (trailing dims match)."""
    ts = list(t.shape)
    while ts and ts[0] == 1:
        ts.pop(0)
    return ts == list(a.shape[a.ndim - len(ts):])


def _blend(mode: str) -> Callable:
    def fn(a: Tensor, b: Tensor, t, out: Optional[Tensor] = None) -> Tensor:
        _require_device(a, f"blend[{mode}]")
        a32, b32 = as_f32(a), as_f32(b.to(a.device))
        if b32.shape != a32.shape:
            b32 = b32.expand_as(a32).contiguous()
        if isinstance(t, Tensor):
            if t.numel() == 1:
                t = float(t)  # host scalar weight (syncs only if the weight lives on the device)
            else:
                t = as_f32(t.to(a.device))
                if not _tiles(t, a32):
                    t = t.expand_as(a32).contiguous()
        res = hip_lib.blend(mode, a32, b32, t, out)
        return res if a.dtype == torch.float32 else res.to(a.dtype)

    fn.__name__ = f"blend_{mode}"
    return fn


# py/utils.py:17-21
BLENDING_MODES = {"lerp": _blend("lerp"), "inject": _blend("inject"), "subtract_b": _blend("subtract_b")}


# --------------------------------------------------------------------------------------------------
def scale_noise(noise: Tensor, factor: float = 1.0, *, normalized: bool = True, threshold_std_devs: float = 2.5,
                normalize_dims: Optional[Sequence[int]] = None) -> Tensor:
    """py/utils.py:85-106, in place.  Two kernels (stats -> apply) unless a producer kernel already
    attached the statistics, in which case only the apply kernel runs."""
    n = noise.numel()
    if not normalized or n == 0:
        if factor == 1:
            return noise  # untouched: a statistics tag (written by the producing kernel, checked against storage + version) still describes it
        pop_stats(noise)  # the values change below: the tag would describe other contents
        _require_device(noise, "scale_noise")
        return hip_lib.scale_noise_(noise, factor, False, None)
    _require_device(noise, "scale_noise")
    if noise.dtype != torch.float32 or not noise.is_contiguous():
        raise hip_lib.SonarHipError("scale_noise: expects a contiguous float32 tensor")
    if normalize_dims is not None:
        pop_stats(noise)
        rows_last, inverse = dims_last(noise, normalize_dims)
        inner = 1
        for d in range(noise.ndim - len({d % noise.ndim for d in normalize_dims}), noise.ndim):
            inner *= rows_last.shape[d]
        return dims_restore(hip_lib.scale_noise_rows_(rows_last, n // inner, inner, factor), inverse)
    partials = pop_stats(noise)
    if partials is None:
        partials = hip_lib.stats(noise)
    noise, after = hip_lib.scale_noise_stats_(noise, factor, partials, threshold_std_devs=threshold_std_devs)
    return attach_stats(noise, after)  # a wrapper that normalises this result again skips its statistics sweep


def dims_last(t: Tensor, dims):
    """(tensor with the dimensions ``dims`` moved to the end and made contiguous, permutation that undoes the move or None).  The row
    kernels reduce over trailing dimensions; any other choice (the reference takes any ``dim`` tuple: py/utils.py:97-99,452-470,
    py/sonar.py:372-377) is the same reduction on a transposed copy -- a layout change, no arithmetic."""
    nd = t.ndim
    dims = sorted({d % nd for d in dims})
    if dims == list(range(nd - len(dims), nd)):
        return t.contiguous(), None
    perm = [d for d in range(nd) if d not in dims] + dims
    return t.permute(perm).contiguous(), [perm.index(d) for d in range(nd)]


def dims_restore(t: Tensor, inverse) -> Tensor:
    return t if inverse is None else t.permute(inverse).contiguous()


def scale_samples(samples: Tensor, width: int, height: int, *, mode: str = "bicubic") -> Tensor:
    """py/utils.py:58-67: ``F.interpolate(samples, size=(height, width), mode=mode)`` on the HIP resampler (bilinear / nearest /
    nearest-exact / bicubic / area / adaptive_avg_pool2d)."""
    _require_device(samples, "scale_samples")
    if mode in hip_lib.UPSCALE_MODES:
        src = as_f32(samples)
        out = torch.empty((*src.shape[:-2], height, width), dtype=torch.float32, device=src.device)
        hip_lib.resample_acc_(out, src, 1.0, mode, accumulate=False)
        return out if samples.dtype == torch.float32 else out.to(samples.dtype)
    raise NotImplementedError(f"upscale mode {mode!r} needs ComfyUI's bislerp, which the reference does not vendor")


def normalize_to_scale(latent: Tensor, target_min: float, target_max: float, *, dim=(-3, -2, -1), eps: float = 1e-07) -> Tensor:
    """py/utils.py:452-470: per-group min / max (HIP reduction), then rescale + clamp in one kernel."""
    _require_device(latent, "normalize_to_scale")
    x = as_f32(latent)
    dims = sorted({d % x.ndim for d in dim}) if len(dim) else list(range(x.ndim))
    x, inverse = dims_last(x, dims)
    inner = 1
    for d in range(x.ndim - len(dims), x.ndim):
        inner *= x.shape[d]
    rows = x.numel() // inner
    lo, hi = hip_lib.minmax_rows(x, rows, inner)
    return dims_restore(hip_lib.minmax_rescale(x, rows, inner, lo, hi, eps, target_min, target_max), inverse)


def normalize_to_scale_adv(t: Tensor, *, min_pos: float, max_pos: float, min_neg: float, max_neg: float, dim=(-3, -2, -1)) -> Tensor:
    """py/utils.py:473-510: negatives and positives rescaled separately between their own extremes.  The reference selects the values of a sign
    into a 1-D tensor first, so whatever ``dim`` says the extremes are those of the WHOLE tensor passed in: one row."""
    _require_device(t, "normalize_to_scale_adv")
    x = as_f32(t).contiguous()
    return hip_lib.signed_rescale(x, 1, x.numel(), min_neg, max_neg, min_pos, max_pos)


def tensor_to(tensor: Tensor, dest) -> Tensor:
    """py/utils.py:112-121."""
    device = dest.device if isinstance(dest, Tensor) else dest
    return tensor.to(device, non_blocking=True)


def crop_samples(tensor: Tensor, width: int, height: int, *, mode: str = "center", offset_width: int = 0, offset_height: int = 0) -> Tensor:
    """py/utils.py:513-570 (pure indexing): a (height, width) window of the last two dimensions, anchored per axis at an edge or the
    centre (`center`, or `<top|center|bottom>_<left|center|right>`) and then moved by the offsets as far as the tensor allows."""
    if tensor.ndim < 3:
        raise ValueError("Can only handle >= 3 dimensional tensors")
    th, tw = tensor.shape[-2:]
    if (tw, th) == (width, height):
        return tensor
    if tw < width or th < height:
        raise ValueError("Can't crop sample smaller than requested width or height")
    parts = ("center", "center") if mode == "center" else tuple(mode.split("_"))
    if len(parts) != 2:
        raise ValueError("Bad composite mode")
    starts = []
    for which, size, want, names in ((parts[0], th, height, ("top", "center", "bottom")), (parts[1], tw, width, ("left", "center", "right"))):
        if which not in names:
            raise ValueError("Bad height mode in composite mode" if names[0] == "top" else "Bad width mode in composite mode")
        starts.append({names[0]: 0, names[1]: (size - want) // 2, names[2]: size - want}[which])

    def shifted(start, want, size, off):
        if off < 0:
            return start - min(start, -off)
        return start + min(size - (start + want), off)

    y0 = shifted(starts[0], height, th, offset_height)
    x0 = shifted(starts[1], width, tw, offset_width)
    return tensor[..., y0:y0 + height, x0:x0 + width]


def tensor_item(val, *, collapse_function=torch.max) -> float:
    if isinstance(val, Tensor):
        return float(collapse_function(val).detach().cpu().item())
    return float(val)
