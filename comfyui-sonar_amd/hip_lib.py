"""ctypes binding of libsonar_hip.so (the C ABI declared in include/sonar_hip.h).

PyTorch is used for device memory and streams only; every function here hands raw device
pointers (``tensor.data_ptr()``) and the current HIP stream to a hand-written kernel.
There is NO fallback: a missing library, a CPU tensor or a non-zero return code raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading as _threading
import weakref
from typing import Optional, Sequence

import torch  # noqa: F401  (must be imported before the .so so both share one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsonar_hip.so")

BLEND_IDS = {"lerp": 0, "inject": 1, "subtract_b": 2}
MODE_IDS = {"CLASSIC": 0, "NEW": 1, "DENOISED": 2}
INIT_IDS = {"NONE": 0, "SAMPLE": 1, "SAMPLE_NORM": 2}
RESAMPLE_IDS = {"bilinear": 0, "nearest-exact": 1, "area": 2, "adaptive_avg_pool2d": 2, "nearest": 3, "bicubic": 4, "bicubic_aligned": 5,
                "bilinear_aligned": 6}
UPSCALE_MODES = ("bilinear", "nearest-exact", "area", "adaptive_avg_pool2d", "nearest", "bicubic")  # F.interpolate modes of `scale_samples`
PYRAMID_FUSED_MODES = ("bilinear", "nearest-exact", "area", "adaptive_avg_pool2d")                  # modes the fused pyramid kernels carry
DWT_MODE_IDS = {"zero": 0, "symmetric": 1, "reflect": 2, "periodization": 3, "periodic": 4, "constant": 5, "replicate": 5}
NPART = 1024
BROWNIAN_MAX_TERMS = 96  # kMaxBrownianNodes (csrc/noise_gen.hip)
ERR_ARG, ERR_UNSUPPORTED, ERR_HIP = -1, -2, -3  # include/sonar_hip.h


class SonarHipError(RuntimeError):
    pass


class MomentumCfg(C.Structure):
    """Mirror of ``sonar_momentum_cfg`` (include/sonar_hip.h)."""

    _fields_ = [
        ("momentum", C.c_float),
        ("hist_ratio", C.c_float),
        ("hist_scale", C.c_float),
        ("md_scale", C.c_float),
        ("mode", C.c_int32),
        ("momentum_blend", C.c_int32),
        ("history_blend", C.c_int32),
        ("use_momentum", C.c_int32),
        ("update_hist", C.c_int32),
        ("init_kind", C.c_int32),
        ("h_in_fresh", C.c_int32),
        ("reserved", C.c_int32),
    ]


_P = C.c_void_p
_I64 = C.c_int64
_U64 = C.c_uint64
_F = C.c_float
_D = C.c_double
_I = C.c_int
_PD = C.POINTER(C.c_double)
_PI64 = C.POINTER(C.c_int64)
_PF = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol declared in include/sonar_hip.h
SIGNATURES = {
    "sonar_abi_version": (_I, []),
    "sonar_noise_stream_version": (_I, []),
    "sonar_last_error": (C.c_char_p, []),
    "sonar_stats_f32": (_I, [_P, _I64, _P, _P]),
    "sonar_stats_finalize": (_I, [_P, _I64, _I64, _P, _P]),
    "sonar_scale_noise_f32": (_I, [_P, _I64, _F, _I, _F, _P, _I64, _I64, _P]),
    "sonar_scale_noise_stats_f32": (_I, [_P, _I64, _F, _F, _P, _I64, _I64, _P, _P]),
    "sonar_scale_noise_rows_f32": (_I, [_P, _I64, _I64, _F, _P]),
    "sonar_blend_f32": (_I, [_I, _P, _P, _F, _P, _I64, _P]),
    "sonar_blend_tensor_f32": (_I, [_I, _P, _P, _P, _I64, _P, _I64, _P]),
    "sonar_axpby_f32": (_I, [_P, _F, _P, _F, _I64, _P]),
    "sonar_axpby_stats_f32": (_I, [_P, _F, _P, _F, _I64, _P, _P]),
    "sonar_affine_f32": (_I, [_P, _F, _F, _F, _I64, _P]),
    "sonar_scalar_op_f32": (_I, [_I, _P, _P, _F, _P, _I64, _P]),
    "sonar_rowstats_f32": (_I, [_P, _I64, _I64, _P, _P, _P]),
    "sonar_row_affine_f32": (_I, [_I, _P, _I64, _I64, _P, _P, _P, _P]),
    "sonar_powerlaw_f32": (_I, [_P, _F, _I, _I64, _P]),
    "sonar_amax_mid_f32": (_I, [_P, _I64, _I64, _I64, _I, _P, _P]),
    "sonar_div_mid_f32": (_I, [_P, _I64, _I64, _I64, _P, _P]),
    "sonar_mask_mix_f32": (_I, [_P, _P, _P, _I64, _P, _I64, _P]),
    "sonar_minmax_rows_f32": (_I, [_P, _I64, _I64, _P, _P, _P]),
    "sonar_momentum_euler_f32": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _F, C.POINTER(MomentumCfg), _I64, C.POINTER(C.c_int), _P, _P]),
    "sonar_dpmpp_stage1_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, C.POINTER(MomentumCfg), _I64, C.POINTER(C.c_int), _P, _P]),
    "sonar_dpmpp_stage2_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, C.POINTER(MomentumCfg), _I64, C.POINTER(C.c_int), _P, _P]),
    "sonar_norm_decision_f32": (_I, [_P, _I64, _I64, _F, _F, _P, _P]),
    "sonar_apply_norm_f32": (_I, [_P, _I64, _P, _P]),
    "sonar_philox_normal_f32": (_I, [_P, _I64, _U64, _U64, _I64, _P, _P]),
    "sonar_philox_uniform_f32": (_I, [_P, _I64, _U64, _U64, _I64, _F, _F, _F, _P, _P]),
    "sonar_philox_noise_f32": (_I, [_I, _P, _I64, _U64, _U64, _I64, _F, _F, _F, _F, _F, _P, _P]),
    "sonar_philox_noise_ahead_ok": (_I, [_I, _I64, _F]),
    "sonar_philox_noise_ahead_f32": (_I, [_I, _P, _I64, _U64, _U64, _I64, _F, _F, _F, _F, _F, _P, _I, _U64, _P, _P]),
    "sonar_brownian_f32": (_I, [_P, _I64, _I64, _P, _P, _I, _U64, _P, _I64, _P]),
    "sonar_brownian_bridge_acc_f32": (_I, [_P, _P, _P, _F, _P, _F, _P, _F, _I64, _I64, _P, _P, _I, _U64, _P, _I64, _P]),
    "sonar_philox_normal_acc_f32": (_I, [_P, _I64, _U64, _U64, _I64, _P]),
    "sonar_perlin_generate_acc_f32": (_I, [_P, _P, _I64, _I64, _I64, _F, _U64, _U64, _I64, _P]),
    "sonar_brownian_bridge_f32": (_I, [_P, _P, _P, _F, _P, _F, _P, _F, _I64, _I64, _P, _P, _I, _U64, _P, _I64, _P, _P]),
    "sonar_brownian_point_f32": (_I, [_P, _P, _P, _F, _I64, _I64, _P, _P, _I, _U64, _P, _I64, _P]),
    "sonar_perlin_terms_f32": (_I, [_P, _P, _I64, _I64, _I64, _I64, _I, _P]),
    "sonar_perlin_lattice_f32": (_I, [_P, _I64, _I64, _I64, _I64, _I, _U64, _U64, _P]),
    "sonar_perlin_apply_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _F, _P, _P]),
    "sonar_perlin_generate_f32": (_I, [_P, _P, _I64, _I64, _I64, _F, _U64, _U64, _I64, _P, _P]),
    "sonar_perlin_noise_f32": (_I, [_P, _P, _I64, _I64, _I64, _F, _U64, _U64, _I64, _F, _F, _P, _P]),
    "sonar_perlin_noise_ahead_ok": (_I, [_I64, _I64, _I64]),
    "sonar_perlin_noise_ahead_f32": (_I, [_P, _P, _I64, _I64, _F, _U64, _U64, _I64, _F, _F, _P, _I, _U64, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _U64, _P]),
    "sonar_resample_acc_f32": (_I, [_P, _P, _I64, _I64, _I64, _I64, _I64, _F, _I, _I, _P, _P]),
    "sonar_pyramid_generate_f32": (_I, [_P, _I64, _I64, _I64, _I64, C.POINTER(_P), _PI64, _PI64, _PF, _I, _U64, _U64, _I64, _P, _P]),
    "sonar_pyramid_noise_f32": (_I, [_P, _I64, _I64, _I64, _I64, C.POINTER(_P), _PI64, _PI64, _PF, _I, _U64, _U64, _I64, _F, _F, _P, _P]),
    "sonar_pyramid_noise_ahead_f32": (_I, [_P, _I64, _I64, _I64, _I64, _PI64, _PI64, _PF, _I, _U64, _U64, _I64, _F, _F, _P, _I, _U64, _I64, _PI64, _PI64, _PF, _P,
                                           _P]),
    "sonar_levels_sampled_f32": (_I, [_P, _I64, _I64, _I64, _I, _PI64, _PI64, _PF, _PF, _I, _U64, _U64, _I64, _I, _P]),
    "sonar_level_normal_f32": (_I, [_P, _I64, _I64, _I64, _F, _U64, _U64, _I64, _P]),
    "sonar_power_noise_f32": (_I, [_P, _P, _I64, _I64, _I64, _U64, _U64, _I64, _I, _F, _F, _P, _P]),
    "sonar_power_noise_ahead_ok": (_I, [_I64, _I64, _I64, _I]),
    "sonar_power_pipeline": (_I, [_I]),
    "sonar_wcfg_hi_storage": (_I, [_I]),
    "sonar_power_noise_ahead_f32": (_I, [_P, _P, _I64, _I64, _I64, _U64, _U64, _I64, _I, _F, _F, _P, _I, _U64, _P, _P]),
    "sonar_power_spectrum_f32": (_I, [_P, _I64, _I64, _I64, _U64, _U64, _I64, _I, _P]),
    "sonar_power_irfft2_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _U64, _U64, _I64, _I, _P, _P]),
    "sonar_spectral_filter_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _P, _P]),
    "sonar_std_scale_f32": (_I, [_P, _I64, _F, _P, _I64, _I64, _P]),
    "sonar_channel_mix_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _P, _P]),
    "sonar_dwt_out_len": (_I64, [_I64, _I64, _I]),
    "sonar_dwt2_ws_bytes": (_I64, [_I64, _I64, _I64, _I, _I, _I, _I]),
    "sonar_dwt2_fwd_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _PD, _PD, _I, _I, _P, _P]),
    "sonar_dwt2_fwd_f64": (_I, [_P, _P, _P, _I64, _I64, _I64, _PD, _PD, _I, _I, _P, _P]),
    "sonar_dwt2_inv_f32": (_I, [_P, _I64, _I64, _P, _P, _I64, _I64, _I64, _I64, _I64, _PD, _PD, _I, _I, _P, _P]),
    "sonar_dwt2_inv_f64": (_I, [_P, _I64, _I64, _P, _P, _I64, _I64, _I64, _I64, _I64, _PD, _PD, _I, _I, _P, _P]),
    "sonar_sq_acc_f32": (_I, [_P, _P, _F, _I, _I64, _P]),
    "sonar_studentt_f32": (_I, [_P, _P, _F, _F, _F, _I64, _P]),
    "sonar_abs_quantile_rows_f32": (_I, [_P, _I64, _I64, _I64, _F, _P, _P]),
    "sonar_clamp_signpow_rows_f32": (_I, [_P, _I64, _I64, _P, _F, _F, _P]),
    "sonar_mul_table_f32": (_I, [_P, _P, _I64, _I64, _I64, _I, _P]),
    "sonar_laplace_add_f32": (_I, [_P, _P, _F, _F, _F, _I64, _P]),
    "sonar_power_plane_kind": (_I, [_I64, _I64]),
    "sonar_power_block_ws_bytes": (_I64, [_I64, _I64, _I64]),
    "sonar_power_block_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _U64, _U64, _I64, _I, _I, _F, _F, _P, _P]),
    "sonar_rfft2_f32": (_I, [_P, _P, _I64, _I64, _I64, _P]),
    "sonar_cdft_mid_f32": (_I, [_P, _P, _I64, _I64, _I64, _I, _I, _I, _P]),
    "sonar_spectral_logamp_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P]),
    "sonar_spectral_signum_mask_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _I64, _F, _F, _I, _P]),
    "sonar_std_mid_f32": (_I, [_P, _I64, _I64, _I64, _P, _P]),
    "sonar_bcast_gain_f32": (_I, [_P, _P, _I64, _I64, _I64, _I, _F, _F, _P, _P, _P]),
    "sonar_ratio_mix_f32": (_I, [_P, _F, _P, _F, _P, _D, _P, _P, _I64, _P]),
    "sonar_dwt1_fwd_f32": (_I, [_P, _P, _P, _I64, _I64, _PD, _PD, _I, _I, _P]),
    "sonar_dwt1_fwd_f64": (_I, [_P, _P, _P, _I64, _I64, _PD, _PD, _I, _I, _P]),
    "sonar_dwt1_inv_f32": (_I, [_P, _I64, _P, _P, _I64, _I64, _I64, _PD, _PD, _I, _I, _P]),
    "sonar_dwt1_inv_f64": (_I, [_P, _I64, _P, _P, _I64, _I64, _I64, _PD, _PD, _I, _I, _P]),
    "sonar_wcfg_band_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _PD, _PD, _PD, _PD, _I, _D, _P]),
    "sonar_wcfg_band_f64": (_I, [_P, _P, _P, _I64, _I64, _I64, _PD, _PD, _PD, _PD, _I, _D, _P]),
    "sonar_wcfg_band_head_f32": (_I, [_P, _P, _P, _I64, _I64, _PD, _PD, _PD, _PD, _I, _D, _P]),
    "sonar_wcfg_band_head_f64": (_I, [_P, _P, _P, _I64, _I64, _PD, _PD, _PD, _PD, _I, _D, _P]),
    "sonar_wcfg_lowpass_lds_bytes": (_I64, [_I64, _I64, _I, _I, _I, _I, _I]),
    "sonar_wcfg_lowpass_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _PD, _PD, _I, _I, _I, _PD, _D, _D, _I, _P]),
    "sonar_wcfg_lowpass_f64": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _PD, _PD, _I, _I, _I, _PD, _D, _D, _I, _P]),
    "sonar_wcfg_output_f32": (_I, [_P, _P, _I, _P, _I64, _I64, _I64, _I64, _I64, _I, _P]),
    "sonar_wcfg_bands_lds_bytes": (_I64, [_I64, _I64, _I, _I, _I, _I, _I, _I]),
    "sonar_wcfg_bands_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _PD, _PD, _PD, _PD, _I, _I, _I, _PD, _D, _D, _D, _I, _P]),
    "sonar_wcfg_bands_f64": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _PD, _PD, _PD, _PD, _I, _I, _I, _PD, _D, _D, _D, _I, _P]),
    "sonar_minmax_rescale_f32": (_I, [_P, _I64, _I64, _P, _P, _F, _D, _D, _P, _P]),
    "sonar_axis_taps_f32": (_I, [_P, _P, _I64, _I64, _I64, _I64, _P, _P, _I, _I, _P]),
    "sonar_axis_taps_f64": (_I, [_P, _P, _I64, _I64, _I64, _I64, _P, _P, _I, _I, _P]),
    "sonar_dtcwt_q2c_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "sonar_dtcwt_c2q_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "sonar_dtcwt_q2c_f64": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "sonar_dtcwt_c2q_f64": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "sonar_philox_normal_chain_f32": (_I, [_P, _P, _I64, _U64, _U64, _I64, _P]),
    "sonar_perlin_generate_chain_f32": (_I, [_P, _P, _P, _I64, _I64, _F, _U64, _U64, _I64, _P]),
    "sonar_pyramid_generate_acc_f32": (_I, [_P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _I, _U64, _U64, _I64, _P]),
    "sonar_pyramid_generate_acc_ahead_f32": (_I, [_P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _I, _U64, _U64, _I64, _P, _I64, _I64, _I, _U64, _P]),
    "sonar_brownian_bridge_chain_f32": (_I, [_P, _P, _P, _P, _F, _P, _F, _P, _F, _I64, _I64, _P, _P, _I, _U64, _I64, _P]),
    "sonar_signed_rescale_f32": (_I, [_P, _I64, _I64, _D, _D, _D, _D, _F, _P, _P, _P]),
    "sonar_dft_rows_r2c_f32": (_I, [_P, _P, _I64, _I64, _P]),
    "sonar_dft_cols_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _I, _P]),
    "sonar_dft_rows_c2r_f32": (_I, [_P, _P, _I64, _I64, _F, _P, _P]),
    "sonar_max_to_host_f32": (_I, [_P, _I64, C.POINTER(C.c_float), _P]),
    "sonar_max_to_host_begin_f32": (_I, [_P, _I64, _P]),
    "sonar_max_to_host_end_f32": (_I, [C.POINTER(C.c_float), _P]),
    "sonar_wcfg_fused_ws_bytes": (_I64, [_I64, _I64, _I64, _I, _I, _I, _I, _I, _I]),
    "sonar_wcfg_fused_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _PD, _PD, _I, _I, _PD, _PD, _I, _I, _PD, _PD, _I, _D, _I, _I, _P, _I64, _P]),
    "sonar_wcfg_fused_f64": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _PD, _PD, _I, _I, _PD, _PD, _I, _I, _PD, _PD, _I, _D, _I, _I, _P, _I64, _P]),
    "sonar_cast_f32_f64": (_I, [_P, _P, _I64, _P]),
    "sonar_plan_fn_id": (_I, [C.c_char_p]),
    "sonar_plan_fn_nargs": (_I, [_I]),
    "sonar_plan_create": (_P, [_I]),
    "sonar_plan_destroy": (None, [_P]),
    "sonar_plan_length": (_I, [_P]),
    "sonar_plan_add": (_I, [_P, _I, C.POINTER(_U64), _I, _P, _I64, _P, _I]),
    "sonar_plan_run": (_I, [_P, C.POINTER(_U64), _I, _U64, _U64, _P, C.POINTER(_I)]),
    "sonar_pyramid_levels": (_I, [_I64, _I64, _I, _D, _U64, _U64, _PI64, _PI64, _PF]),
}

_lib: Optional[C.CDLL] = None
_recorder = None  # the _Recorder of the call being traced into a plan (section "prepared call plans"), else None


def load() -> C.CDLL:
    """Load libsonar_hip.so or raise (never falls back to anything else)."""
    global _lib
    if _lib is not None:
        # a call being traced into a plan sees the recording proxy -- on the tracing thread only (a preview thread keeps the plain library)
        rec = _recorder  # (one read: the tracing thread sets the global back to None at the end of its trace)
        return _lib if rec is None or rec.thread != _threading.get_ident() else rec.lib
    if not os.path.exists(LIB_PATH):
        raise SonarHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950). There is no non-HIP implementation of this path."
        )
    lib = C.CDLL(os.environ.get("SONAR_HIP_LIB") or LIB_PATH)  # the override is for A/B timing of profiling builds (scratch/)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sonar_last_error().decode("utf-8", "replace")
        raise SonarHipError(f"{what} failed (code {rc}): {msg}")


_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device  # (a replayed step has used the device already: no lazy init to check)
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # ~1 us; torch.cuda.current_stream() builds a Stream object (~10 us)
_last_device = -1  # device index of the tensor most recently checked by _dev (arguments are evaluated before _stream())


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device (every kernel of this library is launched on it).  The tensors
    of the call must live on that device: a launch on another device's memory from this device's stream is never what was meant."""
    cur = torch.cuda.current_device()
    if _last_device >= 0 and _last_device != cur:
        raise SonarHipError(f"tensor on cuda:{_last_device} but the current device is cuda:{cur}: select it (torch.cuda.device / set_device) first")
    if _RAW_STREAM is not None:
        return _RAW_STREAM(cur)
    return torch.cuda.current_stream().cuda_stream


STATS_ATTR = "_sonar_partials"  # see py/utils.py attach_stats / pop_stats
# storage address -> weak reference to the tensor that carries a statistics tag for it.  Raw-pointer kernels do not bump torch's
# version counter, so a write through ANY view of a tagged tensor's storage (sibling views, views of views) must drop the tag:
# keyed by storage, not by tensor object.  Tags live from a producer kernel to the next scale_noise, so this is almost always empty.
TAGGED: dict = {}  # storage address -> list of weak references (several views of one storage may each carry a tag)


def _tag_drop_dead(key: int, ref) -> None:
    refs = TAGGED.get(key)
    if refs is not None:
        refs[:] = [r for r in refs if r is not ref and r() is not None]
        if not refs:
            TAGGED.pop(key, None)


def tag_register(t: torch.Tensor) -> None:
    key = t.untyped_storage().data_ptr()
    TAGGED.setdefault(key, []).append(weakref.ref(t, lambda r, k=key: _tag_drop_dead(k, r)))


def tag_forget(t: torch.Tensor) -> None:
    """The tensor gave its tag up itself (pop_stats): other views of the storage keep theirs."""
    if TAGGED:
        key = t.untyped_storage().data_ptr()
        refs = TAGGED.get(key)
        if refs is not None:
            refs[:] = [r for r in refs if r() is not None and r() is not t]
            if not refs:
                TAGGED.pop(key, None)


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> int:
    global _last_device
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if TAGGED:  # any kernel that touches the storage of a tagged tensor drops its statistics tag (producers tag AFTER their launch)
        for ref in TAGGED.pop(t.untyped_storage().data_ptr(), ()):
            owner = ref()
            if owner is not None:
                owner.__dict__.pop(STATS_ATTR, None)
    if not t.is_cuda:
        raise SonarHipError(f"{name}: tensor lives on {t.device}; the Sonar HIP path only runs on a ROCm device")
    if t.dtype != dtype:
        raise SonarHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise SonarHipError(f"{name}: tensor must be contiguous")
    _last_device = t.device.index
    rec = _recorder  # (one read, as in load())
    if rec is not None and rec.thread == _threading.get_ident():
        rec.seen[t.data_ptr()] = t
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype=torch.float32):
    return None if t is None else _dev(t, name, dtype)


def new_partials(device) -> torch.Tensor:
    """Workspace for the (sum, sumsq) partial pairs of one normalisation point."""
    return torch.empty(NPART * 2, dtype=torch.float64, device=device)


# ------------------------------------------------------------------------------------------------ stats / scale_noise
def stats(x: torch.Tensor, partials: Optional[torch.Tensor] = None) -> torch.Tensor:
    partials = new_partials(x.device) if partials is None else partials
    _check(load().sonar_stats_f32(_dev(x, "x"), x.numel(), _dev(partials, "partials", torch.float64), _stream()), "sonar_stats_f32")
    return partials


def stats_finalize(partials: torch.Tensor, n: int, npart: int = NPART) -> torch.Tensor:
    out = torch.empty(3, dtype=torch.float64, device=partials.device)
    _check(
        load().sonar_stats_finalize(_dev(partials, "partials", torch.float64), npart, n, _dev(out, "out", torch.float64), _stream()),
        "sonar_stats_finalize",
    )
    return out


def scale_noise_(x: torch.Tensor, factor: float, normalized: bool, partials: Optional[torch.Tensor], *,
                 threshold_std_devs: float = 2.5, npart: int = NPART, n_total: Optional[int] = None) -> torch.Tensor:
    n = x.numel()
    _check(
        load().sonar_scale_noise_f32(
            _dev(x, "x"), n, float(factor), int(bool(normalized)), float(threshold_std_devs),
            _opt(partials, "partials", torch.float64), npart, n if n_total is None else n_total, _stream(),
        ),
        "sonar_scale_noise_f32",
    )
    return x


def scale_noise_stats_(x: torch.Tensor, factor: float, partials: torch.Tensor, *, threshold_std_devs: float = 2.5, npart: int = NPART,
                       n_total: Optional[int] = None):
    """scale_noise_(normalized=True) that also returns the (sum, sumsq) partials of the result (derived, no extra pass)."""
    n = x.numel()
    out_partials = new_partials(x.device)
    _check(load().sonar_scale_noise_stats_f32(_dev(x, "x"), n, float(factor), float(threshold_std_devs), _dev(partials, "partials", torch.float64),
                                              npart, n if n_total is None else n_total, out_partials.data_ptr(), _stream()),
           "sonar_scale_noise_stats_f32")
    return x, out_partials


def scale_noise_rows_(x: torch.Tensor, rows: int, inner: int, factor: float) -> torch.Tensor:
    _check(load().sonar_scale_noise_rows_f32(_dev(x, "x"), rows, inner, float(factor), _stream()), "sonar_scale_noise_rows_f32")
    return x


def minmax_rows(x: torch.Tensor, rows: int, inner: int):
    lo = torch.empty(rows, dtype=torch.float32, device=x.device)
    hi = torch.empty(rows, dtype=torch.float32, device=x.device)
    _check(load().sonar_minmax_rows_f32(_dev(x, "x"), rows, inner, _dev(lo, "lo"), _dev(hi, "hi"), _stream()), "sonar_minmax_rows_f32")
    return lo, hi


# ------------------------------------------------------------------------------------------------ elementwise
def blend(mode: str, a: torch.Tensor, b: torch.Tensor, t, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(a) if out is None else out
    if a.shape != b.shape:
        raise SonarHipError(f"blend: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    if isinstance(t, torch.Tensor) and t.numel() > 1:
        if a.numel() % t.numel() != 0:
            raise SonarHipError("blend: weight tensor does not tile the operands")
        _check(
            load().sonar_blend_tensor_f32(BLEND_IDS[mode], _dev(a, "a"), _dev(b, "b"), _dev(t, "t"), t.numel(), _dev(out, "out"), a.numel(), _stream()),
            "sonar_blend_tensor_f32",
        )
    else:
        _check(
            load().sonar_blend_f32(BLEND_IDS[mode], _dev(a, "a"), _dev(b, "b"), float(t), _dev(out, "out"), a.numel(), _stream()),
            "sonar_blend_f32",
        )
    return out


def axpby_(y: torch.Tensor, ymul: float, x: torch.Tensor, xmul: float) -> torch.Tensor:
    if x.shape != y.shape:
        raise SonarHipError(f"axpby: shape mismatch {tuple(x.shape)} vs {tuple(y.shape)}")
    _check(load().sonar_axpby_f32(_dev(y, "y"), float(ymul), _dev(x, "x"), float(xmul), y.numel(), _stream()), "sonar_axpby_f32")
    return y


def axpby_stats_(y: torch.Tensor, ymul: float, x: torch.Tensor, xmul: float, partials: Optional[torch.Tensor] = None):
    """axpby_ that also returns the (sum, sumsq) partials of the result; falls back to axpby_ + stats for unaligned views."""
    if x.shape != y.shape:
        raise SonarHipError(f"axpby: shape mismatch {tuple(x.shape)} vs {tuple(y.shape)}")
    partials = new_partials(y.device) if partials is None else partials
    if (y.data_ptr() | x.data_ptr()) & 15:
        return axpby_(y, ymul, x, xmul), stats(y, partials)
    _check(load().sonar_axpby_stats_f32(_dev(y, "y"), float(ymul), _dev(x, "x"), float(xmul), y.numel(), partials.data_ptr(), _stream()),
           "sonar_axpby_stats_f32")
    return y, partials


def affine_(x: torch.Tensor, sub: float, mul: float, add: float) -> torch.Tensor:
    _check(load().sonar_affine_f32(_dev(x, "x"), float(sub), float(mul), float(add), x.numel(), _stream()), "sonar_affine_f32")
    return x


def mul_scalar(a: torch.Tensor, s: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(a) if out is None else out
    _check(load().sonar_scalar_op_f32(0, _dev(a, "a"), None, float(s), _dev(out, "out"), a.numel(), _stream()), "sonar_scalar_op_f32")
    return out


def div_scalar(a: torch.Tensor, s: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(a) if out is None else out
    _check(load().sonar_scalar_op_f32(1, _dev(a, "a"), None, float(s), _dev(out, "out"), a.numel(), _stream()), "sonar_scalar_op_f32")
    return out


def to_d(x: torch.Tensor, denoised: torch.Tensor, sigma: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(x) if out is None else out
    _check(load().sonar_scalar_op_f32(2, _dev(x, "x"), _dev(denoised, "denoised"), float(sigma), _dev(out, "out"), x.numel(), _stream()),
           "sonar_scalar_op_f32")
    return out


def rowstats(x: torch.Tensor, rows: int, inner: int):
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    std = torch.empty(rows, dtype=torch.float32, device=x.device)
    _check(load().sonar_rowstats_f32(_dev(x, "x"), rows, inner, _dev(mean, "mean"), _dev(std, "std"), _stream()), "sonar_rowstats_f32")
    return mean, std


def row_affine(op: int, x: torch.Tensor, rows: int, inner: int, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    _check(load().sonar_row_affine_f32(op, _dev(x, "x"), rows, inner, _dev(a, "a"), _dev(b, "b"), _dev(out, "out"), _stream()),
           "sonar_row_affine_f32")
    return out


def std_mid(x: torch.Tensor, outer: int, mid: int, inner: int) -> torch.Tensor:
    out = torch.empty(outer * inner, dtype=torch.float32, device=x.device)
    _check(load().sonar_std_mid_f32(_dev(x, "x"), outer, mid, inner, _dev(out, "stdv"), _stream()), "sonar_std_mid_f32")
    return out


def bcast_gain(x: torch.Tensor, stdv: torch.Tensor, outer: int, mid: int, inner: int, bcast: int, abs_strength: float, k: float, *,
               store: bool = True, partials: Optional[torch.Tensor] = None):
    """v = x*k*(1/(std*|strength|+1) + 1); returns (v or None, partials with (sum x^2, sum v^2) per slot)."""
    out = torch.empty_like(x) if store else None
    partials = new_partials(x.device) if partials is None else partials
    _check(load().sonar_bcast_gain_f32(_dev(x, "x"), _dev(stdv, "stdv"), outer, mid, inner, bcast, float(abs_strength), float(k), _opt(out, "out"),
                                       partials.data_ptr(), _stream()), "sonar_bcast_gain_f32")
    return out, partials


def ratio_mix(a: torch.Tensor, a_mul: float, x: torch.Tensor, x_mul: float, num_partials: torch.Tensor, num_mul: float,
              den_partials: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(a) if out is None else out
    _check(load().sonar_ratio_mix_f32(_dev(a, "a"), float(a_mul), _dev(x, "x"), float(x_mul), num_partials.data_ptr(), float(num_mul),
                                      den_partials.data_ptr(), _dev(out, "out"), a.numel(), _stream()), "sonar_ratio_mix_f32")
    return out


def sq_acc_(acc: torch.Tensor, z: torch.Tensor, mul: float, first: bool) -> torch.Tensor:
    _check(load().sonar_sq_acc_f32(_dev(acc, "acc"), _dev(z, "z"), float(mul), int(bool(first)), acc.numel(), _stream()), "sonar_sq_acc_f32")
    return acc


def studentt_(x: torch.Tensor, gamma: torch.Tensor, loc: float, scale: float, df: float) -> torch.Tensor:
    _check(load().sonar_studentt_f32(_dev(x, "x"), _dev(gamma, "gamma"), float(loc), float(scale), float(df), x.numel(), _stream()),
           "sonar_studentt_f32")
    return x


def abs_quantile_rows(x: torch.Tensor, rows: int, inner: int, q: float) -> torch.Tensor:
    """torch.quantile(|x|.reshape(rows, inner), q, dim=-1) (linear interpolation); the rank is formed in fp32 like torch's."""
    rank = torch.tensor(q, dtype=torch.float32) * (inner - 1)
    lo = int(torch.floor(rank).item())
    frac = float((rank - lo).item())
    out = torch.empty(rows, dtype=torch.float32, device=x.device)
    _check(load().sonar_abs_quantile_rows_f32(_dev(x, "x"), rows, inner, min(lo, inner - 1), frac, _dev(out, "out"), _stream()),
           "sonar_abs_quantile_rows_f32")
    return out


def clamp_signpow_rows_(x: torch.Tensor, rows: int, inner: int, limit: torch.Tensor, mul: float, p: float) -> torch.Tensor:
    _check(load().sonar_clamp_signpow_rows_f32(_dev(x, "x"), rows, inner, _dev(limit, "limit"), float(mul), float(p), _stream()),
           "sonar_clamp_signpow_rows_f32")
    return x


def mul_table_(x: torch.Tensor, table: torch.Tensor, inner: int, follow_sign: bool = False) -> torch.Tensor:
    """x[i] *= table[(i / inner) % len(table)] in place (optionally copysign(x, 1 - table))."""
    _check(load().sonar_mul_table_f32(_dev(x, "x"), _dev(table, "table"), x.numel(), int(inner), table.numel(), int(bool(follow_sign)), _stream()),
           "sonar_mul_table_f32")
    return x


def laplace_add_(x: torch.Tensor, u: torch.Tensor, div_fac: float, loc: float, scale: float) -> torch.Tensor:
    """x = x/div_fac + Laplace(loc, scale) from the uniform u in (eps-1, 1), in place."""
    if u.numel() != x.numel():
        raise SonarHipError("laplace_add_: size mismatch")
    _check(load().sonar_laplace_add_f32(_dev(x, "x"), _dev(u, "u"), float(div_fac), float(loc), float(scale), x.numel(), _stream()),
           "sonar_laplace_add_f32")
    return x


def powerlaw_(x: torch.Tensor, alpha: float, use_sign: bool) -> torch.Tensor:
    _check(load().sonar_powerlaw_f32(_dev(x, "x"), float(alpha), int(bool(use_sign)), x.numel(), _stream()), "sonar_powerlaw_f32")
    return x


def amax_mid(x: torch.Tensor, outer: int, mid: int, inner: int, use_abs: bool) -> torch.Tensor:
    peak = torch.empty(outer * inner, dtype=torch.float32, device=x.device)
    _check(load().sonar_amax_mid_f32(_dev(x, "x"), outer, mid, inner, int(bool(use_abs)), _dev(peak, "peak"), _stream()), "sonar_amax_mid_f32")
    return peak


def div_mid_(x: torch.Tensor, outer: int, mid: int, inner: int, d: torch.Tensor) -> torch.Tensor:
    _check(load().sonar_div_mid_f32(_dev(x, "x"), outer, mid, inner, _dev(d, "d"), _stream()), "sonar_div_mid_f32")
    return x


def mask_mix(dst: torch.Tensor, src: torch.Tensor, mask: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(dst) if out is None else out
    if dst.numel() % mask.numel() != 0:
        raise SonarHipError("mask_mix: mask does not tile the operands")
    _check(
        load().sonar_mask_mix_f32(_dev(dst, "dst"), _dev(src, "src"), _dev(mask, "mask"), mask.numel(), _dev(out, "out"), dst.numel(), _stream()),
        "sonar_mask_mix_f32",
    )
    return out


def cast_f32_f64(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty(x.shape, dtype=torch.float64, device=x.device)
    _check(load().sonar_cast_f32_f64(_dev(x, "x"), _dev(out, "out", torch.float64), x.numel(), _stream()), "sonar_cast_f32_f64")
    return out


# ------------------------------------------------------------------------------------------------ momentum
def norm_decision(partials: torch.Tensor, n_total: int, factor: float, threshold_std_devs: float = 2.5) -> torch.Tensor:
    """The decision ``scale_noise(factor, normalized=True)`` would take for a tensor with these (sum, sumsq) partials, left on the device
    (``sonar_noise_norm``, 24 bytes) for the kernel that consumes the tensor; see ``apply_norm_`` and the ``noise_norm`` arguments."""
    out = torch.empty(6, dtype=torch.float32, device=partials.device)
    _check(load().sonar_norm_decision_f32(_dev(partials, "partials", torch.float64), NPART, int(n_total), float(factor), float(threshold_std_devs),
                                          out.data_ptr(), _stream()), "sonar_norm_decision_f32")
    return out


def apply_norm_(x: torch.Tensor, norm: torch.Tensor) -> torch.Tensor:
    """Materialise a pending normalisation in place (what the step kernels do on the fly)."""
    _check(load().sonar_apply_norm_f32(_dev(x, "x"), x.numel(), _dev(norm, "norm"), _stream()), "sonar_apply_norm_f32")
    return x


def momentum_euler(x, denoised, h_in, cfg: MomentumCfg, sigma: float, dt: float, *, noise=None, noise_scale: float = 0.0,
                   x_out=None, h_out=None, noise_norm=None):
    x_out = torch.empty_like(x) if x_out is None else x_out
    h_out = torch.empty_like(x) if h_out is None else h_out
    present = C.c_int(0)
    _check(
        load().sonar_momentum_euler_f32(
            _dev(x, "x"), _dev(denoised, "denoised"), _opt(h_in, "h_in"), _dev(x_out, "x_out"), _dev(h_out, "h_out"),
            _opt(noise, "noise"), float(noise_scale), float(sigma), float(dt), C.byref(cfg), x.numel(), C.byref(present),
            _opt(noise_norm, "noise_norm"), _stream(),
        ),
        "sonar_momentum_euler_f32",
    )
    return x_out, (h_out if present.value else None)


def dpmpp_stage1(x, denoised, h_in, cfg: MomentumCfg, sigma: float, expm1_a: float, ratio_a: float, adj_is_one: bool, *,
                 noise=None, noise_scale: float = 0.0, noise_norm=None):
    x2 = torch.empty_like(x)
    md1 = torch.empty_like(x)
    h_out = torch.empty_like(x)
    present = C.c_int(0)
    _check(
        load().sonar_dpmpp_stage1_f32(
            _dev(x, "x"), _dev(denoised, "denoised"), _opt(h_in, "h_in"), _dev(x2, "x2"), _dev(md1, "md1"), _dev(h_out, "h_out"),
            _opt(noise, "noise"), float(noise_scale), float(sigma), float(expm1_a), float(ratio_a), int(bool(adj_is_one)),
            C.byref(cfg), x.numel(), C.byref(present), _opt(noise_norm, "noise_norm"), _stream(),
        ),
        "sonar_dpmpp_stage1_f32",
    )
    return x2, md1, (h_out if present.value else None)


def dpmpp_stage2(x, denoised2, md1, h_in, cfg: MomentumCfg, sigma_s: float, expm1_b: float, ratio_b: float, fac: float,
                 adj_is_one: bool, *, noise=None, noise_scale: float = 0.0, want_dd: bool = False, noise_norm=None):
    x_out = torch.empty_like(x)
    dd = torch.empty_like(x) if want_dd else None
    h_out = torch.empty_like(x)
    present = C.c_int(0)
    _check(
        load().sonar_dpmpp_stage2_f32(
            _dev(x, "x"), _dev(denoised2, "denoised2"), _dev(md1, "md1"), _opt(h_in, "h_in"), _dev(x_out, "x_out"), _opt(dd, "dd"),
            _dev(h_out, "h_out"), _opt(noise, "noise"), float(noise_scale), float(sigma_s), float(expm1_b), float(ratio_b),
            float(fac), int(bool(adj_is_one)), C.byref(cfg), x.numel(), C.byref(present), _opt(noise_norm, "noise_norm"), _stream(),
        ),
        "sonar_dpmpp_stage2_f32",
    )
    return x_out, dd, (h_out if present.value else None)


# ------------------------------------------------------------------------------------------------ generators
def philox_normal(shape, device, seed: int, stream_id: int, elem_offset: int = 0, partials=None, out=None) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=device) if out is None else out
    _check(
        load().sonar_philox_normal_f32(_dev(out, "out"), out.numel(), seed & (2**64 - 1), stream_id, elem_offset,
                                       _opt(partials, "partials", torch.float64), _stream()),
        "sonar_philox_normal_f32",
    )
    return out


def philox_uniform(shape, device, seed: int, stream_id: int, elem_offset: int = 0, *, sub=0.0, mul=1.0, add=0.0,
                   partials=None, out=None) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=device) if out is None else out
    _check(
        load().sonar_philox_uniform_f32(_dev(out, "out"), out.numel(), seed & (2**64 - 1), stream_id, elem_offset,
                                        float(sub), float(mul), float(add), _opt(partials, "partials", torch.float64), _stream()),
        "sonar_philox_uniform_f32",
    )
    return out


def philox_noise(uniform: bool, shape, device, seed: int, stream_id: int, elem_offset: int, factor: float, *, sub=0.0, mul=1.0, add=0.0,
                 threshold_std_devs: float = 2.5) -> torch.Tensor:
    """Normal / affine-uniform draws + scale_noise(factor, normalized=True), the tensor written once."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    ws = new_partials(device)
    _check(load().sonar_philox_noise_f32(int(bool(uniform)), _dev(out, "out"), out.numel(), seed & (2**64 - 1), stream_id, elem_offset,
                                         float(sub), float(mul), float(add), float(factor), float(threshold_std_devs), ws.data_ptr(), _stream()),
           "sonar_philox_noise_f32")
    return out


def brownian(shape, device, node_ids, coefs, seed: int, elem_offset: int = 0, latent_seeds: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[e] = sum_k coefs[k] * z(node_ids[k], e): one Brownian-interval increment from its expansion over the node normals."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    n = out.numel()
    ids = (C.c_uint64 * len(node_ids))(*[int(v) & (2**64 - 1) for v in node_ids])
    cf = (C.c_float * len(coefs))(*[float(v) for v in coefs])
    latent_elems = n // shape[0]
    _check(load().sonar_brownian_f32(_dev(out, "out"), n, elem_offset, ids, cf, len(node_ids), seed & (2**64 - 1),
                                     None if latent_seeds is None else latent_seeds.data_ptr(), latent_elems, _stream()), "sonar_brownian_f32")
    return out


def brownian_point(shape, device, node_ids, coefs, seed: int, elem_offset: int = 0, latent_seeds: Optional[torch.Tensor] = None, *,
                   prev: Optional[torch.Tensor] = None, scale: float = 1.0, want_out: bool = True, want_w: bool = True):
    """W = sum_k coefs[k] z(node_ids[k], e) (ONE path point); returns (scale * (W - prev) or None, W or None)."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device) if want_out else None
    w = torch.empty(tuple(shape), dtype=torch.float32, device=device) if want_w else None
    n = math.prod(shape)
    ids = (C.c_uint64 * len(node_ids))(*[int(v) & (2**64 - 1) for v in node_ids])
    cf = (C.c_float * len(coefs))(*[float(v) for v in coefs])
    _check(load().sonar_brownian_point_f32(_opt(out, "out"), _opt(w, "w_out"), _opt(prev, "prev"), float(scale), n, elem_offset, ids, cf,
                                           len(node_ids), seed & (2**64 - 1), None if latent_seeds is None else latent_seeds.data_ptr(),
                                           n // shape[0], _stream()), "sonar_brownian_point_f32")
    return out, w


class Accumulate(C.Structure):
    """``sonar_accumulate`` (include/sonar_hip.h): y <- y * y_mul + x * x_mul, partials <- statistics of the new y."""

    _fields_ = [("y", C.c_void_p), ("y_mul", C.c_float), ("x_mul", C.c_float), ("partials", C.c_void_p)]


def accumulate_arg(y: torch.Tensor, y_mul: float, x_mul: float, partials: Optional[torch.Tensor]):
    if y.dtype != torch.float32 or not y.is_contiguous():
        raise SonarHipError("accumulate: the running sum must be a contiguous float32 tensor")
    return C.byref(Accumulate(_dev(y, "y"), float(y_mul), float(x_mul), _opt(partials, "partials", torch.float64)))


def _hosts_pair(y: torch.Tensor, elem_offset: int, pre, chw: int = 4) -> bool:
    """The pair kernel's conditions (sonar_*_chain_f32): whole 4-element groups, aligned tensors, a hostable ``pre``."""
    return (pre is not None and y.numel() % 4 == 0 and elem_offset % 4 == 0 and chw % 4 == 0 and pre.hosted(y, elem_offset)
            and (pre.terms is None or pre.terms.numel() % 4 == 0))


def philox_normal_acc_(y: torch.Tensor, y_mul: float, x_mul: float, seed: int, stream_id: int, elem_offset: int = 0, partials=None,
                       pre=None) -> torch.Tensor:
    """y <- y * y_mul + N(0,1) * x_mul (the draw of ``philox_normal``), in place.  ``pre`` (``FoldPrefix``): the previous chain item's fold,
    applied to y first -- in this launch when the pair kernel can take it, by its own launch otherwise."""
    if _hosts_pair(y, elem_offset, pre):
        pre.consumed = True
        _check(load().sonar_philox_normal_chain_f32(accumulate_arg(y, y_mul, x_mul, partials), pre.arg(), y.numel(), seed & (2**64 - 1), stream_id,
                                                    elem_offset, _stream()), "sonar_philox_normal_chain_f32")
        return y
    if pre is not None:
        pre.apply()
    _check(load().sonar_philox_normal_acc_f32(accumulate_arg(y, y_mul, x_mul, partials), y.numel(), seed & (2**64 - 1), stream_id, elem_offset,
                                              _stream()), "sonar_philox_normal_acc_f32")
    return y


def perlin_generate_acc_(y: torch.Tensor, y_mul: float, x_mul: float, terms: torch.Tensor, div_fac: float, seed: int, stream_id: int,
                         elem_offset: int = 0, partials=None, pre=None) -> torch.Tensor:
    """y <- y * y_mul + perlin * x_mul (the values of ``perlin_generate``), in place; ``pre`` as in ``philox_normal_acc_``."""
    b = y.shape[0]
    chw = y.numel() // max(b, 1)
    if terms.shape[0] == 1 and terms.data_ptr() % 16 == 0 and _hosts_pair(y, elem_offset, pre, chw):
        pre.consumed = True
        _check(load().sonar_perlin_generate_chain_f32(accumulate_arg(y, y_mul, x_mul, partials), pre.arg(), _dev(terms, "terms"), b, chw,
                                                      float(div_fac), seed & (2**64 - 1), stream_id, elem_offset, _stream()),
               "sonar_perlin_generate_chain_f32")
        return y
    if pre is not None:
        pre.apply()
    _check(load().sonar_perlin_generate_acc_f32(accumulate_arg(y, y_mul, x_mul, partials), _dev(terms, "terms"), b, chw,
                                                terms.shape[0], float(div_fac), seed & (2**64 - 1), stream_id, elem_offset, _stream()),
           "sonar_perlin_generate_acc_f32")
    return y


class FoldPrefixArg(C.Structure):
    """``sonar_fold_prefix`` (include/sonar_hip.h)."""

    _fields_ = [("kind", C.c_int32), ("y_mul", C.c_float), ("x_mul", C.c_float), ("div_fac", C.c_float), ("seed", C.c_uint64),
                ("stream_id", C.c_uint64), ("terms", C.c_void_p), ("chw", C.c_int64), ("fresh", C.c_int32)]


PREFIX_NORMAL, PREFIX_PERLIN = 1, 2
_TILE_ELEMS = 4096


class FoldPrefix:
    """The fold y <- y * y_mul + x * x_mul of a tile-keyed generator (Gaussian draw / Perlin), captured with its keys already taken so
    that the NEXT chain item's kernel can evaluate it during its own pass over y (``sonar_brownian_bridge_chain_f32``,
    ``sonar_pyramid_generate_acc_f32``).  ``fresh``: the item is the chain's first -- y is a new tensor that holds nothing yet, y <- x.
    Whoever holds one must end with ``apply()``: a no-op once a kernel has hosted it, the item's own launch otherwise."""

    def __init__(self, kind: int, y: torch.Tensor, y_mul: float, x_mul: float, seed: int, stream_id: int, elem_offset: int,
                 terms: Optional[torch.Tensor] = None, div_fac: float = 1.0, fresh: bool = False, view=None):
        self.kind, self.y, self.y_mul, self.x_mul = kind, y, float(y_mul), float(x_mul)
        self.seed, self.stream_id, self.elem_offset, self.terms, self.div_fac = seed, stream_id, elem_offset, terms, float(div_fac)
        self.fresh = bool(fresh)
        self.view = tuple(view) if view is not None else tuple(y.shape)  # the [B, C, H, W] the generator works on (y may be 5-D)
        self.first_factor = 1.0  # fresh: the factor the chain multiplies the first item by (set by NoiseSampler.fold_prefix)
        self.consumed = False
        if self.fresh and self.x_mul != 1.0:
            raise SonarHipError("FoldPrefix: a fresh prefix carries the first item's raw values")

    def apply(self) -> None:
        if self.consumed:
            return
        self.consumed = True
        y = self.y.view(self.view)
        if self.fresh:
            if self.kind == PREFIX_NORMAL:
                philox_normal(self.view, y.device, self.seed, self.stream_id, self.elem_offset, out=y)
            else:
                perlin_generate(self.view, self.terms, self.div_fac, self.seed, self.stream_id, self.elem_offset, out=y)
        elif self.kind == PREFIX_NORMAL:
            philox_normal_acc_(y, self.y_mul, self.x_mul, self.seed, self.stream_id, self.elem_offset)
        else:
            perlin_generate_acc_(y, self.y_mul, self.x_mul, self.terms, self.div_fac, self.seed, self.stream_id, self.elem_offset)

    def hosted(self, y: torch.Tensor, elem_offset: int, tensors=()) -> bool:
        """Can a kernel working on ``y`` at ``elem_offset`` carry this fold?  (Mirrors the checks of the entry points; the hosting
        kernel adds its own shape conditions.)"""
        if (self.consumed or y.data_ptr() != self.y.data_ptr() or y.numel() != self.y.numel() or elem_offset != self.elem_offset
                or y.shape[0] == 0 or any(t is not None and t.data_ptr() % 16 for t in (y, *tensors))):
            return False
        if self.kind == PREFIX_PERLIN:
            latent = y.numel() // y.shape[0]
            return self.terms is not None and self.terms.numel() == latent and self.terms.data_ptr() % 16 == 0
        return self.kind == PREFIX_NORMAL

    def arg(self):
        return C.byref(FoldPrefixArg(self.kind, self.y_mul, self.x_mul, self.div_fac, self.seed & (2**64 - 1), self.stream_id,
                                     None if self.terms is None else _dev(self.terms, "terms"),
                                     0 if self.terms is None else self.terms.numel(), int(self.fresh)))


def brownian_bridge_acc_(y: torch.Tensor, y_mul: float, x_mul: float, node_ids, coefs, seed: int, elem_offset: int = 0,
                         latent_seeds: Optional[torch.Tensor] = None, *, base_a=None, fa: float = 0.0, base_b=None, fb: float = 0.0, prev=None,
                         scale: float = 1.0, partials=None, want_w: bool = True, pre: Optional[FoldPrefix] = None):
    """The increment of ``brownian_bridge`` folded into y (y <- y * y_mul + scale * (W - prev) * x_mul); returns W (or None).  At most 96
    terms; none at all is fine (W = fa * base_a + fb * base_b).  ``pre``: the previous chain item's fold, applied to y first -- inside
    this launch when the tile kernel can host it, by its own launch before this one otherwise."""
    if len(node_ids) > BROWNIAN_MAX_TERMS:
        raise SonarHipError("brownian_bridge_acc_: expansion too long for one launch")
    w = torch.empty_like(y) if want_w else None
    n = y.numel()
    ids = (C.c_uint64 * len(node_ids))(*[int(v) & (2**64 - 1) for v in node_ids])
    cfs = (C.c_float * len(coefs))(*[float(v) for v in coefs])
    if (pre is not None and latent_seeds is None and (n // max(y.shape[0], 1)) % _TILE_ELEMS == 0
            and pre.hosted(y, elem_offset, (w, prev, base_a, base_b))):
        pre.consumed = True
        _check(load().sonar_brownian_bridge_chain_f32(accumulate_arg(y, y_mul, x_mul, partials), pre.arg(), _opt(w, "w_out"), _opt(prev, "prev"),
                                                      float(scale), _opt(base_a, "base_a"), float(fa), _opt(base_b, "base_b"), float(fb), n,
                                                      elem_offset, ids, cfs, len(node_ids), seed & (2**64 - 1), n // y.shape[0], _stream()),
               "sonar_brownian_bridge_chain_f32")
        return w
    if pre is not None:
        pre.apply()
    _check(load().sonar_brownian_bridge_acc_f32(accumulate_arg(y, y_mul, x_mul, partials), _opt(w, "w_out"), _opt(prev, "prev"), float(scale),
                                                _opt(base_a, "base_a"), float(fa), _opt(base_b, "base_b"), float(fb), n, elem_offset, ids, cfs,
                                                len(node_ids), seed & (2**64 - 1),
                                                None if latent_seeds is None else latent_seeds.data_ptr(), n // y.shape[0], _stream()),
           "sonar_brownian_bridge_acc_f32")
    return w


def brownian_bridge(shape, device, node_ids, coefs, seed: int, elem_offset: int = 0, latent_seeds: Optional[torch.Tensor] = None, *,
                    base_a: Optional[torch.Tensor] = None, fa: float = 0.0, base_b: Optional[torch.Tensor] = None, fb: float = 0.0,
                    prev: Optional[torch.Tensor] = None, scale: float = 1.0, want_out: bool = True, want_w: bool = True, partials=None):
    """W = fa * base_a + fb * base_b + sum_k coefs[k] z(node_ids[k], e); returns (scale * (W - prev) or None, W or None).  More than 96
    terms are accumulated in chunks through ``base_a``.  ``partials``: receives the (sum, sumsq) statistics of the increment."""
    node_ids, coefs = list(node_ids), list(coefs)
    n = math.prod(shape)
    lib = load()
    while True:
        last = len(node_ids) <= BROWNIAN_MAX_TERMS
        ids, cf = node_ids[:BROWNIAN_MAX_TERMS], coefs[:BROWNIAN_MAX_TERMS]
        out = torch.empty(tuple(shape), dtype=torch.float32, device=device) if want_out and last else None
        w = torch.empty(tuple(shape), dtype=torch.float32, device=device) if want_w or not last else None
        _check(lib.sonar_brownian_bridge_f32(_opt(out, "out"), _opt(w, "w_out"), _opt(prev, "prev") if last else None, float(scale),
                                             _opt(base_a, "base_a"), float(fa), _opt(base_b, "base_b"), float(fb), n, elem_offset,
                                             (C.c_uint64 * len(ids))(*[int(v) & (2**64 - 1) for v in ids]), (C.c_float * len(cf))(*[float(v) for v in cf]),
                                             len(ids), seed & (2**64 - 1), None if latent_seeds is None else latent_seeds.data_ptr(),
                                             n // shape[0], _opt(partials, "partials", torch.float64) if last and out is not None else None, _stream()),
               "sonar_brownian_bridge_f32")
        if last:
            return out, w
        node_ids, coefs = node_ids[BROWNIAN_MAX_TERMS:], coefs[BROWNIAN_MAX_TERMS:]
        base_a, fa, base_b, fb = w, 1.0, None, 0.0


def perlin_terms(angles: torch.Tensor, blend_mode: str = "lerp") -> torch.Tensor:
    """angles [iters, C, H+1, W+1] -> terms [iters, C, H, W]"""
    iters, c, gh, gw = angles.shape
    terms = torch.empty((iters, c, gh - 1, gw - 1), dtype=torch.float32, device=angles.device)
    _check(
        load().sonar_perlin_terms_f32(_dev(angles, "angles"), _dev(terms, "terms"), iters, c, gh - 1, gw - 1, BLEND_IDS[blend_mode], _stream()),
        "sonar_perlin_terms_f32",
    )
    return terms


def perlin_lattice(iters: int, c: int, h: int, w: int, device, blend_mode: str, seed: int, stream_id: int) -> torch.Tensor:
    """Sum over `iters` of the Perlin cell-centre terms with in-kernel lattice angles: [1, C, H, W]."""
    out = torch.empty((1, c, h, w), dtype=torch.float32, device=device)
    _check(load().sonar_perlin_lattice_f32(_dev(out, "terms"), iters, c, h, w, BLEND_IDS[blend_mode], seed & (2**64 - 1), stream_id, _stream()),
           "sonar_perlin_lattice_f32")
    return out


def perlin_apply(base: torch.Tensor, terms: torch.Tensor, div_fac: float, partials=None) -> torch.Tensor:
    out = torch.empty_like(base)
    b = base.shape[0]
    chw = base.numel() // max(b, 1)
    _check(
        load().sonar_perlin_apply_f32(_dev(base, "base"), _dev(terms, "terms"), _dev(out, "out"), b, chw, terms.shape[0], float(div_fac),
                                      _opt(partials, "partials", torch.float64), _stream()),
        "sonar_perlin_apply_f32",
    )
    return out


def perlin_generate(shape, terms: torch.Tensor, div_fac: float, seed: int, stream_id: int, elem_offset: int = 0, partials=None,
                    out=None) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=terms.device) if out is None else out
    b = shape[0]
    chw = out.numel() // max(b, 1)
    _check(
        load().sonar_perlin_generate_f32(_dev(terms, "terms"), _dev(out, "out"), b, chw, terms.shape[0], float(div_fac),
                                         seed & (2**64 - 1), stream_id, elem_offset, _opt(partials, "partials", torch.float64), _stream()),
        "sonar_perlin_generate_f32",
    )
    return out


def perlin_noise(shape, terms: torch.Tensor, div_fac: float, seed: int, stream_id: int, elem_offset: int, factor: float,
                 threshold_std_devs: float = 2.5) -> torch.Tensor:
    """generate + scale_noise(factor, normalized=True) with one write of the tensor."""
    out = torch.empty(shape, dtype=torch.float32, device=terms.device)
    b = shape[0]
    chw = out.numel() // max(b, 1)
    ws = new_partials(terms.device)
    _check(
        load().sonar_perlin_noise_f32(_dev(terms, "terms"), _dev(out, "out"), b, chw, terms.shape[0], float(div_fac), seed & (2**64 - 1),
                                      stream_id, elem_offset, float(factor), float(threshold_std_devs), _dev(ws, "ws", torch.float64), _stream()),
        "sonar_perlin_noise_f32",
    )
    return out


def levels_sampled(shape, device, levels, mode: str, seed: int, stream_id: int, plane_offset: int = 0, out: Optional[torch.Tensor] = None):
    """sum_l weight_l * interpolate(level_l, size = shape[-2:], mode) with only the interpolation's taps drawn (``sonar_levels_sampled_f32``):
    ``levels`` = [(h, w, weight, sd), ...], level l a normal field of std sd keyed by (stream_id + l, global element index).  ``out``: added
    to instead of overwritten.  None when the kernel does not carry the request (area mode off whole multiples, more than 16 levels)."""
    b, c, h, w = shape
    n = len(levels)
    if n > 16:
        return None
    accumulate = out is not None
    if out is None:
        out = torch.empty((b, c, h, w), dtype=torch.float32, device=device)
    hs = (C.c_int64 * max(n, 1))(*[int(lv[0]) for lv in levels])
    ws = (C.c_int64 * max(n, 1))(*[int(lv[1]) for lv in levels])
    wts = (C.c_float * max(n, 1))(*[float(lv[2]) for lv in levels])
    sds = (C.c_float * max(n, 1))(*[float(lv[3]) for lv in levels])
    rc = load().sonar_levels_sampled_f32(_dev(out, "out"), b * c, h, w, n, hs, ws, wts, sds, RESAMPLE_IDS[mode], seed & (2**64 - 1), stream_id,
                                         plane_offset, int(accumulate), _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _check(rc, "sonar_levels_sampled_f32")
    return out


def level_normal(shape, device, sd: float, seed: int, stream_id: int, plane_offset: int = 0) -> torch.Tensor:
    """One whole level [b, c, h, w] of ``levels_sampled`` (std ``sd``): the same keys."""
    b, c, h, w = shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=device)
    _check(load().sonar_level_normal_f32(_dev(out, "level"), b * c, h, w, float(sd), seed & (2**64 - 1), stream_id, plane_offset, _stream()),
           "sonar_level_normal_f32")
    return out


def resample_acc_(dst: torch.Tensor, src: torch.Tensor, scale: float, mode: str = "bilinear", accumulate: bool = True, partials=None) -> torch.Tensor:
    """dst[..., H, W] (+)= resample(src[..., h, w]) * scale"""
    H, W = dst.shape[-2:]
    h, w = src.shape[-2:]
    planes = dst.numel() // (H * W)
    if src.numel() // (h * w) != planes:
        raise SonarHipError("resample_acc: plane count mismatch")
    _check(
        load().sonar_resample_acc_f32(_dev(dst, "dst"), _dev(src, "src"), planes, H, W, h, w, float(scale), RESAMPLE_IDS[mode],
                                      int(bool(accumulate)), _opt(partials, "partials", torch.float64), _stream()),
        "sonar_resample_acc_f32",
    )
    return dst


def _pyramid_args(levels):
    n = len(levels)
    ptrs = (_P * max(n, 1))(*[(None if lv[0] is None else _dev(lv[0], "level")) for lv in levels])
    hs = (C.c_int64 * max(n, 1))(*[int(lv[1]) for lv in levels])
    ws = (C.c_int64 * max(n, 1))(*[int(lv[2]) for lv in levels])
    wts = (C.c_float * max(n, 1))(*[float(lv[3]) for lv in levels])
    rule = getattr(levels, "rule", None)
    if rule is not None:  # the tables follow from (seed, stream): a plan recomputes them per call (PATCH_LEVELS)
        hs._sonar_levels = rule
        ptrs._sonar_levels_part = ws._sonar_levels_part = wts._sonar_levels_part = True
    return n, ptrs, hs, ws, wts


def pyramid_generate(shape, device, levels: Sequence, mode: str, seed: int, stream_id: int, elem_offset: int = 0, partials=None):
    """levels: list of (tensor-or-None, h, w, weight); None = drawn on device (see sonar_pyramid_generate_f32).
    Returns None when a level is to be drawn in-kernel and the plane kernel cannot run this shape."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    H, W = shape[-2:]
    planes = out.numel() // (H * W)
    n, ptrs, hs, ws, wts = _pyramid_args(levels)
    rc = load().sonar_pyramid_generate_f32(_dev(out, "out"), planes, H, W, n, ptrs, hs, ws, wts, RESAMPLE_IDS[mode], seed & (2**64 - 1),
                                           stream_id, elem_offset, _opt(partials, "partials", torch.float64), _stream())
    if rc == ERR_UNSUPPORTED and any(lv[0] is None and (lv[1], lv[2]) != (H, W) for lv in levels):
        return None
    _check(rc, "sonar_pyramid_generate_f32")
    return out


def pyramid_generate_acc_(y: torch.Tensor, y_mul: float, x_mul: float, levels: Sequence, mode: str, seed: int, stream_id: int,
                          elem_offset: int = 0, partials=None, pre: Optional[FoldPrefix] = None) -> bool:
    """y[B, C, H, W] <- y * y_mul + pyramid * x_mul (the values of ``pyramid_generate``) in place; ``pre``: the previous chain item's fold,
    applied to y first (inside the same launch when it can be).  False when the plane kernel cannot run the shape: the pyramid values
    were not folded (a ``pre`` that is still unconsumed is the caller's to apply first)."""
    H, W = y.shape[-2:]
    planes = y.numel() // (H * W)
    n, ptrs, hs, ws, wts = _pyramid_args(levels)
    host = pre is not None and pre.hosted(y, elem_offset, [lv[0] for lv in levels])
    if pre is not None and not host:
        pre.apply()
    rc = load().sonar_pyramid_generate_acc_f32(accumulate_arg(y, y_mul, x_mul, partials), pre.arg() if host else None, planes, H, W, n, ptrs, hs,
                                               ws, wts, RESAMPLE_IDS[mode], seed & (2**64 - 1), stream_id, elem_offset, _stream())
    if rc == ERR_UNSUPPORTED:
        return False
    _check(rc, "sonar_pyramid_generate_acc_f32")
    if host:
        pre.consumed = True
    return True


def pyramid_noise(shape, device, levels: Sequence, mode: str, seed: int, stream_id: int, elem_offset: int, factor: float,
                  threshold_std_devs: float = 2.5) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=device)
    H, W = shape[-2:]
    planes = out.numel() // (H * W)
    n, ptrs, hs, ws, wts = _pyramid_args(levels)
    work = new_partials(device)
    rc = load().sonar_pyramid_noise_f32(_dev(out, "out"), planes, H, W, n, ptrs, hs, ws, wts, RESAMPLE_IDS[mode], seed & (2**64 - 1),
                                        stream_id, elem_offset, float(factor), float(threshold_std_devs), _dev(work, "ws", torch.float64), _stream())
    if rc == ERR_UNSUPPORTED and any(lv[0] is None and (lv[1], lv[2]) != (H, W) for lv in levels):
        return None
    _check(rc, "sonar_pyramid_noise_f32")
    return out


def rng_group_for(shape) -> int:
    """RNG group of the device spectrum draws: a function of the channel count only, so every shard of a batch agrees."""
    return 4 if len(shape) >= 3 and shape[-3] % 4 == 0 else 1


def power_irfft2(z: Optional[torch.Tensor], filt: torch.Tensor, shape, *, seed: int = 0, stream_id: int = 0, plane_offset: int = 0,
                 partials=None) -> torch.Tensor:
    """out[shape] = irfft2(z * filt, norm='ortho'); z = None draws the spectrum on device."""
    H, W = shape[-2:]
    out = torch.empty(shape, dtype=torch.float32, device=filt.device)
    planes = out.numel() // (H * W)
    zp = None
    if z is not None:
        if z.dtype != torch.complex64 or not z.is_cuda or not z.is_contiguous():
            raise SonarHipError("power_irfft2: z must be a contiguous complex64 device tensor")
        if z.numel() != planes * H * (W // 2 + 1):
            raise SonarHipError("power_irfft2: spectrum size mismatch")
        zp = z.data_ptr()
    if filt.numel() != H * (W // 2 + 1):
        raise SonarHipError("power_irfft2: filter size mismatch")
    kind = power_plane_kind(H, W)
    if kind == 4 and z is None:
        return _power_block(0, filt, out, seed, stream_id, plane_offset, partials=partials)
    if kind in (3, 4):
        if z is None:
            # the reference's own route: white noise, rfft2, x filter, irfft2 (py/nodes/powernoise.py:356-366); global element keys
            white = philox_normal(tuple(shape), filt.device, seed, stream_id, plane_offset * H * W)
            return _direct_spectral_filter(white, filt, partials)
        return _direct_inverse(z, filt, out, 1.0 / math.sqrt(H * W), partials)
    _check(
        load().sonar_power_irfft2_f32(zp, _dev(filt, "filter"), _dev(out, "out"), planes, H, W, seed & (2**64 - 1), stream_id, plane_offset,
                                      rng_group_for(shape), _opt(partials, "partials", torch.float64), _stream()),
        "sonar_power_irfft2_f32",
    )
    return out


class PowerLookahead:
    """What one power-noise sampler remembers between calls (``power_noise(lookahead=...)``): the statistics the previous call's
    final pass computed for the stream id it expected next, and the key (filter, shape, seed, stream, offset, group) they belong to.
    A sampler's calls take consecutive stream ids (one ``DeviceRNG.take`` per call; a constant step when other generators draw in
    between), so the expectation is the last stream id plus the last step."""

    __slots__ = ("key", "partials", "last_stream", "last_delta", "step", "hits", "misses")

    def __init__(self):
        self.key = None
        self.partials = None
        self.last_stream = None
        self.last_delta = None
        self.step = 1
        self.hits = 0
        self.misses = 0


def power_noise(filt: torch.Tensor, shape, *, seed: int, stream_id: int, plane_offset: int, factor: float,
                threshold_std_devs: float = 2.5, lookahead: Optional[PowerLookahead] = None) -> torch.Tensor:
    """draw + filter + irfft2 + scale_noise(factor, normalized=True); the tensor is written once.  ``lookahead``: on the pipelined path
    the call also leaves the statistics of the call expected next (idle waves of its final pass), and skips its own statistics launch
    when the previous call left them -- same output bits with or without."""
    H, W = shape[-2:]
    out = torch.empty(shape, dtype=torch.float32, device=filt.device)
    planes = out.numel() // (H * W)
    kind = power_plane_kind(H, W)
    if kind == 4:
        return _power_block(1, filt, out, seed, stream_id, plane_offset, partials=new_partials(filt.device), factor=factor,
                            threshold_std_devs=threshold_std_devs)
    if kind == 3:
        ws = new_partials(filt.device)
        white = philox_normal(tuple(shape), filt.device, seed, stream_id, plane_offset * H * W)
        return scale_noise_(_direct_spectral_filter(white, filt, ws), factor, True, ws, threshold_std_devs=threshold_std_devs)
    group = rng_group_for(shape)
    lib = load()
    seed &= 2**64 - 1
    if lookahead is not None and lib.sonar_power_noise_ahead_ok(planes, H, W, group):
        st = _stream()

        def key_for(stream, seed=seed, st=st):
            # the HIP stream is part of the key: the statistics are written and read in stream order only
            return (filt.data_ptr(), filt._version, tuple(shape), seed, stream, plane_offset, group, filt.device, st)

        have = lookahead.key is not None and lookahead.key == key_for(stream_id)
        ws = lookahead.partials if have else new_partials(filt.device)
        if lookahead.last_stream is not None and stream_id > lookahead.last_stream:
            delta = stream_id - lookahead.last_stream
            if lookahead.last_delta is None or delta == lookahead.last_delta:
                lookahead.step = delta  # a step seen twice in a row (or the first one seen); a one-off shift is not adopted
            lookahead.last_delta = delta
        nxt = (stream_id + lookahead.step) & (2**64 - 1)
        nws = new_partials(filt.device)
        _check(
            lib.sonar_power_noise_ahead_f32(_dev(filt, "filter"), _dev(out, "out"), planes, H, W, seed, stream_id, plane_offset, group,
                                            float(factor), float(threshold_std_devs), _dev(ws, "ws", torch.float64), int(have), nxt,
                                            _dev(nws, "ws_next", torch.float64), st),
            "sonar_power_noise_ahead_f32",
        )
        lookahead.hits += int(have)
        lookahead.misses += int(not have)
        lookahead.key, lookahead.partials, lookahead.last_stream = key_for(nxt), nws, stream_id
        rec = _recorder  # (one read: another thread's trace may end meanwhile)
        if rec is not None and rec.thread == _threading.get_ident():  # THIS thread's call is being traced: a plan replays the steady state only
            if have and lookahead.last_delta == lookahead.step:
                rec.hooks.append(_PowerAheadHook(lookahead, key_for, ws, nws, stream_id - rec.base, lookahead.step, filt.device))
            else:
                rec.fail("power-law look-ahead not in its steady state")
        return out
    ws = new_partials(filt.device)
    _check(
        lib.sonar_power_noise_f32(_dev(filt, "filter"), _dev(out, "out"), planes, H, W, seed, stream_id, plane_offset,
                                  group, float(factor), float(threshold_std_devs), _dev(ws, "ws", torch.float64), _stream()),
        "sonar_power_noise_f32",
    )
    return out


def spectral_filter(x: torch.Tensor, filt: torch.Tensor, partials: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = irfft2(rfft2(x, 'ortho') * filt, 'ortho') over the last two dims; filt is real [H, W/2+1]."""
    H, W = int(x.shape[-2]), int(x.shape[-1])
    planes = x.numel() // (H * W)
    if filt.numel() != H * (W // 2 + 1):
        raise SonarHipError("spectral_filter: filter size mismatch")
    if power_plane_kind(H, W) in (3, 4):
        return _direct_spectral_filter(x, filt, partials)
    out = torch.empty_like(x)
    _check(load().sonar_spectral_filter_f32(_dev(x, "x"), _dev(filt, "filter"), _dev(out, "out"), planes, H, W,
                                            _opt(partials, "partials", torch.float64), _stream()), "sonar_spectral_filter_f32")
    return out


def rfft2(x: torch.Tensor) -> torch.Tensor:
    """torch.fft.rfft2(x) (unscaled) over the last two dims as a complex64 tensor [..., H, W/2+1]: LDS-resident for the power-of-two
    planes, two direct passes (rows r2c, columns) for every other size (lines of at most 2048)."""
    H, W = int(x.shape[-2]), int(x.shape[-1])
    K = W // 2 + 1
    z = torch.empty((*x.shape[:-1], K), dtype=torch.complex64, device=x.device)
    _dev(x, "x")
    planes = x.numel() // (H * W)
    lib = load()
    if int(lib.sonar_power_plane_kind(H, W)) == 1:
        _check(lib.sonar_rfft2_f32(x.data_ptr(), z.data_ptr(), planes, H, W, _stream()), "sonar_rfft2_f32")
        return z
    rows = torch.empty_like(z)
    _check(lib.sonar_dft_rows_r2c_f32(x.data_ptr(), rows.data_ptr(), planes * H, W, _stream()), "sonar_dft_rows_r2c_f32")
    _check(lib.sonar_dft_cols_f32(rows.data_ptr(), None, z.data_ptr(), planes, H, K, 0, _stream()), "sonar_dft_cols_f32")
    return z


def cdft_mid(z: torch.Tensor, outer: int, C: int, inner: int, *, inverse: bool, real_out: bool = False) -> torch.Tensor:
    """DFT along the middle axis of z viewed as [outer][C][inner]; ``z`` complex64 or (real input) float32.  Inverse: no 1/C."""
    real_in = z.dtype == torch.float32
    if not real_in and z.dtype != torch.complex64:
        raise SonarHipError("cdft_mid: float32 or complex64 input")
    if not z.is_cuda or not z.is_contiguous():
        raise SonarHipError("cdft_mid: contiguous device tensor required")
    out = torch.empty(z.shape, dtype=torch.float32 if real_out else torch.complex64, device=z.device)
    _check(load().sonar_cdft_mid_f32(z.data_ptr(), out.data_ptr(), outer, C, inner, int(bool(inverse)), int(real_in), int(bool(real_out)), _stream()),
           "sonar_cdft_mid_f32")
    return out


def spectral_logamp(z: torch.Tensor, planes: int, C: int, H: int, W: int):
    """(la, full): log amplitude of the complex64 spectrum z [planes, H, Wz] and |la| over the full H x W spectrum (see the header)."""
    Wz = int(z.shape[-1])
    la = torch.empty(z.shape, dtype=torch.float32, device=z.device)
    full = torch.empty((planes, H, W), dtype=torch.float32, device=z.device)
    _check(load().sonar_spectral_logamp_f32(z.data_ptr(), _dev(la, "la"), _dev(full, "full"), planes, C, H, W, Wz, _stream()), "sonar_spectral_logamp_f32")
    return la, full


def spectral_signum_mask_(z: torch.Tensor, la: torch.Tensor, q: torch.Tensor, planes: int, C: int, plane_elems: int, intensity: float, gain: float,
                          channel_sym: bool = False):
    _check(load().sonar_spectral_signum_mask_f32(z.data_ptr(), _dev(la, "la"), _dev(q, "q"), q.shape[0], planes, C, plane_elems, float(intensity),
                                                 float(gain), int(bool(channel_sym)), _stream()), "sonar_spectral_signum_mask_f32")
    return z


def std_scale_(x: torch.Tensor, mul: float, partials: torch.Tensor) -> torch.Tensor:
    _check(load().sonar_std_scale_f32(_dev(x, "x"), x.numel(), float(mul), _dev(partials, "partials", torch.float64), NPART, x.numel(),
                                      _stream()), "sonar_std_scale_f32")
    return x


def power_spectrum(shape, device, *, seed: int, stream_id: int, plane_offset: int = 0) -> torch.Tensor:
    """The complex64 half-spectrum [..., H, W/2+1] the device-mode power kernels draw."""
    H, W = shape[-2:]
    z = torch.empty((*shape[:-1], W // 2 + 1), dtype=torch.complex64, device=device)
    planes = z.numel() // (H * (W // 2 + 1))
    if power_plane_kind(H, W) == 4:
        _check(load().sonar_power_block_f32(None, z.data_ptr(), None, planes, H, W, seed & (2**64 - 1), stream_id, plane_offset, rng_group_for(shape), 2,
                                            1.0, 0.0, None, _stream()), "sonar_power_block_f32")
        return z
    _check(load().sonar_power_spectrum_f32(z.data_ptr(), planes, H, W, seed & (2**64 - 1), stream_id, plane_offset, rng_group_for(shape), _stream()),
           "sonar_power_spectrum_f32")
    return z


DIRECT_DFT_MAX = 2048


def _power_block(mode: int, filt: torch.Tensor, out: torch.Tensor, seed: int, stream_id: int, plane_offset: int, *, partials=None, factor: float = 1.0,
                 threshold_std_devs: float = 2.5) -> torch.Tensor:
    """Kind-4 planes (half-spectrum beyond LDS), spectrum drawn on device: draw + filter + columns into a complex workspace, rows out of it
    (``sonar_power_block_f32``); mode 1 writes the tensor normalised (Parseval statistics first)."""
    H, W = out.shape[-2:]
    planes = out.numel() // (H * W)
    ws = torch.empty((planes, H, W // 2 + 1), dtype=torch.complex64, device=out.device)
    _check(load().sonar_power_block_f32(_dev(filt, "filter"), _dev(ws, "ws", torch.complex64), _dev(out, "out"), planes, H, W, seed & (2**64 - 1), stream_id, plane_offset,
                                        rng_group_for(out.shape), mode, float(factor), float(threshold_std_devs), _opt(partials, "partials", torch.float64),
                                        _stream()), "sonar_power_block_f32")
    return out


def power_plane_kind(H: int, W: int) -> int:
    """1: fixed-size LDS kernels, 2: general-size LDS kernels (even sizes that fit), 4: beyond LDS, generated in column blocks (a supplied
    spectrum / the spectral filter: the direct passes, as 3), 3: direct DFT passes (any size up to 2048), 0: none."""
    kind = int(load().sonar_power_plane_kind(int(H), int(W)))
    if kind == 0 and 1 <= H <= DIRECT_DFT_MAX and 1 <= W <= DIRECT_DFT_MAX:
        return 3
    return kind


def power_supported(H: int, W: int) -> bool:
    """True when the spectral kernels take an H x W plane: LDS-resident FFTs, or the direct DFT passes for the rest."""
    return power_plane_kind(H, W) != 0


def _direct_inverse(z: torch.Tensor, filt: Optional[torch.Tensor], out: torch.Tensor, scale: float, partials) -> torch.Tensor:
    """out = scale * irfft2-unnormalised(z * filt) for complex64 z [planes, H, W/2+1] through the direct passes."""
    H, W = out.shape[-2:]
    K = W // 2 + 1
    planes = out.numel() // (H * W)
    lib = load()
    ws = torch.empty((planes, H, K), dtype=torch.complex64, device=out.device)
    _check(lib.sonar_dft_cols_f32(z.data_ptr(), None if filt is None else _dev(filt, "filter"), ws.data_ptr(), planes, H, K, 1, _stream()),
           "sonar_dft_cols_f32")
    _check(lib.sonar_dft_rows_c2r_f32(ws.data_ptr(), _dev(out, "out"), planes * H, W, float(scale), _opt(partials, "partials", torch.float64),
                                      _stream()), "sonar_dft_rows_c2r_f32")
    return out


def _direct_spectral_filter(x: torch.Tensor, filt: torch.Tensor, partials) -> torch.Tensor:
    H, W = int(x.shape[-2]), int(x.shape[-1])
    K = W // 2 + 1
    planes = x.numel() // (H * W)
    lib = load()
    a = torch.empty((planes, H, K), dtype=torch.complex64, device=x.device)
    b = torch.empty_like(a)
    _check(lib.sonar_dft_rows_r2c_f32(_dev(x, "x"), a.data_ptr(), planes * H, W, _stream()), "sonar_dft_rows_r2c_f32")
    # forward columns, x filter, inverse columns in one pass over the workspace when the columns go through LDS (mode 2)
    if lib.sonar_dft_cols_f32(a.data_ptr(), _dev(filt, "filter"), b.data_ptr(), planes, H, K, 2, _stream()) == 0:
        out = torch.empty_like(x)
        _check(lib.sonar_dft_rows_c2r_f32(b.data_ptr(), _dev(out, "out"), planes * H, W, 1.0 / (H * W), _opt(partials, "partials", torch.float64),
                                          _stream()), "sonar_dft_rows_c2r_f32")
        return out
    _check(lib.sonar_dft_cols_f32(a.data_ptr(), None, b.data_ptr(), planes, H, K, 0, _stream()), "sonar_dft_cols_f32")
    return _direct_inverse(b, filt, torch.empty_like(x), 1.0 / (H * W), partials)


def channel_mix(x: torch.Tensor, mixer: torch.Tensor, partials=None) -> torch.Tensor:
    b, c = x.shape[:2]
    hw = x.numel() // (b * c)
    out = torch.empty_like(x)
    _check(
        load().sonar_channel_mix_f32(_dev(x, "x"), _dev(mixer, "mixer"), _dev(out, "out"), b, c, hw, _opt(partials, "partials", torch.float64), _stream()),
        "sonar_channel_mix_f32",
    )
    return out


# ------------------------------------------------------------------------------------------------ DWT / WaveletCFG
def _darr(vals):
    vals = [float(v) for v in vals]
    return (C.c_double * len(vals))(*vals)


def _wavelet_dtype(t: torch.Tensor) -> str:
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise SonarHipError(f"DWT: float32 or float64 only (got {t.dtype})")


def dwt_out_len(n: int, flen: int, mode: str) -> int:
    return int(load().sonar_dwt_out_len(n, flen, DWT_MODE_IDS[mode]))


def dwt2_forward(x: torch.Tensor, dec_lo, dec_hi, mode: str):
    """One analysis level: x[..., H, W] -> (ll[..., h, w], hi[..., 3, h, w])."""
    kind = _wavelet_dtype(x)
    H, W = x.shape[-2:]
    lead = tuple(x.shape[:-2])
    planes = x.numel() // (H * W)
    flen, m = len(dec_lo), DWT_MODE_IDS[mode]
    h, w = dwt_out_len(H, flen, mode), dwt_out_len(W, flen, mode)
    ll = torch.empty((*lead, h, w), dtype=x.dtype, device=x.device)
    hi = torch.empty((*lead, 3, h, w), dtype=x.dtype, device=x.device)
    ws = torch.empty(max(int(load().sonar_dwt2_ws_bytes(planes, H, W, flen, m, x.element_size(), 0)), 8), dtype=torch.uint8, device=x.device)
    fn = load().sonar_dwt2_fwd_f32 if kind == "f32" else load().sonar_dwt2_fwd_f64
    _check(fn(_dev(x, "x", x.dtype), _dev(ll, "ll", x.dtype), _dev(hi, "hi", x.dtype), planes, H, W, _darr(dec_lo), _darr(dec_hi), flen, m,
              ws.data_ptr(), _stream()), f"sonar_dwt2_fwd_{kind}")
    return ll, hi


def dwt2_inverse(ll: torch.Tensor, hi: torch.Tensor, rec_lo, rec_hi, mode: str, out_hw=None) -> torch.Tensor:
    """One synthesis level.  ``ll`` may be one row/column larger than the band (its leading block is used)."""
    kind = _wavelet_dtype(hi)
    if ll.dtype != hi.dtype:
        raise SonarHipError("DWT inverse: ll / hi dtype mismatch")
    h, w = hi.shape[-2:]
    lead = tuple(hi.shape[:-3])
    planes = hi.numel() // (3 * h * w)
    ll_h, ll_w = ll.shape[-2:]
    flen, m = len(rec_lo), DWT_MODE_IDS[mode]
    full_h = 2 * h if mode == "periodization" else 2 * h - flen + 2
    full_w = 2 * w if mode == "periodization" else 2 * w - flen + 2
    Ho, Wo = (full_h, full_w) if out_hw is None else out_hw
    out = torch.empty((*lead, Ho, Wo), dtype=hi.dtype, device=hi.device)
    ws = torch.empty(max(int(load().sonar_dwt2_ws_bytes(planes, h, w, flen, m, hi.element_size(), 1)), 8), dtype=torch.uint8, device=hi.device)
    fn = load().sonar_dwt2_inv_f32 if kind == "f32" else load().sonar_dwt2_inv_f64
    _check(fn(_dev(ll, "ll", hi.dtype), ll_h, ll_w, _dev(hi, "hi", hi.dtype), _dev(out, "out", hi.dtype), planes, h, w, Ho, Wo,
              _darr(rec_lo), _darr(rec_hi), flen, m, ws.data_ptr(), _stream()), f"sonar_dwt2_inv_{kind}")
    return out


def dwt1_forward(x: torch.Tensor, dec_lo, dec_hi, mode: str):
    """One 1-D analysis level along the last axis: x[..., L] -> (lo[..., n], hi[..., n])."""
    kind = _wavelet_dtype(x)
    L = x.shape[-1]
    rows = x.numel() // max(L, 1)
    flen, m = len(dec_lo), DWT_MODE_IDS[mode]
    n = dwt_out_len(L, flen, mode)
    lo = torch.empty((*x.shape[:-1], n), dtype=x.dtype, device=x.device)
    hi = torch.empty_like(lo)
    fn = load().sonar_dwt1_fwd_f32 if kind == "f32" else load().sonar_dwt1_fwd_f64
    _check(fn(_dev(x, "x", x.dtype), _dev(lo, "lo", x.dtype), _dev(hi, "hi", x.dtype), rows, L, _darr(dec_lo), _darr(dec_hi), flen, m, _stream()),
           f"sonar_dwt1_fwd_{kind}")
    return lo, hi


def dwt1_inverse(lo: torch.Tensor, hi: torch.Tensor, rec_lo, rec_hi, mode: str, out_len: Optional[int] = None) -> torch.Tensor:
    """One 1-D synthesis level.  ``lo`` may be one sample longer than the band (its leading samples are used)."""
    kind = _wavelet_dtype(hi)
    if lo.dtype != hi.dtype:
        raise SonarHipError("DWT inverse: lo / hi dtype mismatch")
    n = hi.shape[-1]
    rows = hi.numel() // max(n, 1)
    flen, m = len(rec_lo), DWT_MODE_IDS[mode]
    full = 2 * n if mode == "periodization" else 2 * n - flen + 2
    Lo = full if out_len is None else int(out_len)
    out = torch.empty((*hi.shape[:-1], Lo), dtype=hi.dtype, device=hi.device)
    fn = load().sonar_dwt1_inv_f32 if kind == "f32" else load().sonar_dwt1_inv_f64
    _check(fn(_dev(lo, "lo", hi.dtype), lo.shape[-1], _dev(hi, "hi", hi.dtype), _dev(out, "out", hi.dtype), rows, n, Lo, _darr(rec_lo),
              _darr(rec_hi), flen, m, _stream()), f"sonar_dwt1_inv_{kind}")
    return out


def wcfg_band(cond: torch.Tensor, uncond: torch.Tensor, groups: int, s_cond, s_uncond, s_diff, s_final, blend_mode: str, strength: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    kind = _wavelet_dtype(cond)
    out = torch.empty_like(cond) if out is None else out
    # elements that share a scale: h * w of a [B, C, groups, h, w] band (h * w * 2 of a DTCWT band [B, C, 6, h, w, 2])
    group_size = math.prod(cond.shape[3:]) if cond.ndim >= 5 else cond.shape[-1] * cond.shape[-2]
    fn = load().sonar_wcfg_band_f32 if kind == "f32" else load().sonar_wcfg_band_f64

    def arr(v):
        return None if v is None else _darr(v)

    _check(fn(_dev(cond, "cond", cond.dtype), _dev(uncond, "uncond", cond.dtype), _dev(out, "out", cond.dtype), cond.numel(), group_size, groups,
              arr(s_cond), arr(s_uncond), arr(s_diff), arr(s_final), BLEND_IDS[blend_mode], float(strength), _stream()), f"sonar_wcfg_band_{kind}")
    return out


def wcfg_band_head(cond: torch.Tensor, uncond: torch.Tensor, s_cond, s_uncond, s_diff, s_final, blend_mode: str, strength: float,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The band arithmetic on a 1-D band [B, C, l]: the four scales (floats or None) act on coefficient 0 of each row only, as the
    reference's ``ht[:, :, lidx]`` does there (py/wavelet_functions.py:212-215); the blend acts on every coefficient."""
    kind = _wavelet_dtype(cond)
    out = torch.empty_like(cond) if out is None else out
    fn = load().sonar_wcfg_band_head_f32 if kind == "f32" else load().sonar_wcfg_band_head_f64

    def arr(v):
        return None if v is None else _darr([float(v)])

    _check(fn(_dev(cond, "cond", cond.dtype), _dev(uncond, "uncond", cond.dtype), _dev(out, "out", cond.dtype), cond.numel(), cond.shape[-1],
              arr(s_cond), arr(s_uncond), arr(s_diff), arr(s_final), BLEND_IDS[blend_mode], float(strength), _stream()), f"sonar_wcfg_band_head_{kind}")
    return out


def band_scale_head_(band: torch.Tensor, scale: float) -> torch.Tensor:
    """band[:, :, 0] *= scale in place for a 1-D band [B, C, l] (see wcfg_band_head): the band rides in the uncond slot, whose scale
    reaches coefficient 0 only, and an inject blend of strength 0 returns it (u + d * 0)."""
    if float(scale) == 1.0:
        return band
    return wcfg_band_head(band, band, None, float(scale), None, None, "inject", 0.0, out=band)


def band_scale_(band: torch.Tensor, scales) -> torch.Tensor:
    """band[..., g, :, :] *= scales[g] in place (wavelet_scaling, py/wavelet_functions.py:193-216).  Uses the WaveletCFG band
    kernel with uncond scale 0 and lerp strength 1: out = fma(0, d - 0, d) = cond * scale exactly."""
    groups = len(scales)
    if all(float(v) == 1.0 for v in scales):
        return band
    return wcfg_band(band, band, groups, list(scales), [0.0] * groups, None, None, "lerp", 1.0, out=band)


def wcfg_output(x: Optional[torch.Tensor], result: torch.Tensor, shape, subtract_from_x: bool) -> torch.Tensor:
    """out[shape] (fp32) = x - (float)crop(result)  or  (float)crop(result)."""
    H, W = shape[-2:]
    Hr, Wr = result.shape[-2:]
    out = torch.empty(tuple(shape), dtype=torch.float32, device=result.device)
    planes = out.numel() // (H * W)
    _check(load().sonar_wcfg_output_f32(_opt(x, "x"), _dev(result, "result", result.dtype), int(result.dtype == torch.float64), _dev(out, "out"),
                                        planes, H, W, Hr, Wr, int(bool(subtract_from_x)), _stream()), "sonar_wcfg_output_f32")
    return out


_LOW_OK: dict = {}


def wcfg_lowpass_plan(cond: torch.Tensor, uncond: torch.Tensor, x: Optional[torch.Tensor], *, levels: int, dec_lo, rec_lo, mode: str, inv_mode: str, g,
                      ku: float, kt: float, subtract_from_x: bool, high_precision: bool):
    """Everything of ``wcfg_lowpass`` but the launch: a zero-argument callable that launches and returns the output tensor, or None
    when the plane's pyramid does not fit in LDS.  WaveletCFG builds it while its sigma read is in flight."""
    B, Cc, H, W = cond.shape
    lib = load()
    elem = 8 if high_precision else 4
    key = (H, W, levels, len(dec_lo), mode, inv_mode, elem)
    ok = _LOW_OK.get(key)
    if ok is None:
        ok = _LOW_OK[key] = len(dec_lo) == len(rec_lo) and lib.sonar_wcfg_lowpass_lds_bytes(H, W, levels, len(dec_lo), DWT_MODE_IDS[mode],
                                                                                            DWT_MODE_IDS[inv_mode], elem) >= 0
    if not ok:
        return None
    if len(g) != levels + 1:
        raise SonarHipError("wcfg_lowpass: g must hold levels + 1 weights")
    out = torch.empty_like(cond)
    fn = lib.sonar_wcfg_lowpass_f64 if high_precision else lib.sonar_wcfg_lowpass_f32
    call = (_dev(cond, "cond"), _dev(uncond, "uncond"), _opt(x, "x"), _dev(out, "out"), B * Cc, H, W, levels, _taps_arr(dec_lo), _taps_arr(rec_lo),
            len(dec_lo), DWT_MODE_IDS[mode], DWT_MODE_IDS[inv_mode], _darr([float(v) for v in g]), float(ku), float(kt), int(bool(subtract_from_x)),
            _stream())
    keep = (cond, uncond, x)

    def launch():
        _check(fn(*call), "sonar_wcfg_lowpass")
        return out

    launch.keep = keep
    return launch


def wcfg_lowpass(cond: torch.Tensor, uncond: torch.Tensor, x: Optional[torch.Tensor], *, levels: int, dec_lo, rec_lo, mode: str, inv_mode: str, g,
                 ku: float, kt: float, subtract_from_x: bool, high_precision: bool) -> Optional[torch.Tensor]:
    """WaveletCFG for difference-only rules with one detail scale per level, ONE launch (``sonar_wcfg_lowpass_*``: low-pass pyramid in
    LDS, 16N bytes of HBM traffic per latent); None when the plane's pyramid does not fit in LDS (caller: ``wcfg_fused``)."""
    launch = wcfg_lowpass_plan(cond, uncond, x, levels=levels, dec_lo=dec_lo, rec_lo=rec_lo, mode=mode, inv_mode=inv_mode, g=g, ku=ku, kt=kt,
                               subtract_from_x=subtract_from_x, high_precision=high_precision)
    return None if launch is None else launch()


_WCFG_WS: dict = {}
_WCFG_NEED: dict = {}
_TAPS: dict = {}


def wcfg_bands(a: torch.Tensor, b: Optional[torch.Tensor], x: Optional[torch.Tensor], out: Optional[torch.Tensor] = None, *, levels: int, dec_lo,
               dec_hi, rec_lo, rec_hi, mode: str, inv_mode: str, yh_scales, yl_scale: float, ku: float, kt: float, subtract_from_x: bool,
               high_precision: bool) -> Optional[torch.Tensor]:
    """x - (ku * b + kt * Phi(a - b)) (or without the x) with Phi(v) = IDWT(D DWT(v)), D = ``yl_scale`` for the approximation and
    ``yh_scales[level][3]`` (cH, cV, cD; finest level first) for the details: ONE launch, the coefficients stay in LDS
    (``sonar_wcfg_bands_*``).  ``b`` None: v = a.  ``out`` may be ``x`` (the second launch of a cond / uncond rule).  None when the
    kernel does not take the shape / wavelet (callers use ``wcfg_fused``)."""
    B, Cc, H, W = a.shape
    flat = [float(v) for row in yh_scales for v in row]
    if len(flat) != 3 * levels or not (len(dec_lo) == len(dec_hi) == len(rec_lo) == len(rec_hi)):
        raise SonarHipError("wcfg_bands: scale table must be [levels][3], the four filters of one length")
    out = torch.empty_like(a) if out is None else out
    fn = load().sonar_wcfg_bands_f64 if high_precision else load().sonar_wcfg_bands_f32
    rc = fn(_dev(a, "a"), _opt(b, "b"), _opt(x, "x"), _dev(out, "out"), B * Cc, H, W, int(levels), _taps_arr(dec_lo), _taps_arr(dec_hi),
            _taps_arr(rec_lo), _taps_arr(rec_hi), len(dec_lo), DWT_MODE_IDS[mode], DWT_MODE_IDS[inv_mode], _darr(flat), float(yl_scale), float(ku),
            float(kt), int(bool(subtract_from_x)), _stream())
    if rc == ERR_UNSUPPORTED:
        return None
    _check(rc, "sonar_wcfg_bands")
    return out


def _taps_arr(vals):
    """ctypes double array of a filter-tap list, built once per distinct list (the taps of a Wavelet never change)."""
    key = tuple(vals)
    hit = _TAPS.get(key)
    if hit is None:
        if len(_TAPS) > 256:
            _TAPS.clear()
        hit = _TAPS[key] = _darr(key)
    return hit


def axis_taps(x: torch.Tensor, idx: torch.Tensor, coef: torch.Tensor, axis: int, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """out[..., j, ...] (+)= sum_k coef[j, k] * x[..., idx[j, k], ...] along ``axis`` (-2 or -1) of a contiguous fp32 / fp64 tensor."""
    kind = _wavelet_dtype(x)
    if axis not in (-1, -2) or x.ndim < 2 or not x.is_contiguous() or coef.dtype != x.dtype or idx.dtype != torch.int32:
        raise SonarHipError("axis_taps: a contiguous tensor, axis -1 / -2 and tables of matching type are required")
    n_out, taps = idx.shape
    n_in = x.shape[axis]
    inner = x.shape[-1] if axis == -2 else 1
    outer = x.numel() // (n_in * inner)
    shape = list(x.shape)
    shape[axis] = n_out
    if out is None:
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    elif list(out.shape) != shape or not out.is_contiguous() or out.dtype != x.dtype:
        raise SonarHipError("axis_taps: output shape mismatch")
    fn = load().sonar_axis_taps_f32 if kind == "f32" else load().sonar_axis_taps_f64
    _check(fn(_dev(x, "x", x.dtype), _dev(out, "out", x.dtype), outer, n_in, n_out, inner, idx.data_ptr(), _dev(coef, "coef", x.dtype), taps,
              int(bool(accumulate)), _stream()), f"sonar_axis_taps_{kind}")
    return out


def dtcwt_q2c(lh: torch.Tensor, hh: torch.Tensor, hl: torch.Tensor) -> torch.Tensor:
    """Three [B, C, 2h, 2w] planes -> [B, C, 6, h, w, 2] complex bands (orientation order 15 .. 165 degrees)."""
    kind = _wavelet_dtype(lh)
    B, Cc, H2, W2 = lh.shape
    out = torch.empty((B, Cc, 6, H2 // 2, W2 // 2, 2), dtype=lh.dtype, device=lh.device)
    fn = load().sonar_dtcwt_q2c_f32 if kind == "f32" else load().sonar_dtcwt_q2c_f64
    _check(fn(_dev(lh, "lh", lh.dtype), _dev(hh, "hh", lh.dtype), _dev(hl, "hl", lh.dtype), _dev(out, "bands", lh.dtype), B * Cc, H2 // 2, W2 // 2,
              _stream()), f"sonar_dtcwt_q2c_{kind}")
    return out


def dtcwt_c2q(bands: torch.Tensor):
    """[B, C, 6, h, w, 2] -> (lh, hh, hl), each [B, C, 2h, 2w]."""
    kind = _wavelet_dtype(bands)
    B, Cc, six, h, w, two = bands.shape
    if six != 6 or two != 2:
        raise SonarHipError("dtcwt_c2q: bands must be [B, C, 6, h, w, 2]")
    bands = bands.contiguous()
    outs = [torch.empty((B, Cc, 2 * h, 2 * w), dtype=bands.dtype, device=bands.device) for _ in range(3)]
    fn = load().sonar_dtcwt_c2q_f32 if kind == "f32" else load().sonar_dtcwt_c2q_f64
    _check(fn(_dev(bands, "bands", bands.dtype), *[_dev(o, "plane", bands.dtype) for o in outs], B * Cc, h, w, _stream()), f"sonar_dtcwt_c2q_{kind}")
    return tuple(outs)


def minmax_rescale(x: torch.Tensor, rows: int, inner: int, lo: torch.Tensor, hi: torch.Tensor, eps: float, target_min: float,
                   target_max: float) -> torch.Tensor:
    out = torch.empty_like(x)
    _check(load().sonar_minmax_rescale_f32(_dev(x, "x"), rows, inner, _dev(lo, "lo"), _dev(hi, "hi"), float(eps), float(target_min),
                                           float(target_max), _dev(out, "out"), _stream()), "sonar_minmax_rescale_f32")
    return out


def signed_rescale(x: torch.Tensor, rows: int, inner: int, min_neg: float, max_neg: float, min_pos: float, max_pos: float, eps: float = 1e-07) -> torch.Tensor:
    """normalize_to_scale_adv per row of ``inner`` elements (``sonar_signed_rescale_f32``)."""
    out = torch.empty_like(x)
    ws = torch.empty(max(rows, 1) * 4, dtype=torch.float32, device=x.device)
    _check(load().sonar_signed_rescale_f32(_dev(x, "x"), rows, inner, float(min_neg), float(max_neg), float(min_pos), float(max_pos), float(eps),
                                           _dev(ws, "stats_ws"), _dev(out, "out"), _stream()), "sonar_signed_rescale_f32")
    return out


def max_to_host(x: torch.Tensor) -> float:
    """``x.max().item()`` for a non-empty fp32 device tensor (NaN if any element is NaN) in one launch and one stream wait."""
    x = x.contiguous()
    slot = C.c_float()  # per call: the sampler thread and a preview thread may both be in here
    _check(load().sonar_max_to_host_f32(_dev(x, "x"), x.numel(), C.byref(slot), _stream()), "sonar_max_to_host_f32")
    return slot.value


def max_to_host_begin(x: torch.Tensor):
    """Launch half of ``max_to_host``: returns the token ``max_to_host_end`` takes.  The caller does whatever does not need the value
    in between (one request per thread at a time; always collect it, also on an error path)."""
    x = x.contiguous()
    stream = _stream()
    _check(load().sonar_max_to_host_begin_f32(_dev(x, "x"), x.numel(), stream), "sonar_max_to_host_begin_f32")
    return (stream, x)  # x: keeps the vector alive until the value has been read


def max_to_host_end(token) -> float:
    slot = C.c_float()
    _check(load().sonar_max_to_host_end_f32(C.byref(slot), token[0]), "sonar_max_to_host_end_f32")
    return slot.value


class FusedCall:
    """``sonar_wcfg_fused_*`` with everything but the tensors converted once: the taps, scale tables and scalars of a rule that is not
    scheduled are the same numbers at every step (WaveletCFG keeps one per rule), a step supplies cond / uncond / x."""

    def __init__(self, *, levels: int, dec_lo, dec_hi, mode: str, rec_lo, rec_hi, inv_mode: str, yl_scales, yh_scales, blend_mode: str,
                 strength: float, subtract_from_x: bool, high_precision: bool, perfect_reconstruction: bool = False):
        flat = [float(v) for lvl in yh_scales for name in lvl for v in name]
        if len(flat) != levels * 12 or len(yl_scales) != 4:
            raise SonarHipError("wcfg_fused: scale tables must be [levels][4][3] and [4]")
        lib = load()
        self.levels, self.elem = int(levels), 8 if high_precision else 4
        self.dec_len, self.rec_len, self.mode, self.inv_mode = len(dec_lo), len(rec_lo), DWT_MODE_IDS[mode], DWT_MODE_IDS[inv_mode]
        self.fn = lib.sonar_wcfg_fused_f64 if high_precision else lib.sonar_wcfg_fused_f32
        self.mid = (self.levels, _taps_arr(dec_lo), _taps_arr(dec_hi), self.dec_len, self.mode, _taps_arr(rec_lo), _taps_arr(rec_hi), self.rec_len,
                    self.inv_mode, _darr([float(v) for v in yl_scales]), _darr(flat), BLEND_IDS[blend_mode], float(strength), int(bool(subtract_from_x)),
                    int(bool(perfect_reconstruction)))

    def __call__(self, cond: torch.Tensor, uncond: torch.Tensor, x: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        B, Cc, H, W = cond.shape
        planes = B * Cc
        nkey = (planes, H, W, self.levels, self.dec_len, self.mode, self.rec_len, self.inv_mode, self.elem)
        need = _WCFG_NEED.get(nkey)
        if need is None:
            need = _WCFG_NEED[nkey] = load().sonar_wcfg_fused_ws_bytes(planes, H, W, self.levels, self.dec_len, self.mode, self.rec_len, self.inv_mode,
                                                                       self.elem)
        if need < 0:
            return None
        stream = _stream()
        key = (cond.device, stream)
        ws = _WCFG_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _WCFG_WS[key] = torch.empty(max(need, 1), dtype=torch.uint8, device=cond.device)  # reused across steps of a sampling run
        out = torch.empty_like(cond)
        rc = self.fn(_dev(cond, "cond"), _dev(uncond, "uncond"), _opt(x, "x"), _dev(out, "out"), planes, H, W, *self.mid, ws.data_ptr(), ws.numel(),
                     stream)
        if rc == ERR_UNSUPPORTED:
            return None
        _check(rc, "sonar_wcfg_fused")
        return out


def wcfg_fused(cond: torch.Tensor, uncond: torch.Tensor, x: Optional[torch.Tensor], **params):
    """WaveletCFG's transform-domain step for fp32 [B, C, H, W] cond / uncond (and x) in three launches (level-1 analysis, the deeper
    levels, level-1 synthesis); returns the fp32 output, or None when a level does not fit the LDS tile (caller uses the per-pass
    kernels).  ``perfect_reconstruction``: the analysis / synthesis pair is one wavelet both ways, which lets difference-only rules
    transform cond - uncond alone and every rule run its deeper levels with the coefficients resident in LDS.  Parameters: ``FusedCall``."""
    return FusedCall(**params)(cond, uncond, x)


# ------------------------------------------------------------------------------------------------ prepared call plans
# A sampler step whose launches depend on nothing but the RNG position and fresh output tensors (generate-mode Gaussian / uniform / Perlin
# / pyramid / power-law items and chains of them) is traced ONCE -- the entry points it calls, in order, with their arguments -- and from
# then on issued by one foreign call (include/sonar_hip.h, "prepared call plans"; csrc/plan.hip).  The replay calls the same entry points
# with the same arguments, so the output bits are those of the ordinary path; what disappears is the interpreter work per step (~50 us
# for a two-item chain at any batch size, against 20-35 us of kernels at batch 64).
PLANS_ENABLED = os.environ.get("SONAR_PLANS", "1") != "0"
PLAN_WARM_CALLS = 2     # ordinary calls before a step is traced (first-call setup, look-ahead misses)
PLAN_MAX_ATTEMPTS = 3   # traces that may fail (a call that took a fallback route) before the step stays on the ordinary path
FILL_AHEAD = os.environ.get("SONAR_FILL_AHEAD", "1") != "0"  # plans run a normalised uniform / Gaussian fill's statistics a call ahead (_FillAheadHook)
PYRAMID_AHEAD = os.environ.get("SONAR_PYRAMID_AHEAD", "1") != "0"  # plans run a normalised pyramid call's statistics a call ahead, one launch per call
PERLIN_AHEAD = os.environ.get("SONAR_PERLIN_AHEAD", "1") != "0"  # plans fuse a normalised Perlin call's three launches (_PerlinAheadHook)
NOT_RUN = object()      # Plan.run: the step was not issued (a guard changed, an entry point refused): take the ordinary path
_M64 = 2**64 - 1
_HOST_QUERIES = frozenset(("sonar_abi_version", "sonar_noise_stream_version", "sonar_last_error", "sonar_power_noise_ahead_ok", "sonar_perlin_noise_ahead_ok", "sonar_philox_noise_ahead_ok", "sonar_power_pipeline", "sonar_wcfg_hi_storage", "sonar_power_plane_kind", "sonar_dwt_out_len",
                           "sonar_dwt2_ws_bytes", "sonar_wcfg_lowpass_lds_bytes", "sonar_wcfg_fused_ws_bytes", "sonar_pyramid_levels",
                           "sonar_plan_fn_id", "sonar_plan_fn_nargs"))
PATCH_SLOT, PATCH_STREAM, PATCH_SEED, PATCH_BLOB, PATCH_LEVELS = range(5)


class PlanPatch(C.Structure):
    """``sonar_plan_patch`` (include/sonar_hip.h)."""

    _fields_ = [("source", C.c_int32), ("target", C.c_int32), ("index", C.c_int32), ("width", C.c_int32), ("addend", C.c_int64)]


class PlanLevels(C.Structure):
    """``sonar_plan_levels`` (include/sonar_hip.h)."""

    _fields_ = [("H", C.c_int64), ("W", C.c_int64), ("discount", C.c_double), ("iterations", C.c_int32), ("reserved", C.c_int32),
                ("h_offset", C.c_int64), ("w_offset", C.c_int64), ("weight_offset", C.c_int64)]


class AutoLevels(list):
    """The level list of a device-mode pyramid draw together with the rule it came from (``sonar_pyramid_levels``: sizes and weights are a
    function of (seed, stream)), so that a plan can recompute the table per call instead of freezing one call's sizes."""

    def __init__(self, h: int, w: int, iterations: int, discount: float, seed: int, stream: int):
        hs, ws, wts = (C.c_int64 * max(iterations, 1))(), (C.c_int64 * max(iterations, 1))(), (C.c_float * max(iterations, 1))()
        n = load().sonar_pyramid_levels(h, w, iterations, float(discount), seed & _M64, stream & _M64, hs, ws, wts)
        if n < 0:
            _check(n, "sonar_pyramid_levels")
        super().__init__((None, int(hs[i]), int(ws[i]), float(wts[i])) for i in range(n))
        self.rule = (int(h), int(w), int(iterations), float(discount), stream & _M64)


class PlanError(Exception):
    """The traced call cannot be replayed (an entry point outside the replayable set, an address nobody accounts for, ...)."""


class _RecordingLib:
    """Stands in for the library while a call is traced: every entry point is called as usual and noted with its arguments."""

    def __init__(self, lib, rec):
        self._lib, self._rec = lib, rec

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        rec = self._rec

        def call(*args):
            rc = fn(*args)
            rec.on_call(name, args, rc)
            return rc

        setattr(self, name, call)
        return call


class _Recorder:
    def __init__(self, lib):
        self.lib = _RecordingLib(lib, self)
        self.thread = _threading.get_ident()  # only this thread's calls and allocations belong to the trace
        self.calls = []       # (name, args)
        self.seen = {}        # address -> tensor handed to a kernel through _dev (constants the plan keeps alive)
        self.temps = []       # tensors allocated while the call ran
        self.seed = None
        self.base = None      # first RNG stream id the call took
        self.count = 0        # stream ids taken
        self.hooks = []       # PlanHook objects registered by wrappers whose arguments follow host state (power look-ahead)
        self.failed = None

    def fail(self, why: str):
        if self.failed is None:
            self.failed = why

    def on_call(self, name, args, rc):
        if name in _HOST_QUERIES:
            return
        if load_raw().sonar_plan_fn_id(name.encode()) < 0:
            return self.fail(f"{name} is not replayable")
        if rc != 0:
            return self.fail(f"{name} returned {rc} during the trace")
        self.calls.append((name, args))

    def on_take(self, seed: int, stream: int, count: int):
        seed &= _M64
        if self.base is None:
            self.seed, self.base = seed, stream
        elif seed != self.seed or stream != self.base + self.count:
            self.fail("the RNG position moved during the call")
        self.count += count

    def on_alloc(self, t):
        if isinstance(t, torch.Tensor) and t.is_cuda and self.thread == _threading.get_ident():
            self.temps.append(t)
        return t


def load_raw() -> C.CDLL:
    return _lib if _lib is not None else load()


class PlanHook:
    """A wrapper whose arguments follow host state between calls (the power-law look-ahead) takes part in a plan through this protocol:
    ``managed`` maps the addresses it supplies per call to names, ``bind`` learns their slot numbers, ``pre_run`` decides whether the
    recorded call is still the right one and fills its slots, ``post_run`` updates the host state as the ordinary path would."""

    managed: dict = {}

    def bind(self, slot_of: dict):
        raise NotImplementedError

    def pre_run(self, seed: int, base: int, slots, stream: int) -> bool:
        raise NotImplementedError

    def post_run(self, seed: int, base: int):
        pass


class _PowerAheadHook(PlanHook):
    """``power_noise(lookahead=...)`` inside a plan: the statistics input is what the previous call left (``lookahead.partials``), the
    statistics output alternates between two tensors of the plan's own, and the call is only replayed while the sampler's stream ids
    advance by the settled step -- anything else is the ordinary path's business, which shares the ``PowerLookahead`` state."""

    def __init__(self, la, key_for, ws, nws, rel: int, step: int, device):
        self.la, self.key_for, self.rel, self.step, self.device = la, key_for, rel, step, device
        self.managed = {ws.data_ptr(): "ws", nws.data_ptr(): "nws"}
        self.pair = None
        self.flip = 0
        self.now = None

    def bind(self, slot_of):
        self.ws_slot, self.nws_slot = slot_of["ws"], slot_of["nws"]

    def pre_run(self, seed, base, table, st):
        la = self.la
        stream = base + self.rel
        if (la.key is None or la.step != self.step or la.last_stream is None or stream - la.last_stream != self.step
                or la.key != self.key_for(stream, seed, st)):
            return False
        if self.pair is None:
            self.pair = (new_partials(self.device), new_partials(self.device))
        nws = self.pair[self.flip]
        if nws is la.partials:
            self.flip ^= 1
            nws = self.pair[self.flip]
        table[self.ws_slot] = la.partials.data_ptr()
        table[self.nws_slot] = nws.data_ptr()
        self.now = (stream, seed, st, nws)
        return True

    def post_run(self, seed, base):
        stream, seed, st, nws = self.now
        la = self.la
        la.hits += 1
        la.key, la.partials, la.last_stream, la.last_delta = self.key_for((stream + self.step) & _M64, seed, st), nws, stream, self.step
        self.flip ^= 1


class _PerlinAheadHook(PlanHook):
    """A normalised Perlin call inside a plan, ONE launch per call in the steady state (``sonar_perlin_noise_ahead_f32``): the plan knows
    the stream ids of the calls that follow (this call's + the streams a call takes), so a call's launch also runs the statistics pass
    of the next call and the lattice of the call after it.  This hook keeps what earlier launches left -- three lattice buffers, two
    statistics buffers per HIP stream, keyed by (seed, stream id) -- and supplies whatever is missing the ordinary way (a lattice launch;
    ``have_stats`` = 0 makes the entry point launch the statistics pass): a reseed or a draw by somebody else costs one call its
    shortcuts, never its values."""

    KEYS = ("t_now", "p_now", "have", "t_next", "p_next", "t_out", "l_out")

    def __init__(self, sa: int, la: int, count: int, lattice, device):
        self.sa, self.la, self.count, self.lattice, self.device = sa, la, count, lattice, device  # lattice: (iters, C, H, W, blend)
        self.managed = {}
        self.by_stream = {}
        self.now = None
        self.hits = self.misses = 0

    def bind(self, slot_of):
        self.slot = {k: slot_of["perlin_" + k] for k in self.KEYS}

    def pre_run(self, seed, base, table, st):
        state = self.by_stream.get(st)
        if state is None:
            _it, c, h, w, _bl = self.lattice
            state = self.by_stream[st] = {"terms": [torch.empty((1, c, h, w), dtype=torch.float32, device=self.device) for _ in range(3)],
                                          "parts": [new_partials(self.device) for _ in range(2)], "ready_terms": {}, "ready_parts": {}}
            state["tptr"] = [t.data_ptr() for t in state["terms"]]  # (the addresses never change: one foreign call each, once)
            state["pptr"] = [t.data_ptr() for t in state["parts"]]
        terms, tptr, pptr = state["terms"], state["tptr"], state["pptr"]
        l_now, s_now = (base + self.la) & _M64, (base + self.sa) & _M64
        l_next, s_next, l_next2 = (l_now + self.count) & _M64, (s_now + self.count) & _M64, (l_now + 2 * self.count) & _M64
        ti = state["ready_terms"].get((seed, l_now))
        tn = state["ready_terms"].get((seed, l_next))
        it, c, h, w, bl = self.lattice
        if ti is None or tn is None:
            # nobody left this call's lattice, or the next call's (the first run; after a reseed or somebody else's draw): the ordinary
            # launch for whichever is missing, so that from the next call on every launch finds both its inputs
            self.misses += 1
            if ti is None:
                ti = next(i for i in range(3) if i != tn)
                _check(_lib.sonar_perlin_lattice_f32(terms[ti].data_ptr(), it, c, h, w, bl, seed, l_now, st), "sonar_perlin_lattice_f32")
            if tn is None:
                tn = next(i for i in range(3) if i != ti)
                _check(_lib.sonar_perlin_lattice_f32(terms[tn].data_ptr(), it, c, h, w, bl, seed, l_next, st), "sonar_perlin_lattice_f32")
        else:
            self.hits += 1
        pi = state["ready_parts"].get((seed, s_now))
        have = pi is not None
        if not have:
            pi = 0
        pn = 1 - pi
        to = 3 - ti - tn  # the third of the buffers 0, 1, 2
        target = l_next2
        sl = self.slot
        table[sl["t_now"]] = tptr[ti]
        table[sl["p_now"]] = pptr[pi]
        table[sl["have"]] = int(have)
        table[sl["t_next"]] = tptr[tn]
        table[sl["p_next"]] = pptr[pn]
        table[sl["t_out"]] = tptr[to]
        table[sl["l_out"]] = target
        self.now = (state, {(seed, target): to, (seed, l_next): tn}, {(seed, s_next): pn})
        # the launch overwrites a lattice buffer and a statistics buffer: until post_run says what they hold, nothing is "ready" -- a run
        # that fails in between (an entry point refusing the call) must not leave keys pointing at overwritten buffers
        state["ready_terms"], state["ready_parts"] = {}, {}
        return True

    def post_run(self, seed, base):
        state, terms_ready, parts_ready = self.now
        state["ready_terms"], state["ready_parts"] = terms_ready, parts_ready


def _peephole_perlin_ahead(records, b, rec):
    """[sonar_perlin_lattice_f32 -> terms] [sonar_perlin_noise_f32(terms, ...)] of a traced step become one sonar_perlin_noise_ahead_f32
    record driven by a ``_PerlinAheadHook`` -- where the entry point takes the shape (launch-bound sizes, whole tiles)."""
    lib = load_raw()
    out = []
    i = 0
    while i < len(records):
        name, words, blob, patches = records[i]
        nxt = records[i + 1] if i + 1 < len(records) else None
        done = False
        if name == "sonar_perlin_lattice_f32" and nxt is not None and nxt[0] == "sonar_perlin_noise_f32" and not blob and not nxt[2]:
            lp = {pt.target: pt for pt in patches}
            np_ = {pt.target: pt for pt in nxt[3]}
            nw = nxt[1]
            same_terms = (0 in lp and 0 in np_ and lp[0].source == PATCH_SLOT and np_[0].source == PATCH_SLOT and lp[0].index == np_[0].index
                          and lp[0].addend == 0 and np_[0].addend == 0)
            signed = lambda v: v - (1 << 64) if v >> 63 else v  # noqa: E731
            B, chw, iters, offs = signed(nw[2]), signed(nw[3]), signed(nw[4]), signed(nw[8])
            lat = tuple(signed(words[k]) for k in (1, 2, 3, 4, 5))  # iters, C, H, W, blend
            if (same_terms and iters == 1 and lat[1] * lat[2] * lat[3] == chw and lib.sonar_perlin_noise_ahead_ok(B, chw, offs)
                    and all(k in np_ for k in (1, 6, 7, 11)) and 7 in lp and lp[7].source == PATCH_STREAM and np_[7].source == PATCH_STREAM):
                hook = _PerlinAheadHook(int(np_[7].addend), int(lp[7].addend), rec.count, lat, next(iter(b.temp_ranges))[2].device)
                slots = {}
                for key in hook.KEYS:
                    slots[key] = b.slot_of[("hook", "perlin_" + key)] = len(b.slots)
                    b.slots.append(None)
                w2 = [0, 0, nw[2], nw[3], nw[5], 0, 0, nw[8], nw[9], nw[10], 0, 0, 0, 0, 0, 0, words[1], words[2], words[3], words[4], words[5], 0, 0]
                p2 = [PlanPatch(PATCH_SLOT, 0, slots["t_now"], 8, 0), PlanPatch(PATCH_SLOT, 1, np_[1].index, 8, np_[1].addend),
                      PlanPatch(PATCH_SEED, 5, 0, 8, 0), PlanPatch(PATCH_STREAM, 6, 0, 8, np_[7].addend),
                      PlanPatch(PATCH_SLOT, 10, slots["p_now"], 8, 0), PlanPatch(PATCH_SLOT, 11, slots["have"], 8, 0),
                      PlanPatch(PATCH_STREAM, 12, 0, 8, np_[7].addend + rec.count), PlanPatch(PATCH_SLOT, 13, slots["t_next"], 8, 0),
                      PlanPatch(PATCH_SLOT, 14, slots["p_next"], 8, 0), PlanPatch(PATCH_SLOT, 15, slots["t_out"], 8, 0),
                      PlanPatch(PATCH_SLOT, 21, slots["l_out"], 8, 0)]
                out.append(("sonar_perlin_noise_ahead_f32", w2, b"", p2))
                rec.hooks.append(hook)
                i += 2
                done = True
        if not done:
            out.append(records[i])
            i += 1
    return out


class _FillAheadHook(PlanHook):
    """A normalised uniform / Gaussian fill (``sonar_philox_noise_ahead_f32``) or pyramid call (``sonar_pyramid_noise_ahead_f32``) inside a
    plan, round 6: the plan knows the stream ids of the call that follows, so this call's launch also computes that call's statistics
    -- the fill in the same waves, behind the stores of the final pass; the pyramid in workgroups of their own.  Two statistics buffers
    per HIP stream, keyed by (seed, stream id); a call that finds nothing (the first, after a reseed, somebody else's draw or a call
    the entry point refused) gets ``have_stats`` = 0 and the entry point computes the statistics first: it loses its shortcut, never
    its values."""

    KEYS = ("p_now", "have", "p_next")

    def __init__(self, tag: str, sa: int, count: int, device):
        self.tag, self.sa, self.count, self.device = tag, sa, count, device
        self.managed = {}
        self.by_stream = {}
        self.now = None
        self.hits = self.misses = 0

    def bind(self, slot_of):
        self.slot = {k: slot_of[self.tag + k] for k in self.KEYS}

    def pre_run(self, seed, base, table, st):
        state = self.by_stream.get(st)
        if state is None:
            state = self.by_stream[st] = {"parts": [new_partials(self.device) for _ in range(2)], "ready": {}}
        s_now = (base + self.sa) & _M64
        s_next = (s_now + self.count) & _M64
        pi = state["ready"].get((seed, s_now))
        have = pi is not None
        if have:
            self.hits += 1
        else:
            self.misses += 1
            pi = 0
        sl = self.slot
        table[sl["p_now"]] = state["parts"][pi].data_ptr()
        table[sl["have"]] = int(have)
        table[sl["p_next"]] = state["parts"][1 - pi].data_ptr()
        self.now = (state, {(seed, s_next): 1 - pi})
        state["ready"] = {}  # (as in _PerlinAheadHook: valid again in post_run)
        return True

    def post_run(self, seed, base):
        self.now[0]["ready"] = self.now[1]


def _peephole_fill_ahead(records, b, rec):
    """Every [sonar_philox_noise_f32] record of a traced step whose shape has a statistics pass (uniform draws, or factor != 1) becomes a
    sonar_philox_noise_ahead_f32 record driven by a ``_FillAheadHook``."""
    lib = load_raw()
    out = []
    signed = lambda v: v - (1 << 64) if v >> 63 else v  # noqa: E731
    for k, (name, words, blob, patches) in enumerate(records):
        if name == "sonar_philox_noise_f32" and not blob:
            pt = {p.target: p for p in patches}
            factor = C.c_float.from_buffer_copy(int(words[9] & 0xFFFFFFFF).to_bytes(4, "little")).value
            if (all(t in pt for t in (1, 3, 4, 11)) and pt[3].source == PATCH_SEED and pt[4].source == PATCH_STREAM
                    and lib.sonar_philox_noise_ahead_ok(int(signed(words[0])), signed(words[2]), factor)):
                tag = f"fill{k}_"
                hook = _FillAheadHook(tag, int(pt[4].addend), rec.count, next(iter(b.temp_ranges))[2].device)
                slots = {}
                for key in hook.KEYS:
                    slots[key] = b.slot_of[("hook", tag + key)] = len(b.slots)
                    b.slots.append(None)
                w2 = list(words[:11]) + [0, 0, 0, 0, 0]
                p2 = [pt[1], pt[3], pt[4], PlanPatch(PATCH_SLOT, 11, slots["p_now"], 8, 0), PlanPatch(PATCH_SLOT, 12, slots["have"], 8, 0),
                      PlanPatch(PATCH_STREAM, 13, 0, 8, pt[4].addend + rec.count), PlanPatch(PATCH_SLOT, 14, slots["p_next"], 8, 0)]
                out.append(("sonar_philox_noise_ahead_f32", w2, b"", p2))
                rec.hooks.append(hook)
                continue
        out.append((name, words, blob, patches))
    return out


def _peephole_pyramid_ahead(records, b, rec):
    """A [sonar_pyramid_noise_f32] record of a traced step whose levels are all drawn in the kernel (a level rule: PATCH_LEVELS) with the
    bilinear mode becomes a sonar_pyramid_noise_ahead_f32 record: a second level rule -- the next call's, its stream ids ``rec.count``
    further on -- joins the blob, a ``_FillAheadHook`` supplies the two statistics buffers."""
    out = []
    for k, (name, words, blob, patches) in enumerate(records):
        done = False
        if name == "sonar_pyramid_noise_f32" and blob:
            pt = {p.target: p for p in patches}
            lev = pt.get(4)
            if (lev is not None and lev.source == PATCH_LEVELS and all(t in pt for t in (0, 6, 7, 8, 10, 11, 15)) and pt[10].source == PATCH_SEED
                    and pt[11].source == PATCH_STREAM and all(pt[t].source == PATCH_BLOB for t in (6, 7, 8)) and int(words[9]) == 0):
                rule = PlanLevels.from_buffer_copy(bytes(blob[lev.index:lev.index + C.sizeof(PlanLevels)]))
                n = max(int(rule.iterations), 1)
                blob2 = bytearray(blob)
                h_off = _blob_reserve(blob2, 8 * n, bytes(8 * n))
                w_off = _blob_reserve(blob2, 8 * n, bytes(8 * n))
                wt_off = _blob_reserve(blob2, 4 * n, bytes(4 * n))
                rule_off = _blob_reserve(blob2, C.sizeof(PlanLevels),
                                         bytes(PlanLevels(rule.H, rule.W, rule.discount, rule.iterations, 0, h_off, w_off, wt_off)))
                tag = f"pyr{k}_"
                hook = _FillAheadHook(tag, int(pt[11].addend), rec.count, next(iter(b.temp_ranges))[2].device)
                slots = {}
                for key in hook.KEYS:
                    slots[key] = b.slot_of[("hook", tag + key)] = len(b.slots)
                    b.slots.append(None)
                w2 = [0, words[1], words[2], words[3], 0, 0, 0, 0, words[9], 0, 0, words[12], words[13], words[14], 0, 0, 0, 0, 0, 0, 0, 0, 0]
                p2 = [PlanPatch(PATCH_SLOT, 0, pt[0].index, 8, pt[0].addend), PlanPatch(PATCH_LEVELS, 4, lev.index, 8, lev.addend),
                      PlanPatch(PATCH_BLOB, 5, 0, 8, pt[6].addend), PlanPatch(PATCH_BLOB, 6, 0, 8, pt[7].addend), PlanPatch(PATCH_BLOB, 7, 0, 8, pt[8].addend),
                      PlanPatch(PATCH_SEED, 9, 0, 8, 0), PlanPatch(PATCH_STREAM, 10, 0, 8, pt[11].addend),
                      PlanPatch(PATCH_SLOT, 14, slots["p_now"], 8, 0), PlanPatch(PATCH_SLOT, 15, slots["have"], 8, 0),
                      PlanPatch(PATCH_STREAM, 16, 0, 8, pt[11].addend + rec.count), PlanPatch(PATCH_LEVELS, 17, rule_off, 8, lev.addend + rec.count),
                      PlanPatch(PATCH_BLOB, 18, 0, 8, h_off), PlanPatch(PATCH_BLOB, 19, 0, 8, w_off), PlanPatch(PATCH_BLOB, 20, 0, 8, wt_off),
                      PlanPatch(PATCH_SLOT, 21, slots["p_next"], 8, 0)]
                out.append(("sonar_pyramid_noise_ahead_f32", w2, bytes(blob2), p2))
                rec.hooks.append(hook)
                done = True
        if not done:
            out.append((name, words, blob, patches))
    return out


class _LatticeAheadHook(PlanHook):
    """A chain whose Perlin item is hosted by the pyramid kernel, inside a plan: the lattice launch leaves the call's critical path --
    the plane kernel's launch of call n computes the lattice of call n + 1 in extra workgroups (``sonar_pyramid_generate_acc_ahead_f32``).
    Two buffers per HIP stream, keyed by (seed, stream id); a call that finds nothing (the first, or after a reseed) launches its lattice
    the ordinary way."""

    KEYS = ("t_now", "t_out", "l_out")

    def __init__(self, la: int, count: int, lattice, device):
        self.la, self.count, self.lattice, self.device = la, count, lattice, device  # lattice: (iters, C, H, W, blend)
        self.managed = {}
        self.by_stream = {}
        self.now = None
        self.hits = self.misses = 0

    def bind(self, slot_of):
        self.slot = {k: slot_of["lattice_" + k] for k in self.KEYS}

    def pre_run(self, seed, base, table, st):
        it, c, h, w, bl = self.lattice
        state = self.by_stream.get(st)
        if state is None:
            state = self.by_stream[st] = {"terms": [torch.empty((1, c, h, w), dtype=torch.float32, device=self.device) for _ in range(2)], "ready": {}}
        terms = state["terms"]
        l_now = (base + self.la) & _M64
        l_next = (l_now + self.count) & _M64
        ti = state["ready"].get((seed, l_now))
        if ti is None:
            ti = 0
            _check(_lib.sonar_perlin_lattice_f32(terms[ti].data_ptr(), it, c, h, w, bl, seed, l_now, st), "sonar_perlin_lattice_f32")
            self.misses += 1
        else:
            self.hits += 1
        sl = self.slot
        table[sl["t_now"]] = terms[ti].data_ptr()
        table[sl["t_out"]] = terms[1 - ti].data_ptr()
        table[sl["l_out"]] = l_next
        self.now = (state, {(seed, l_next): 1 - ti})
        state["ready"] = {}  # (as in _PerlinAheadHook: valid again in post_run)
        return True

    def post_run(self, seed, base):
        self.now[0]["ready"] = self.now[1]


def _peephole_lattice_ahead(records, b, rec):
    """[sonar_perlin_lattice_f32 -> terms] ... [sonar_pyramid_generate_acc_f32(pre = Perlin prefix over those terms)] of a traced chain step:
    the lattice record goes, the plane kernel's record becomes sonar_pyramid_generate_acc_ahead_f32 driven by a ``_LatticeAheadHook``."""
    signed = lambda v: v - (1 << 64) if v >> 63 else v  # noqa: E731
    terms_off = FoldPrefixArg.terms.offset
    for i, (name, words, blob, patches) in enumerate(records):
        if name != "sonar_perlin_lattice_f32" or blob:
            continue
        lp = {pt.target: pt for pt in patches}
        if 0 not in lp or lp[0].source != PATCH_SLOT or 7 not in lp or lp[7].source != PATCH_STREAM:
            continue
        tslot = lp[0].index
        users = [j for j, r in enumerate(records) if j != i and any(pt.source == PATCH_SLOT and pt.index == tslot for pt in r[3])]
        if len(users) != 1 or records[users[0]][0] != "sonar_pyramid_generate_acc_f32" or users[0] < i:
            continue
        j = users[0]
        _n, w2, blob2, p2 = records[j]
        pre_arg = next((pt for pt in p2 if pt.source == PATCH_BLOB and pt.target == 1), None)
        if pre_arg is None:
            continue
        field = -(int(pre_arg.addend) + terms_off + 1)  # the blob patch that fills pre.terms
        tpatch = next((pt for pt in p2 if pt.source == PATCH_SLOT and pt.index == tslot and pt.target == field and pt.addend == 0), None)
        lat = tuple(signed(words[k]) for k in (1, 2, 3, 4, 5))  # iters, C, H, W, blend
        if tpatch is None or (signed(w2[3]), signed(w2[4])) != (lat[2], lat[3]):
            continue
        hook = _LatticeAheadHook(int(lp[7].addend), rec.count, lat, next(iter(b.temp_ranges))[2].device)
        slots = {}
        for key in hook.KEYS:
            slots[key] = b.slot_of[("hook", "lattice_" + key)] = len(b.slots)
            b.slots.append(None)
        new_patches = [PlanPatch(PATCH_SLOT, field, slots["t_now"], 8, 0) if pt is tpatch else pt for pt in p2]
        new_patches += [PlanPatch(PATCH_SLOT, 14, slots["t_out"], 8, 0), PlanPatch(PATCH_SLOT, 18, slots["l_out"], 8, 0)]
        new_words = list(w2[:14]) + [0, words[1], words[2], words[5], 0, 0]
        out = list(records)
        out[j] = ("sonar_pyramid_generate_acc_ahead_f32", new_words, blob2, new_patches)
        del out[i]
        rec.hooks.append(hook)
        return _peephole_lattice_ahead(out, b, rec)
    return records


def _float_word(v: float) -> int:
    return int.from_bytes(C.c_float(v), "little")  # the bit pattern of the rounded float


def _double_word(v: float) -> int:
    return int.from_bytes(C.c_double(v), "little")


class _PlanBuilder:
    """Turns a recorded call into a ``sonar_plan`` + the Python-side description of its slots and result."""

    def __init__(self, rec: _Recorder, result):
        self.rec = rec
        self.slots = []      # [tensor template] in slot order
        self.slot_of = {}    # id(tensor) or hook key -> slot index
        self.constants = []  # tensors the plan only reads (kept alive)
        self.managed = {}    # address -> hook key
        for h in rec.hooks:
            self.managed.update(h.managed)
        self.temp_ranges = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), t) for t in rec.temps if t.numel()]
        self.result = result
        self.scratch_refs = []  # weak references to the temporaries the plan will own copies of

    def resolve(self, addr: int):
        """(slot index, byte offset) for an address the plan supplies per call, or None for a constant the caller owns."""
        if addr in self.managed:
            key = ("hook", self.managed[addr])
            if key not in self.slot_of:
                self.slot_of[key] = len(self.slots)
                self.slots.append(None)
            return self.slot_of[key], 0
        for lo, hi, t in self.temp_ranges:
            if lo <= addr < hi:
                if id(t) not in self.slot_of:
                    self.slot_of[id(t)] = len(self.slots)
                    self.slots.append(t)
                return self.slot_of[id(t)], addr - lo
        t = self.rec.seen.get(addr)
        if t is None:
            for base, cand in self.rec.seen.items():
                if base <= addr < base + cand.numel() * cand.element_size():
                    t = cand
                    break
        if t is None:
            raise PlanError(f"address {addr:#x} belongs to no tensor the trace knows")
        self.constants.append(t)
        return None

    def stream_addend(self, value: int) -> int:
        if self.rec.base is None or value < self.rec.base or value - self.rec.base > self.rec.count + 64:
            raise PlanError("a stream id that does not follow from the call's RNG position")
        return value - self.rec.base

    def struct_blob(self, obj, blob: bytearray, patches: list) -> int:
        """Copy a by-reference struct into the blob; its addresses, seed and stream fields become patches.  Returns its offset."""
        off = _blob_reserve(blob, C.sizeof(obj), bytes(obj))
        for name, ftype in obj._fields_:
            fo = off + getattr(type(obj), name).offset
            val = getattr(obj, name)
            if ftype is C.c_void_p:
                if val:
                    where = self.resolve(int(val))
                    if where is not None:
                        patches.append(PlanPatch(PATCH_SLOT, -(fo + 1), where[0], 8, where[1]))
            elif ftype is C.c_uint64:
                if name == "seed":
                    if val != self.rec.seed:
                        raise PlanError("a seed that is not the call's RNG seed")
                    patches.append(PlanPatch(PATCH_SEED, -(fo + 1), 0, 8, 0))
                elif name == "stream_id":
                    patches.append(PlanPatch(PATCH_STREAM, -(fo + 1), 0, 8, self.stream_addend(val)))
                else:
                    raise PlanError(f"unknown 64-bit field {name}")
        return off

    def record(self, name: str, args):
        argtypes = SIGNATURES[name][1]
        if len(args) != len(argtypes):
            raise PlanError(f"{name}: argument count")
        words, blob, patches = [], bytearray(), []
        u64_seen = 0
        i = 0
        last = len(args) - 1
        for i, (at, val) in enumerate(zip(argtypes, args)):
            if i == last:  # the stream: patched by every run
                words.append(0)
            elif at is _U64:
                u64_seen += 1
                if u64_seen == 1:
                    if int(val) & _M64 != self.rec.seed:
                        raise PlanError(f"{name}: a seed that is not the call's RNG seed")
                    patches.append(PlanPatch(PATCH_SEED, i, 0, 8, 0))
                else:
                    patches.append(PlanPatch(PATCH_STREAM, i, 0, 8, self.stream_addend(int(val) & _M64)))
                words.append(0)
            elif at in (_I64, _I, C.c_int32):
                words.append(int(val) & _M64)
            elif at is _F:
                words.append(_float_word(val))
            elif at is _D:
                words.append(_double_word(val))
            elif val is None:
                words.append(0)
            elif isinstance(val, int):  # a device address
                where = self.resolve(val)
                if where is None:
                    words.append(val)
                else:
                    patches.append(PlanPatch(PATCH_SLOT, i, where[0], 8, where[1]))
                    words.append(0)
            elif isinstance(val, C.Array):
                rule = getattr(val, "_sonar_levels", None)
                if rule is not None:
                    self.levels_patch(rule, i, args, blob, patches)
                else:
                    if val._type_ is C.c_void_p and any(val):
                        raise PlanError(f"{name}: an array of addresses")
                    if getattr(val, "_sonar_levels_part", False):
                        words.append(0)  # ws / weights / level pointers: placed by the levels patch of this record
                        continue
                    patches.append(PlanPatch(PATCH_BLOB, i, 0, 8, _blob_reserve(blob, C.sizeof(val), bytes(val))))
                words.append(0)
            elif hasattr(val, "_obj") and isinstance(val._obj, C.Structure):
                patches.append(PlanPatch(PATCH_BLOB, i, 0, 8, self.struct_blob(val._obj, blob, patches)))
                words.append(0)
            else:
                raise PlanError(f"{name}: argument {i} ({type(val).__name__}) cannot be replayed")
        return name, words, bytes(blob), patches

    def levels_patch(self, rule, i: int, args, blob: bytearray, patches: list):
        """Argument i is the level-height table of a pyramid entry point (..., nlevels, level_ptrs, level_h, level_w, weight, ...)."""
        h, w, iterations, discount, stream = rule
        n = max(iterations, 1)
        ptr_off = _blob_reserve(blob, 8 * n, bytes(8 * n))
        h_off = _blob_reserve(blob, 8 * n, bytes(8 * n))
        w_off = _blob_reserve(blob, 8 * n, bytes(8 * n))
        wt_off = _blob_reserve(blob, 4 * n, bytes(4 * n))
        rule_off = _blob_reserve(blob, C.sizeof(PlanLevels), bytes(PlanLevels(h, w, discount, iterations, 0, h_off, w_off, wt_off)))
        for arg, off in ((i - 1, ptr_off), (i, h_off), (i + 1, w_off), (i + 2, wt_off)):
            patches.append(PlanPatch(PATCH_BLOB, arg, 0, 8, off))
        patches.append(PlanPatch(PATCH_LEVELS, i - 2, rule_off, 8, self.stream_addend(stream)))


def _blob_reserve(blob: bytearray, size: int, data: bytes) -> int:
    while len(blob) % 16:
        blob.append(0)
    off = len(blob)
    blob.extend(data[:size].ljust(size, b"\0"))
    return off


class Plan:
    """A recorded step (``sonar_plan``) with what the host must supply per call: fresh tensors for everything the result owns, the
    plan's own scratch tensors (one set per HIP stream), the RNG position, and the hooks' values."""

    def __init__(self, handle, nslots, fresh, scratch, result_spec, constants, rec: _Recorder, take, rewind, guards, device):
        self.handle, self.nslots = handle, nslots
        self.fresh = fresh            # [(slot, shape, dtype)] allocated per run
        self.scratch = scratch        # [(slot, shape, dtype)] allocated once per stream
        self.result_spec = result_spec
        self.constants = constants
        self.hooks = rec.hooks
        self.rng_count = rec.count
        self.take, self.rewind, self.guards, self.device = take, rewind, guards, device
        self.by_stream = {}           # hipStream_t -> (ctypes slot table, scratch tensors)
        self.failed = C.c_int(-1)
        self.runs = 0

    def __del__(self):
        if self.handle and _lib is not None:
            _lib.sonar_plan_destroy(self.handle)
            self.handle = None

    def _table(self, st: int):
        entry = self.by_stream.get(st)
        if entry is None:
            table = (C.c_uint64 * max(self.nslots, 1))()
            keep = []
            for slot, shape, dtype in self.scratch:
                t = torch.empty(shape, dtype=dtype, device=self.device)
                table[slot] = t.data_ptr()
                keep.append(t)
            entry = self.by_stream[st] = (table, keep)
        return entry[0]

    def run(self):
        for getter, want in self.guards:
            if getter() != want:
                return NOT_RUN
        cur = _CUR_DEVICE()  # (the raw query: torch.cuda.current_device() is a Python wrapper around it with a lazy-init check, ~0.6 us more)
        if cur != self.device.index:
            return NOT_RUN
        st = _RAW_STREAM(cur) if _RAW_STREAM is not None else torch.cuda.current_stream().cuda_stream
        table = self._table(st)
        seed, base = self.take(self.rng_count) if self.rng_count else (0, 0)
        seed &= _M64
        for h in self.hooks:
            if not h.pre_run(seed, base, table, st):
                if self.rng_count:
                    self.rewind(base, self.rng_count)
                return NOT_RUN
        fresh = []
        device = self.device
        for slot, shape, dtype in self.fresh:
            t = torch.empty(shape, dtype=dtype, device=device)
            table[slot] = t.data_ptr()
            fresh.append(t)
        failed = C.c_int(-1)
        rc = _lib.sonar_plan_run(self.handle, table, self.nslots, seed, base, st, failed)
        self.failed = failed
        if rc != 0:
            if self.rng_count:
                self.rewind(base, self.rng_count)
            if rc == ERR_UNSUPPORTED:
                return NOT_RUN  # an entry point refused this call's values (e.g. a level table the plane kernel cannot hold): ordinary path
            _check(rc, f"sonar_plan_run (record {self.failed.value})")
        for h in self.hooks:
            h.post_run(seed, base)
        self.runs += 1
        return _rebuild(self.result_spec, fresh)


def _rebuild(spec, fresh):
    kind = spec[0]
    if kind == "t":
        _k, idx, view, tag_idx = spec
        t = fresh[idx]
        if view is not None:
            t = t.view(view)
        if tag_idx is not None:
            setattr(t, STATS_ATTR, (fresh[tag_idx], t._version))
            tag_register(t)
        return t
    if kind == "n":
        return None
    return tuple(_rebuild(s, fresh) for s in spec[1])


# torch operations a traced step may run on device tensors: allocations (noted) and metadata-only views.  Anything else on a device
# tensor -- an arithmetic kernel of torch's own, a copy, an .item() -- is work the recorded entry points would not replay: no plan.
_TRACE_ALLOCS = frozenset(("empty", "empty_like", "empty_strided", "new_empty"))
_TRACE_VIEWS = frozenset(("view", "reshape", "_unsafe_view", "as_strided", "alias", "detach", "expand", "slice", "select", "t", "transpose", "permute",
                          "unsqueeze", "squeeze", "flatten", "unflatten", "narrow", "view_as", "_reshape_alias", "lift_fresh", "unbind", "split",
                          "is_same_size", "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "size", "stride", "numel", "dim",
                          "is_contiguous", "storage_offset", "data_ptr"))


def _tensors_in(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors_in(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors_in(o)


def _trace_mode(rec: _Recorder):
    from torch.utils._python_dispatch import TorchDispatchMode

    class TraceMode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
            if name in _TRACE_ALLOCS:
                for t in _tensors_in(out):
                    rec.on_alloc(t)
            elif name not in _TRACE_VIEWS and any(t.is_cuda for t in _tensors_in((args, kwargs, out))):
                rec.fail(f"torch operation aten.{name} on a device tensor inside the step")
            return out

    return TraceMode()


_trace_lock = _threading.Lock()


def trace_plan(fn, args, *, take, rewind, guards):
    """Run ``fn(*args)`` once on the ordinary path while recording it.  Returns (result, Plan or None): the result is the call's real
    result either way; the plan is None when the call cannot be replayed (``PlanError`` reasons are kept in ``trace_plan.last_reason``)."""
    global _recorder
    lib = load()
    if not _trace_lock.acquire(blocking=False):
        return fn(*args), None
    rec = _Recorder(lib)
    try:
        _recorder = rec
        try:
            with _trace_mode(rec):  # thread-local: sees this thread's torch operations only
                result = fn(*args)
        finally:
            _recorder = None
        plan = None
        try:
            if rec.failed is not None:
                raise PlanError(rec.failed)
            plan = _build_plan(rec, result, take, rewind, guards)
            trace_plan.last_reason = None
        except PlanError as exc:
            trace_plan.last_reason = str(exc)
        except Exception as exc:  # noqa: BLE001 -- the step itself ran and has its result: a builder bug must not become a failed sampler step
            trace_plan.last_reason = f"plan builder: {type(exc).__name__}: {exc}"
            plan = None
        if plan is not None:
            refs, plan.scratch_refs = plan.scratch_refs, None
            rec.temps.clear()
            rec.seen.clear()
            rec.calls.clear()
            if any(r() is not None for r in refs):  # somebody kept a tensor the call allocated (and did not return): not ours to replace
                trace_plan.last_reason = "a tensor allocated during the call outlives it"
                plan = None
        return result, plan
    finally:
        _trace_lock.release()


trace_plan.last_reason = None


def _build_plan(rec: _Recorder, result, take, rewind, guards) -> Plan:
    if not rec.calls:
        raise PlanError("the call launched nothing")
    b = _PlanBuilder(rec, result)
    records = [b.record(name, args) for name, args in rec.calls]
    # the tensors the entry points were handed are known now (constants are kept by the builder): let go of them, so that the only
    # holders of a temporary's storage left are this trace's own note of the allocation -- and whoever else kept it (checked below)
    rec.seen.clear()
    if PERLIN_AHEAD:
        records = _peephole_lattice_ahead(_peephole_perlin_ahead(records, b, rec), b, rec)
    if FILL_AHEAD:
        records = _peephole_fill_ahead(records, b, rec)
    if PYRAMID_AHEAD:
        records = _peephole_pyramid_ahead(records, b, rec)
    # what the result owns must be fresh per call: the tensors handed back and the statistics partials tagged onto them
    owned = {}  # id(temp) -> index in the fresh list

    def spec_of(obj):
        if obj is None:
            return ("n",)
        if isinstance(obj, tuple):
            return ("u", tuple(spec_of(o) for o in obj))
        if not isinstance(obj, torch.Tensor):
            raise PlanError(f"a result of type {type(obj).__name__}")
        base = next((t for lo, hi, t in b.temp_ranges if lo == obj.data_ptr()), None)
        if base is None or not obj.is_contiguous() or obj.numel() != base.numel() or obj.dtype != base.dtype:
            raise PlanError("the result is not a whole tensor the call allocated")
        tag = obj.__dict__.get(STATS_ATTR)
        tag_idx = None
        if tag is not None:
            partials, version = tag
            if version != obj._version:
                tag = None
            else:
                pt = next((t for lo, hi, t in b.temp_ranges if lo == partials.data_ptr()), None)
                if pt is None or pt.numel() != partials.numel():
                    raise PlanError("a statistics tag the call did not allocate")
                tag_idx = owned.setdefault(id(pt), len(owned))
        idx = owned.setdefault(id(base), len(owned))
        return ("t", idx, None if tuple(obj.shape) == tuple(base.shape) else tuple(obj.shape), tag_idx)

    result_spec = spec_of(result)
    fresh = [None] * len(owned)
    scratch = []
    by_id = {id(t): t for _lo, _hi, t in b.temp_ranges}
    for tid, idx in owned.items():
        t = by_id[tid]
        if tid not in b.slot_of:  # a result no kernel wrote?  (cannot happen on a device path)
            raise PlanError("a result tensor that no entry point was handed")
        fresh[idx] = (b.slot_of[tid], tuple(t.shape), t.dtype)
    hook_slots = {}
    for key, slot in b.slot_of.items():
        if isinstance(key, tuple) and key[0] == "hook":
            hook_slots[key[1]] = slot
        elif key not in owned:
            t = by_id[key]
            # a scratch tensor somebody else still holds after the call (a generator's cache filled during this very call) is not scratch:
            # every live tensor on its storage shows in the storage's use count (2 = the trace's own note + the handle asked for here)
            if torch._C._storage_Use_Count(t.untyped_storage()._cdata) > 2:
                raise PlanError("a tensor allocated during the call outlives it")
            scratch.append((slot, tuple(t.shape), t.dtype))
            b.scratch_refs.append(weakref.ref(t))
    for h in rec.hooks:
        h.bind(hook_slots)
    device = next(iter(by_id.values())).device if by_id else torch.device("cuda", torch.cuda.current_device())
    lib = load_raw()
    nslots = len(b.slots)
    handle = lib.sonar_plan_create(nslots)
    if not handle:
        raise PlanError("sonar_plan_create failed")
    try:
        for name, words, blob, patches in records:
            arr = (C.c_uint64 * len(words))(*words)
            parr = (PlanPatch * max(len(patches), 1))(*patches)
            buf = (C.c_char * max(len(blob), 1)).from_buffer_copy(blob.ljust(1, b"\0"))
            rc = lib.sonar_plan_add(handle, lib.sonar_plan_fn_id(name.encode()), arr, len(words), C.cast(buf, C.c_void_p), len(blob),
                                    C.cast(parr, C.c_void_p), len(patches))
            if rc != 0:
                raise PlanError(f"sonar_plan_add({name}): {lib.sonar_last_error().decode('utf-8', 'replace')}")
    except Exception:
        lib.sonar_plan_destroy(handle)
        raise
    plan = Plan(handle, nslots, fresh, scratch, result_spec, b.constants, rec, take, rewind, guards, device)
    plan.scratch_refs = b.scratch_refs
    # the recursive spec_of closure keeps this frame (and through it every traced tensor) in a reference cycle: let go explicitly
    b.temp_ranges, b.slots, b.slot_of = [], [], {}
    by_id.clear()
    return plan


class Planned:
    """``fn(sigma, sigma_next)`` with a prepared plan in front of it: the first calls take the ordinary path, one of them is traced, and
    from then on a call is ``Plan.run()`` -- unless a guard changed or an entry point refused, in which case that call is the ordinary
    path again.  Only for steps that do not depend on their arguments (the caller vouches for that: ``plan_static``)."""

    def __init__(self, fn, *, take, rewind, guards=()):
        self.fn, self.take, self.rewind, self.guards = fn, take, rewind, tuple(guards)
        self.plan = None
        self.calls = 0
        self.attempts = 0
        self.reason = None
        # A sampler is a stateful object: it consumes an RNG sequence and keeps what one call leaves for the next (look-ahead statistics,
        # lattices, the plan's per-stream slot table).  Two threads calling the SAME sampler are serialised here (the C side of a replay is
        # re-entrant, sonar_plan_run; different samplers do not share the lock): each call gets its own RNG position and its own result.
        self.lock = _threading.RLock()
        for name in ("unscaled", "accumulate", "fold_prefix", "accepts_prefix", "normalized_call", "plan_static"):
            if hasattr(fn, name):
                setattr(self, name, getattr(fn, name))

    def __call__(self, sigma=None, sigma_next=None):
        with self.lock:
            return self._call(sigma, sigma_next)

    def _call(self, sigma, sigma_next):
        rec = _recorder  # (one read, as in load())
        if PLANS_ENABLED and (rec is None or rec.thread != _threading.get_ident()):
            plan = self.plan
            if plan is not None:
                out = plan.run()
                if out is not NOT_RUN:
                    return out
            elif self.attempts < PLAN_MAX_ATTEMPTS:
                self.calls += 1
                if self.calls > PLAN_WARM_CALLS:
                    self.attempts += 1
                    guards = tuple((g, g()) for g in self.guards)
                    out, self.plan = trace_plan(self.fn, (sigma, sigma_next), take=self.take, rewind=self.rewind, guards=guards)
                    self.reason = trace_plan.last_reason
                    return out
        return self.fn(sigma, sigma_next)
