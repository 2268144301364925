"""ctypes binding of libhdrnet_amd.so -- the C-ABI declared in include/hdrnet_amd.h.

The library is the product: there is NO fallback.  If it cannot be built / loaded the
import of the ops fails loudly (``HdrnetLibraryError``); nothing here routes to PyTorch
eager code or to any CPU checker.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

HDRNET_OK = 0
HDRNET_INVALID_ARGUMENT = 1
HDRNET_RUNTIME_FAILURE = 2

KERNEL_AUTO = 0
GUIDE_SIGMOID_FAST = 0x10000  # HDRNET_GUIDE_SIGMOID_FAST (flags of the guide-network ..._ex entry points)
GUIDE_RELU_PRESCALED = 0x20000  # HDRNET_GUIDE_RELU_PRESCALED: conv1 / conv2 are hdrnet_guide_nn_prescale_f32's arrays
KERNEL_GENERIC = 1
KERNEL_FAST = 2

_FP = ctypes.c_void_p  # device pointers travel as integers (tensor.data_ptr())
_I = ctypes.c_int
_U = ctypes.c_uint
_SZ = ctypes.c_size_t
_VP = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/hdrnet_amd.h declares
# (tests/test_capi_symbols.py cross-checks this table against the header).
SIGNATURES = {
    "hdrnet_version": (_I, []),
    "hdrnet_last_error": (ctypes.c_char_p, []),
    "hdrnet_last_kernel": (ctypes.c_char_p, []),
    "hdrnet_enable_kernel_names": (None, [_I]),
    "hdrnet_bilateral_slice_apply_f32": (_I, [_FP] * 4 + [_I] * 9 + [_VP]),
    "hdrnet_bilateral_slice_apply_f32_ex": (_I, [_FP] * 4 + [_I] * 9 + [_U, _VP]),
    "hdrnet_bilateral_slice_apply_rows_f32": (_I, [_FP] * 4 + [_I] * 11 + [_VP]),
    "hdrnet_bilateral_slice_apply_rows_f32_ex": (_I, [_FP] * 4 + [_I] * 11 + [_U, _VP]),
    "hdrnet_bilateral_slice_apply_nnguide_f32": (_I, [_FP] * 6 + [_I] * 10 + [_VP]),
    "hdrnet_bilateral_slice_apply_nnguide_f32_ex": (_I, [_FP] * 6 + [_I] * 10 + [_U, _VP]),
    "hdrnet_bilateral_slice_apply_io_curves": (_I, [_FP] * 3 + [_I] * 10 + [ctypes.c_float, _I] + [_FP] * 4 + [_I, _FP, _VP]),
    "hdrnet_curves_guide_prepared_bytes": (_SZ, [_I]),
    "hdrnet_curves_guide_prepare_f32": (_I, [_FP, _FP, _I, _I, _VP, _SZ, ctypes.POINTER(ctypes.c_int), _VP]),
    "hdrnet_bilateral_slice_apply_io_curves_prepared": (_I, [_FP] * 3 + [_I] * 10 + [ctypes.c_float, _I] + [_FP] * 4 + [_I, _VP, _FP, _VP]),
    "hdrnet_bilateral_slice_apply_upadd_f32": (_I, [_FP] * 4 + [_I, _I, _FP] + [_I] * 9 + [_FP, _FP, _I, _VP]),
    "hdrnet_bilateral_slice_apply_upadd_f32_ex": (_I, [_FP] * 4 + [_I, _I, _FP] + [_I] * 9 + [_FP, _FP, _I, _U, _VP]),
    "hdrnet_resize_bilinear_f32": (_I, [_FP, _FP] + [_I] * 6 + [_VP]),
    "hdrnet_pointwise_guide_grad_workspace_bytes": (_SZ, [ctypes.c_longlong, _I, _I]),
    "hdrnet_pointwise_guide_grad_f32": (_I, [_FP] * 6 + [_I] + [_FP] * 2 + [ctypes.c_longlong, _I, _I, _VP, _SZ, _VP]),
    "hdrnet_curves_guide_grad_workspace_bytes": (_SZ, [ctypes.c_longlong, _I, _I]),
    "hdrnet_curves_guide_grad_f32": (_I, [_FP] * 7 + [_I] + [_FP] * 4 + [ctypes.c_longlong, _I, _I, _VP, _SZ, _VP]),
    "hdrnet_input_moments_workspace_bytes": (_SZ, [ctypes.c_longlong, _I]),
    "hdrnet_input_moments_f32": (_I, [_FP, ctypes.c_longlong, _I, _FP, _FP, _VP, _SZ, _VP]),
    "hdrnet_l2_loss_workspace_bytes": (_SZ, [ctypes.c_longlong]),
    "hdrnet_l2_loss_f32": (_I, [_FP, _FP, ctypes.c_longlong, _FP, _VP, _SZ, _VP]),
    "hdrnet_l2_loss_grad_f32": (_I, [_FP, _FP, _FP, ctypes.c_longlong, _FP, _VP]),
    "hdrnet_guide_nn_prescale_f32": (_I, [_FP, _FP, _I, _I, ctypes.c_float, _FP, _FP, _VP]),
    "hdrnet_guide_fold_batch_f32": (_I, [_FP, _FP, ctypes.c_longlong] + [_FP] * 5 + [ctypes.c_double, ctypes.c_double, _I, _I] + [_FP] * 5 + [_VP]),
    "hdrnet_guide_fold_batch_grad_f32": (_I, [_FP, _FP, ctypes.c_longlong] + [_FP] * 3 + [ctypes.c_double, _I, _I] + [_FP] * 6 + [_VP]),
    "hdrnet_coefficients_workspace_bytes": (_SZ, [_VP, _I]),
    "hdrnet_coefficients_f32": (_I, [_FP, _VP, _FP, _I, _VP, _SZ, _VP]),
    "hdrnet_coefficients_grad_workspace_bytes": (_SZ, [_VP, _I]),
    "hdrnet_coefficients_grad_f32": (_I, [_FP, _VP, _VP, _FP, _VP, _I, _VP, _SZ, _VP]),
    "hdrnet_bilateral_slice_apply_io": (_I, [_FP] * 4 + [_I] * 10 + [ctypes.c_float, _I] + [_FP] * 2 + [_I, _FP, _VP]),
    "hdrnet_bilateral_slice_apply_io_ex": (_I, [_FP] * 4 + [_I] * 10 + [ctypes.c_float, _I] + [_FP] * 2 + [_I, _FP, _U, _VP]),
    "hdrnet_bilateral_slice_apply_grad_workspace_bytes": (_SZ, [_I] * 9),
    "hdrnet_bilateral_slice_apply_grad_f32": (_I, [_FP] * 7 + [_I] * 9 + [_VP, _SZ, _VP]),
    "hdrnet_bilateral_slice_apply_grad_f32_ex": (_I, [_FP] * 7 + [_I] * 9 + [_VP, _SZ, _U, _VP]),
    "hdrnet_bilateral_slice_f32": (_I, [_FP] * 3 + [_I] * 7 + [_VP]),
    "hdrnet_bilateral_slice_f32_ex": (_I, [_FP] * 3 + [_I] * 7 + [_U, _VP]),
    "hdrnet_bilateral_slice_grad_workspace_bytes": (_SZ, [_I] * 7),
    "hdrnet_bilateral_slice_grad_f32": (_I, [_FP] * 5 + [_I] * 7 + [_VP, _SZ, _VP]),
    "hdrnet_bilateral_slice_grad_f32_ex": (_I, [_FP] * 5 + [_I] * 7 + [_VP, _SZ, _U, _VP]),
}


class CoeffNet(ctypes.Structure):
    """``hdrnet_coeff_net`` of include/hdrnet_amd.h (the coefficient network's hyper-parameters and parameters)."""

    _fields_ = [("net_input_size", _I), ("spatial_bin", _I), ("luma_bins", _I), ("channel_multiplier", _I),
                ("n_out", _I), ("n_in", _I), ("n_levels", _I),
                ("splat_w", _VP * 8), ("splat_b", _VP * 8),
                ("global_conv_w", _VP * 2), ("global_conv_b", _VP * 2),
                ("fc_w", _VP * 3), ("fc_b", _VP * 3),
                ("local_w", _VP * 2), ("local_b", _VP * 2),
                ("pred_w", _VP), ("pred_b", _VP), ("fc_layout", _I)]


class CoeffNetGrads(ctypes.Structure):
    """``hdrnet_coeff_net_grads``: where ``hdrnet_coefficients_grad_f32`` writes the parameter gradients."""

    _fields_ = [("splat_w", _VP * 8), ("splat_b", _VP * 8),
                ("global_conv_w", _VP * 2), ("global_conv_b", _VP * 2),
                ("fc_w", _VP * 3), ("fc_b", _VP * 3),
                ("local_w", _VP * 2), ("local_b", _VP * 2),
                ("pred_w", _VP), ("pred_b", _VP)]


class HdrnetLibraryError(RuntimeError):
    """libhdrnet_amd.so is missing, failed to build, or failed to load."""


class HdrnetInvalidArgument(ValueError):
    """HDRNET_INVALID_ARGUMENT (the reference: tensorflow::errors::InvalidArgument)."""


class HdrnetRuntimeError(RuntimeError):
    """HDRNET_RUNTIME_FAILURE (the reference: errors::Internal("... kernel failed."))."""


# include/hdrnet_amd_train.h: training-loop helpers outside the operator boundary (same library).
TRAIN_SIGNATURES = {
    "hdrnet_l2_loss_with_grad_f32": (_I, [_FP, _FP, ctypes.c_longlong, _FP, _FP, _VP, ctypes.c_size_t, _VP]),
    "hdrnet_l2_loss_grad_scale_f32": (_I, [_FP, _FP, ctypes.c_longlong, _VP]),
    "hdrnet_resize_add_f32": (_I, [_FP, _FP, _FP] + [_I] * 6 + [_VP]),
    "hdrnet_resize_bilinear_grad_f32": (_I, [_FP, _FP] + [_I] * 6 + [_VP]),
    "hdrnet_adam_step_f32": (_I, [_FP, _FP, _FP, _FP, ctypes.c_longlong, _FP] + [ctypes.c_float] * 4 + [_VP]),
    "hdrnet_adam_step_tf_f32": (_I, [_FP, _FP, _FP, _FP, ctypes.c_longlong, _FP] + [ctypes.c_float] * 4 + [_VP]),
}

_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None
_tools_lib: Optional[ctypes.CDLL] = None

# Symbols only the tools build exports (include/hdrnet_amd_tools.h).
TOOLS_SIGNATURES = {
    "hdrnet_tools_set_trace": (None, [_VP]),
    "hdrnet_tools_set_knob": (None, [ctypes.c_int, ctypes.c_int]),
    "hdrnet_tools_pyramid_onepass_f32": (_I, [_VP] * 4 + [_I, _FP] + [_I] * 7 + [_U, _VP]),
}


def lib_path() -> str:
    from . import build as _build

    return _build.LIB_PATH


def _open(tools: bool) -> ctypes.CDLL:
    # torch must be imported first so that its bundled libamdhip64.so.7 is the
    # HIP runtime both sides share (same SONAME => the loader reuses it).
    import torch  # noqa: F401

    from . import build as _build

    target = _build.TOOLS_LIB_PATH if tools else _build.LIB_PATH
    try:
        path = _build.build(tools=tools)
    except Exception as e:  # noqa: BLE001
        # A library that is older than its sources is NOT silently acceptable: tests passing on a
        # stale binary is exactly what the "which .so was loaded" check cannot see.  Refuse unless
        # the caller explicitly allows it (a host without hipcc that was handed a prebuilt tree).
        if os.path.exists(target) and os.environ.get("HDRNET_AMD_ALLOW_STALE_LIB") == "1":
            import warnings

            warnings.warn(f"hdrnet_amd: rebuilding {os.path.basename(target)} FAILED ({e}); loading the "
                          "existing, possibly STALE library because HDRNET_AMD_ALLOW_STALE_LIB=1",
                          RuntimeWarning, stacklevel=3)
            path = target
        else:
            hint = (" (an older library exists; set HDRNET_AMD_ALLOW_STALE_LIB=1 to load it anyway)"
                    if os.path.exists(target) else "")
            raise HdrnetLibraryError(
                f"cannot build {os.path.basename(target)} (hipcc for gfx950 required): {e}{hint}") from e
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise HdrnetLibraryError(f"cannot load {path}: {e}") from e
    table = dict(SIGNATURES)
    table.update(TRAIN_SIGNATURES)
    if tools:
        table.update(TOOLS_SIGNATURES)
    for name, (res, args) in table.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HdrnetLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def load() -> ctypes.CDLL:
    """Load the product library (building first if the in-tree library is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            _lib = _open(tools=False)
    return _lib


def load_tools() -> ctypes.CDLL:
    """Load the TOOLS build (benchmark variants, memory skeletons, timeline trace); used by
    tools/*.py and the variant tests only -- never by the ops."""
    global _tools_lib
    if _tools_lib is not None:
        return _tools_lib
    with _lock:
        if _tools_lib is None:
            _tools_lib = _open(tools=True)
    return _tools_lib


def enable_kernel_names(on: bool = True) -> None:
    """Switch the hdrnet_last_kernel() bookkeeping on / off (off by default)."""
    load().hdrnet_enable_kernel_names(1 if on else 0)


def last_error() -> str:
    return load().hdrnet_last_error().decode()


def last_kernel() -> str:
    return load().hdrnet_last_kernel().decode()


def check(rc: int, what: str) -> None:
    if rc == HDRNET_OK:
        return
    msg = f"{what}: {last_error()}"
    if rc == HDRNET_INVALID_ARGUMENT:
        raise HdrnetInvalidArgument(msg)
    raise HdrnetRuntimeError(msg)
