"""ctypes binding of ``librvdd_hip.so`` (declared in ``include/rvdd.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C
rvdd-release_amd/csrc``).  Loading fails loudly when it is missing: there is
no CPU fallback behind this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librvdd_hip.so")

RVDD_OK = 0
ARCH_CONVUNET, ARCH_CONVUNET_FEAT, ARCH_CONVNEXT, ARCH_CONVNEXT_FEAT = 0, 1, 2, 3


class RvddCfg(C.Structure):
    _fields_ = [("arch", C.c_int32), ("future", C.c_int32), ("batch", C.c_int32),
                ("height", C.c_int32), ("width", C.c_int32), ("device", C.c_int32)]


# symbol -> (restype, argtypes); exactly the functions include/rvdd.h declares
_P = C.c_void_p
_PROTOS = {
    "rvdd_create": (C.c_int, [C.POINTER(RvddCfg), C.POINTER(_P)]),
    "rvdd_destroy": (None, [_P]),
    "rvdd_last_error": (C.c_char_p, [_P]),
    "rvdd_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32]),
    "rvdd_finalize_weights": (C.c_int, [_P]),
    "rvdd_reset": (C.c_int, [_P]),
    "rvdd_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "rvdd_step_strided": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, _P, _P]),
    "rvdd_get_state": (C.c_int, [_P, _P, _P, _P]),
    "rvdd_set_state": (C.c_int, [_P, _P, _P, _P]),
    "rvdd_psnr_l1": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_float), _P]),
    "rvdd_unet_forward": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "rvdd_demosaic_ha": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "rvdd_warp_bicubic": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "rvdd_upsample_factor_2": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                         _P, _P]),
    "rvdd_tvl1flow": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _P]),
    "rvdd_tiff_lzw_decode": (C.c_int64, [_P, C.c_int64, _P, C.c_int64]),
    "rvdd_set_option": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "rvdd_tvl1flow_batch": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _P]),
    "rvdd_ppipe": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                             C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, _P, _P, _P]),
    "rvdd_srgb_metrics": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), _P]),
    "rvdd_profile_enable": (C.c_int, [_P, C.c_int32]),
    "rvdd_profile_select": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "rvdd_profile_count": (C.c_int, [_P]),
    "rvdd_profile_read": (C.c_int, [_P, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rvdd_timer_start": (C.c_int, [_P, _P]),
    "rvdd_timer_stop_ms": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "rvdd_debug_conv_bench": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), _P]),
    "rvdd_version": (C.c_char_p, []),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP runtime library; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C rvdd-release_amd/csrc`). This package has no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 and an
    # unversioned libhsa-runtime64.so.  Import torch FIRST so that our NEEDED
    # libamdhip64.so.7 resolves to the copy torch already mapped; loaded the
    # other way round the process ends up with two HSA runtimes and
    # hipGetDeviceCount() reports no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_PROTOS)
