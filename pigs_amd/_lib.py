"""ctypes binding of the C ABI in include/pigs_amd.h (pigs_amd/libpigs_amd.so).

There is no fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

# torch must be imported before libpigs_amd.so is loaded: both need libamdhip64, and the library
# must bind to the HIP runtime instance PyTorch brings along (loaded the other way round, the
# process holds two runtimes and ours reports "no ROCm-capable device").
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
# PIGS_AMD_LIB selects another build of the same ABI (kernel experiments); default: in-tree library
LIB_PATH = os.environ.get("PIGS_AMD_LIB") or os.path.join(HERE, "libpigs_amd.so")

PIGS_F32, PIGS_F64 = 0, 1
ABI_VERSION = 8

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64

# name -> (restype, argtypes); must list every symbol include/pigs_amd.h declares
SIGNATURES = {
    "pigs_abi_version": (_i, []),
    "pigs_status_string": (ctypes.c_char_p, [_i]),
    "pigs_last_hip_error": (ctypes.c_char_p, []),
    "pigs_sample_forward": (_i, [_i, _i, _i, _i, _i64, _i64] + [_vp] * 4 + [_vp] * 4 + [_vp]),
    "pigs_sample_backward": (_i, [_i, _i, _i, _i, _i64, _i64] + [_vp] * 4 + [_vp] * 4 + [_vp] * 3 + [_vp]),
    "pigs_build_covariances": (_i, [_i, _i64] + [_vp] * 4 + [_vp]),
    "pigs_build_covariances_backward": (_i, [_i, _i64] + [_vp] * 6 + [_vp]),
    "pigs_samples_workspace_bytes": (ctypes.c_size_t, [_i64]),
    "pigs_plan_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i]),
    "pigs_plan_layout_info": (_i, [_i64, _i64, _i, ctypes.POINTER(_i64)]),
    "pigs_samples_error_offset": (ctypes.c_size_t, []),
    "pigs_plan_error_offset": (ctypes.c_size_t, []),
    "pigs_samples_lattice_offset": (ctypes.c_size_t, []),
    "pigs_plan_strips_offset": (ctypes.c_size_t, []),
    "pigs_samples_build": (_i, [_vp, ctypes.c_size_t, _i64, _vp, _vp]),
    "pigs_samples_order_hint": (_i, [_i64]),
    "pigs_plan_build": (_i, [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _i, _i64, _i64, _i, ctypes.c_float, ctypes.c_float]
                        + [_vp] * 4 + [_vp]),
    "pigs_plan_forward": (_i, [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _i64, _i64, _i, ctypes.c_float, _i]
                          + [_vp] * 4 + [_vp]),
    "pigs_plan_backward": (_i, [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _i64, _i64, _i, ctypes.c_float, _i]
                           + [_vp] * 4 + [_vp] * 3 + [_vp]),
    "pigs_residual_forward": (_i, [_i, _i, _i, _i64, _i64] + [_vp] * 4 + [ctypes.POINTER(ctypes.c_double), _vp, _vp]
                              + [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp]),
    "pigs_residual_backward": (_i, [_i, _i, _i, _i64, _i64] + [_vp] * 4 + [ctypes.POINTER(ctypes.c_double), _vp] + [_vp] * 3
                               + [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp]),
    "pigs_aggregate_workspace_bytes": (ctypes.c_size_t, [_i, _i64]),
    "pigs_aggregate_lists": (_i, [_i, _i64, _i64, _vp, _vp, ctypes.c_double, _vp, ctypes.c_size_t, _i] + [_vp] * 5 + [_vp]),
    "pigs_aggregate_forward": (_i, [_i, _i64, _i64, _i, _i, _i] + [_vp] * 4 + [_vp] * 6 + [_vp] * 3 + [_vp]),
    "pigs_aggregate_backward_scratch_bytes": (ctypes.c_size_t, [_i, _i64, _i, _i]),
    "pigs_aggregate_backward": (_i, [_i, _i64, _i64, _i, _i, _i] + [_vp] * 6 + [_vp] * 6 + [_vp] * 3 + [_vp, ctypes.c_size_t]
                                + [_vp] * 6 + [_vp]),
}

_lib = None


class PigsError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle of libpigs_amd.so."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP library first (python -m pigs_amd.build). "
            "pigs_amd has no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.pigs_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.pigs_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        lib = load()
        msg = lib.pigs_status_string(status).decode()
        if status == 3:
            msg += ": " + lib.pigs_last_hip_error().decode()
        raise PigsError(f"{what}: {msg} (status {status})")
