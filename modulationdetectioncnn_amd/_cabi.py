"""ctypes binding of libmdc.so (include/mdc.h).  This is the stub INTEGRATION.md shows.

The product path has no CPU fallback: if the shared library is missing, cannot be
loaded, or reports no gfx950 device, the calls below raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmdc.so")
# "alternates" = the -DMDC_ALTERNATES test build (build.py): the product kernels plus the measured-slower alternates the
# GPU suite holds them against.  Tests ask for it by name; nothing in the package does.
LIB_PATHS = {"product": LIB_PATH, "alternates": os.path.join(_HERE, "libmdc_alt.so")}

KIND_DEPLOYED, KIND_VTCNN2, KIND_CNNPY = 1, 2, 3
F32, BF16, FP8, F16 = 0, 1, 2, 3
TAP_NONE, TAP_CONV, TAP_FLAT, TAP_DENSE, TAP_HIDDEN = 0, 1, 2, 3, 4

EXPORTS = [
    "mdc_abi_version", "mdc_create", "mdc_num_layers", "mdc_layer_sizes", "mdc_set_weights",
    "mdc_finalize", "mdc_workspace_bytes", "mdc_forward", "mdc_set_profiling", "mdc_profile_slots",
    "mdc_profile_name", "mdc_profile_read", "mdc_profile_reset", "mdc_last_error", "mdc_destroy",
    "mdc_forward_q612", "mdc_confusion", "mdc_iq_u8_to_frames", "mdc_set_fp8_input_absmax",
    "mdc_forward_iq_u8", "mdc_confusion_binned", "mdc_iq_u8_windows", "mdc_predict_host", "mdc_predict_host_iq_u8",
    "mdc_crossentropy", "mdc_set_fp8_feature_absmax",
    "mdc_trainer_create", "mdc_trainer_num_layers", "mdc_trainer_layer_sizes", "mdc_trainer_set_adam", "mdc_trainer_set_dropout", "mdc_trainer_set_tensor",
    "mdc_trainer_get_tensor", "mdc_trainer_set_iterations", "mdc_train_batch", "mdc_trainer_evaluate", "mdc_trainer_read",
    "mdc_trainer_destroy",
]
ABI_VERSION = 5
TRAIN_WEIGHTS, TRAIN_ADAM_M, TRAIN_ADAM_V, TRAIN_GRADIENT = 0, 1, 2, 3
MDC_OPT_FP8_BF16_FEATURES = 1      # include/mdc.h: option bit in mdc_topology.reserved[0]
HOP_FRAME = 128


class MdcTopology(C.Structure):
    _fields_ = [("kind", C.c_int32), ("filters", C.c_int32), ("hidden", C.c_int32),
                ("classes", C.c_int32), ("reserved", C.c_int32 * 4)]


class MdcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmdc error {code}: {msg}")
        self.code = code


_libs: dict = {}


def lib(variant: str = "product") -> C.CDLL:
    """Load libmdc.so (once).  Raises if the HIP extension has not been built."""
    if variant in _libs:
        return _libs[variant]
    path = LIB_PATHS[variant]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m modulationdetectioncnn_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    # torch FIRST: its wheel bundles its own libamdhip64.so, libmdc.so names the system's as DT_NEEDED, and whichever HIP runtime a
    # process loads first serves both (same soname).  Loaded the other way round the process ends up with TWO runtimes, and
    # the one under libmdc.so then finds no device (`mdc_create`: MDC_ENODEV) -- measured on the GPU box, round 5.  The Python
    # mirror hands torch tensors to the library anyway; a caller without torch (examples/c_client.c) has one runtime by construction.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    L.mdc_abi_version.restype = i32
    L.mdc_create.argtypes = [C.POINTER(MdcTopology), i32, C.POINTER(vp)]
    L.mdc_num_layers.argtypes = [vp]
    L.mdc_layer_sizes.argtypes = [vp, i32, C.POINTER(sz), C.POINTER(sz)]
    L.mdc_set_weights.argtypes = [vp, i32, C.POINTER(C.c_float), sz, C.POINTER(C.c_float), sz]
    L.mdc_finalize.argtypes = [vp, i32]
    L.mdc_workspace_bytes.argtypes = [vp, i64]
    L.mdc_workspace_bytes.restype = sz
    L.mdc_forward.argtypes = [vp, vp, i64, vp, vp, vp, i32, vp, sz, vp]
    L.mdc_set_profiling.argtypes = [vp, i32]
    L.mdc_profile_slots.argtypes = [vp]
    L.mdc_profile_name.argtypes = [vp, i32]
    L.mdc_profile_name.restype = C.c_char_p
    L.mdc_profile_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64)]
    L.mdc_profile_reset.argtypes = [vp]
    L.mdc_last_error.restype = C.c_char_p
    L.mdc_destroy.argtypes = [vp]
    L.mdc_destroy.restype = None
    L.mdc_forward_q612.argtypes = [vp, vp, i32, i64, vp, vp, vp]
    L.mdc_confusion.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.mdc_iq_u8_to_frames.argtypes = [vp, i64, C.c_float, vp, vp]
    L.mdc_set_fp8_input_absmax.argtypes = [vp, C.c_float]
    L.mdc_set_fp8_feature_absmax.argtypes = [vp, C.c_float]
    L.mdc_forward_iq_u8.argtypes = [vp, vp, i64, i64, C.c_float, vp, vp, vp, sz, vp]
    L.mdc_confusion_binned.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp]
    L.mdc_iq_u8_windows.argtypes = [vp, i64, i64, C.c_float, vp, vp]
    L.mdc_predict_host.argtypes = [vp, vp, i64, vp, vp, i64]
    L.mdc_predict_host_iq_u8.argtypes = [vp, vp, i64, i64, C.c_float, vp, vp, i64]
    L.mdc_crossentropy.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    fp = C.POINTER(C.c_float)
    L.mdc_trainer_create.argtypes = [C.POINTER(MdcTopology), i32, C.POINTER(vp)]
    L.mdc_trainer_num_layers.argtypes = [vp]
    L.mdc_trainer_layer_sizes.argtypes = [vp, i32, C.POINTER(sz), C.POINTER(sz)]
    L.mdc_trainer_set_adam.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
    L.mdc_trainer_set_dropout.argtypes = [vp, C.c_float, C.c_uint32]
    L.mdc_trainer_set_tensor.argtypes = [vp, i32, i32, fp, sz, fp, sz, vp]
    L.mdc_trainer_get_tensor.argtypes = [vp, i32, i32, fp, sz, fp, sz, vp]
    L.mdc_trainer_set_iterations.argtypes = [vp, i64, vp]
    L.mdc_train_batch.argtypes = [vp, vp, vp, i64, vp, i64, i64, i32, vp]
    L.mdc_trainer_evaluate.argtypes = [vp, vp, vp, i64, vp, i64, i64, vp]
    L.mdc_trainer_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64), vp]
    L.mdc_trainer_destroy.argtypes = [vp]
    L.mdc_trainer_destroy.restype = None
    for name in ("mdc_create", "mdc_num_layers", "mdc_layer_sizes", "mdc_set_weights", "mdc_finalize",
                 "mdc_forward", "mdc_set_profiling", "mdc_profile_slots", "mdc_profile_read", "mdc_profile_reset",
                 "mdc_forward_q612", "mdc_confusion", "mdc_iq_u8_to_frames", "mdc_set_fp8_input_absmax", "mdc_forward_iq_u8",
                 "mdc_confusion_binned", "mdc_iq_u8_windows", "mdc_predict_host", "mdc_predict_host_iq_u8", "mdc_crossentropy", "mdc_set_fp8_feature_absmax",
                 "mdc_trainer_create", "mdc_trainer_num_layers", "mdc_trainer_layer_sizes", "mdc_trainer_set_adam", "mdc_trainer_set_dropout", "mdc_trainer_set_tensor",
                 "mdc_trainer_get_tensor", "mdc_trainer_set_iterations", "mdc_train_batch", "mdc_trainer_evaluate", "mdc_trainer_read"):
        getattr(L, name).restype = i32
    if L.mdc_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{os.path.basename(path)} ABI version {L.mdc_abi_version()} != {ABI_VERSION}; rebuild it")
    _libs[variant] = L
    return L


def check(rc: int, variant: str = "product") -> int:
    if rc < 0:
        raise MdcError(rc, lib(variant).mdc_last_error().decode("utf-8", "replace"))
    return rc
