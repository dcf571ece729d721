"""ctypes binding of libb2h.so (the C ABI declared in include/b2h.h).

There is no CPU fallback: if the library is missing it is built with hipcc; if
that is impossible, or no gfx950 device is visible when a model is created, the
failure is loud (RuntimeError), never a silent eager-PyTorch path.
"""
import ctypes
import os

from . import build as _build

_fp = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p

# enum b2h_kernel
KERNEL_AUTO, KERNEL_F32_VALU, KERNEL_F32_MFMA, KERNEL_BF16_MFMA, KERNEL_F16_MFMA, KERNEL_F16X3_MFMA = range(6)
KERNELS = {"auto": KERNEL_AUTO, "fp32": KERNEL_AUTO, "f32": KERNEL_AUTO,
           "f32_valu": KERNEL_F32_VALU, "fp32_valu": KERNEL_F32_VALU,
           "f32_mfma": KERNEL_F32_MFMA, "fp32_mfma": KERNEL_F32_MFMA,
           "bf16": KERNEL_BF16_MFMA, "bf16_mfma": KERNEL_BF16_MFMA,
           "f16": KERNEL_F16_MFMA, "fp16": KERNEL_F16_MFMA, "f16_mfma": KERNEL_F16_MFMA,
           "f16x3": KERNEL_F16X3_MFMA, "f16x3_mfma": KERNEL_F16X3_MFMA}

# b2h_forward_fused flags
PRE_CHEST_DIFF, PRE_NORMALIZE, POST_DENORMALIZE, POST_MASK_TAIL = 1, 2, 4, 8

# enum b2h_status
OK, ERR_INVALID, ERR_SHAPE, ERR_NO_WEIGHTS, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6

SYMBOLS = {
    "b2h_version": (ctypes.c_int, []),
    "b2h_build_flags": (ctypes.c_int, []),
    "b2h_last_error": (ctypes.c_char_p, []),
    "b2h_device_count": (ctypes.c_int, []),
    "b2h_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(_vp)]),
    "b2h_destroy": (ctypes.c_int, [_vp]),
    "b2h_load_weights": (ctypes.c_int, [_vp] + [_vp] * 8 + [ctypes.c_int]),
    "b2h_forward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _vp]),
    "b2h_forward_fused": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_float, _vp, ctypes.c_int, _vp]),
    "b2h_target_transform": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_float, _vp]),
    "b2h_masked_l1": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp]),
    "b2h_weighted_l1": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp]),
    "b2h_tenc_create": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.POINTER(_vp)]),
    "b2h_tenc_destroy": (ctypes.c_int, [_vp]),
    "b2h_tenc_set_kernel": (ctypes.c_int, [_vp, ctypes.c_int]),
    "b2h_tenc_load_weights": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int]),
    "b2h_tenc_workspace_bytes": (ctypes.c_size_t, [_vp, ctypes.c_int64, ctypes.c_int64]),
    "b2h_tenc_forward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_size_t, _vp]),
    "b2h_tenc_forward_fused": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                              _vp, _vp, ctypes.c_size_t, _vp]),
    "b2h_model_info": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                      ctypes.POINTER(ctypes.c_int)]),
    "b2h_kernel_supported": (ctypes.c_int, [_vp, ctypes.c_int]),
    "b2h_kernel_name": (ctypes.c_char_p, [_vp, ctypes.c_int]),
    "b2h_time_forward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_float)]),
    "b2h_stream_sync": (ctypes.c_int, [_vp]),
}

_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load libb2h.so (building it first if needed) and type its entry points.

    torch is imported first on purpose: its bundled libamdhip64.so.7 then
    satisfies our DT_NEEDED entry (same SONAME), so tensors, streams and our
    kernels share ONE HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (HIP runtime owner; see docstring)
    path = _build.build()
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    flags = lib.b2h_build_flags()
    if flags != 0 and os.environ.get("B2H_ALLOW_ABLATE") != "1":
        # a timing-only development build (tools/ablate_*.sh) left in place of the product: wrong results by design
        raise RuntimeError(f"{path} is a B2H_ABLATE={flags} development build (results wrong by construction); rebuild "
                           f"with `python -m hand_pose_sl_amd.build --force`, or set B2H_ALLOW_ABLATE=1 to measure with it")
    _lib = lib
    return lib


def last_error():
    return load().b2h_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a b2h_status to the exception type the reference raises for it."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_INVALID:
        raise ValueError(msg)            # e.g. activation != "ReLU", HandPoseModels.py:34-37
    raise RuntimeError(f"libb2h: {msg} (status {rc})")


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(dev):
    """Context in which `dev` is the current HIP device: a no-op object when it already is (the
    usual one-process-per-GPU case), torch.cuda.device(dev) otherwise.  The library checks the
    current device on every forward (b2h.h), so the switch is for correctness, the shortcut for the
    few microseconds two hipSetDevice calls cost on a small-batch call."""
    import torch
    if torch.cuda.current_device() == dev.index:
        return _NO_SWITCH
    return torch.cuda.device(dev)
