"""ctypes binding of liblbm_hip.so (the C ABI declared in include/lbm_abi.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this raises.
PyTorch is used only for device memory and streams; every pointer handed to the library is a
``tensor.data_ptr()`` of a CUDA (ROCm) tensor.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LBM_AMD_LIB") or os.path.join(_HERE, "liblbm_hip.so")   # env override: debug builds

LBM_OK = 0
PAD_REFLECT_LIMITED, PAD_ZERO = 0, 1
STD_NONE, STD_EPOCH_CHANNEL = 0, 1
WIN_BOXCAR, WIN_HANN, WIN_HAMMING = 0, 1, 2
AVG_MEAN, AVG_MEDIAN, AVG_NONE = 0, 1, 2
VQ_ROWS, VQ_CHANNEL_MAJOR, VQ_NO_XNORM = 0, 1, 2

_p, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float

# name -> (restype, argtypes); mirrors include/lbm_abi.h one to one
SIGNATURES = {
    "lbm_version": (C.c_int, []),
    "lbm_last_error": (C.c_char_p, []),
    "lbm_fir_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _i32, _i32, _p]),
    "lbm_fir_fft_tables_bytes": (C.c_size_t, [_i32, _i32]),
    "lbm_fir_fft_tables_init": (C.c_int, [_p, _p, _i32, _i32, _p]),
    "lbm_fir_fft_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _i32, _i32, _p]),
    "lbm_affine_ch_f32": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _p, _p, _p]),
    "lbm_ica_stats_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "lbm_ica_stats_f32": (C.c_int, [_p, _i64, _i64, _i32, _p, _i32, _f32, _p, _p, _p, _p]),
    "lbm_epochs_gather_f32": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _i64, _p]),
    "lbm_resample_length": (C.c_int64, [_i64, _i32, _i32]),
    "lbm_resample_workspace_bytes": (C.c_size_t, [_i64, _i64, _i32, _i32]),
    "lbm_resample_fft_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i64, _i32, _i32, _p, _p]),
    "lbm_epoch_post_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _p]),
    "lbm_affine_post_supported": (C.c_int, [_i32, _i32, _i32]),
    "lbm_affine_post_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _f32, _p]),
    "lbm_channel_moments_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "lbm_channel_moments_static_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _p]),
    "lbm_channel_scale_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "lbm_welch_tables_bytes": (C.c_size_t, [_i32, _i32, _i32, _i32]),
    "lbm_welch_tables_init": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p]),
    "lbm_welch_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _f32, _i32, _i32, _i32, _p, _i32, _i32, _i32, _p]),
    "lbm_welch_scaled_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _f32, _i32, _i32, _i32, _p, _i32, _i32, _i32, _p]),
    "lbm_welch_keep_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _f32, _i32, _i32, _i32, _p, _i32, _i32, _f32, _p]),
    "lbm_bandpower_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "lbm_conv1d_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "lbm_conv1d_pitched_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "lbm_vq_remap_codes_i64": (C.c_int, [_p, _i64, _p, _i32, _p, _p, _p]),
    "lbm_conv1d_k4s2_wino_supported": (C.c_int, [_i32, _i32]),
    "lbm_conv1d_k4s2_wino_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i32, _p]),
    "lbm_conv1d_wino3_tables_bytes": (C.c_size_t, [_i32, _i32]),
    "lbm_conv1d_wino3_tables_init": (C.c_int, [_p, _p, _i32, _i32, _p]),
    "lbm_conv1d_wino3_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i32, _p]),
    "lbm_conv1d_split_tables_bytes": (C.c_size_t, [_i32, _i32, _i32]),
    "lbm_conv1d_split_tables_init": (C.c_int, [_p, _p, _i32, _i32, _i32, _p]),
    "lbm_conv1d_bf16x6_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "lbm_rownorm2_f32": (C.c_int, [_p, _p, _i64, _i32, _p]),
    "lbm_vq_argmin_f32": (C.c_int, [_p, _i64, _i32, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "lbm_vq_split_tables_bytes": (C.c_size_t, [_i32, _i32]),
    "lbm_vq_split_tables_init": (C.c_int, [_p, _p, _i32, _i32, _p]),
    "lbm_vq_argmin_bf16x6_f32": (C.c_int, [_p, _i64, _i32, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "lbm_vq_screen_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "lbm_vq_argmin_screened_f32": (C.c_int, [_p, _i64, _i32, _i64, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "lbm_vq_gather_loss_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "lbm_vq_ema_workspace_bytes": (C.c_size_t, [_i64, _i32, _i32]),
    "lbm_vq_ema_sums_f32": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "lbm_vq_ema_apply_f32": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _f32, _f32, _f32, _p]),
    "lbm_perplexity_u64": (C.c_int, [_p, _p, _i32, _p]),
}

_lib = None


class LbmError(RuntimeError):
    pass


def load():
    """Load liblbm_hip.so (once).  Raises if it has not been built -- never falls back to CPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LbmError(f"{LIB_PATH} is missing: build it with `python -m lbm_amd.build` "
                           "(hipcc --offload-arch=gfx950); lbm_amd has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != LBM_OK:
        msg = load().lbm_last_error().decode(errors="replace")
        exc = ValueError if rc == -1 else (NotImplementedError if rc == -2 else LbmError)
        raise exc(f"{what or 'liblbm_hip'} failed (code {rc}): {msg}")


def stream_ptr() -> int:
    """The current stream of the CURRENT device: launches go to the current HIP device, so every tensor handed to a call
    must live there (``ptr`` checks it)."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Raises for CPU tensors and for tensors of another GPU than the current
    one: a kernel launched on device A's stream against device B's pointers faults or silently goes through peer access --
    wrap the call in ``with torch.cuda.device(t.device):`` instead."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LbmError("lbm_amd kernels need CUDA/ROCm tensors (no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        raise LbmError(f"tensor lives on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}: "
                       "run the call under `with torch.cuda.device(tensor.device)`")
    return t.data_ptr()


def on_tensor_device(fn):
    """Decorator of the public array-in/array-out wrappers: run ``fn`` with the device of its first CUDA tensor argument
    current, so that ``stream_ptr()`` / ``ptr()`` inside see that device -- a tensor on cuda:1 works while cuda:0 is current,
    as with plain torch ops.  Arguments that are not CUDA tensors (arrays, CPU tensors: moved to the current device by the
    wrapper) leave the current device alone."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index == torch.cuda.current_device():
                    break
                with torch.cuda.device(a.device):
                    return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapper


def dev_f32(t, device=None) -> torch.Tensor:
    """Contiguous float32 CUDA tensor view/copy of ``t`` (torch tensor or array-like)."""
    if not isinstance(t, torch.Tensor):
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(t))
    if device is None:
        device = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return t.to(device=device, dtype=torch.float32).contiguous()
