"""Process-wide runtime state of the HIP path: compute dtype, stream handle, small helpers."""
import os

import torch

from . import _lib

_STATE = {"dtype": os.environ.get("CPCSV_DTYPE", "bf16"), "subpixel": os.environ.get("CPCSV_SUBPIXEL", "1") != "0"}


def set_compute_dtype(name):
    """'bf16' (MFMA bf16 operands, fp32 accumulate — the performance mode BASELINE.json names)
    or 'fp32' (exact f32 MFMA — the parity mode; the reference is fp32 only)."""
    if name not in ("bf16", "fp32"):
        raise ValueError("compute dtype must be 'bf16' or 'fp32'")
    _STATE["dtype"] = name


def set_subpixel(on):
    """Sub-pixel form of upsample+conv3x3 (2.25x fewer FLOPs; default on). Read when a layer is first planned."""
    _STATE["subpixel"] = bool(on)


def subpixel():
    return _STATE["subpixel"]


def compute_dtype_name():
    return _STATE["dtype"]


def tdtype():
    return torch.bfloat16 if _STATE["dtype"] == "bf16" else torch.float32


def dcode(t=None):
    if t is None:
        return _lib.BF16 if _STATE["dtype"] == "bf16" else _lib.F32
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.bfloat16:
        return _lib.BF16
    raise TypeError("unsupported dtype %s" % t.dtype)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_current_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else torch.cuda.current_device


def stream():
    """hipStream_t of torch's current stream (raw handle; ~0.2 us instead of ~9 us through torch.cuda.current_stream())."""
    if _raw_stream is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


def pad8(n):
    return (int(n) + 7) // 8 * 8


def ptr(t):
    return None if t is None else t.data_ptr()


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("cpcsv ops run on the GPU only (HIP kernels); got a %s tensor. "
                           "There is no CPU fallback in the product path." % t.device)
