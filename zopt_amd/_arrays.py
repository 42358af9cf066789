"""Array plumbing between NumPy / torch-ROCm callers and the C ABI (device pointers)."""
from __future__ import annotations

import numpy as np

from ._lib import ZoptAmdError

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def require_gpu():
    if torch is None or not torch.cuda.is_available():
        raise ZoptAmdError("zopt_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback exists")


def to_device(x, dtype, device=None):
    """-> contiguous torch ROCm tensor of `dtype` (NumPy inputs are copied host-to-device)."""
    require_gpu()
    if is_torch(x):
        t = x
        if not t.is_cuda:
            t = t.to(device or "cuda")
    else:
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(x)), device=device or "cuda")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def result_like(t, template):
    """Return `t` (torch ROCm) in the caller's array family: NumPy in -> NumPy out (synchronises)."""
    if is_torch(template):
        return t
    return t.cpu().numpy()


def stream_ptr(t):
    return torch.cuda.current_stream(t.device).cuda_stream
