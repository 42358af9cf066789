"""Drop-in for the array-level hot path of ``zopt.ilqrUtils`` on MI355X HIP kernels.

Same function names / arguments / return types as the reference (ilqrUtils.py); arrays may carry extra LEADING
batch axes.  NumPy in -> NumPy out; torch ROCm tensors in -> torch ROCm tensors out.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _arrays as arr
from . import _lib
from .pytrees import AffineDynamics, AffinePolicy, QuadraticCostFunction, QuadraticValueFunction  # noqa: F401

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _fields(t):
    """Fields of a (Named)tuple; the pytrees override __getitem__ (per-time-step slice), so use tuple.__iter__."""
    return list(tuple.__iter__(t)) if isinstance(t, tuple) else list(t)


def _shape(x):
    return tuple(x.shape) if hasattr(x, "shape") else tuple(np.shape(x))


def backwardPass_ilqr(dynamics, cost, Vf):
    """Backwards pass of the iLQR algorithm (reference ilqrUtils.py:176-181, step :153-173).

    Arguments
    ---------
        dynamics : AffineDynamics(f (..., N, n), f_x (..., N, n, n), f_u (..., N, n, m))   (f is unused, :156)
        cost : QuadraticCostFunction(c (..., N), c_x (..., N, n), c_u (..., N, m), c_xx, c_ux (..., N, m, n), c_uu)
        Vf : QuadraticValueFunction(v (...), v_x (..., n), v_xx (..., n, n)) terminal value function

    Returns
    -------
        AffinePolicy(l (..., N, m), L (..., N, m, n))
    """
    _, f_x, f_u = _fields(dynamics)[:3]
    c, c_x, c_u, c_xx, c_ux, c_uu = _fields(cost)
    v, v_x, v_xx = _fields(Vf)
    shp = _shape(f_u)
    if len(shp) < 3:
        raise ValueError("f_u must have shape (..., N, n, m)")
    lead, (N, n, m) = shp[:-3], shp[-3:]
    expect = {"f_x": (f_x, lead + (N, n, n)), "c_x": (c_x, lead + (N, n)), "c_u": (c_u, lead + (N, m)),
              "c_xx": (c_xx, lead + (N, n, n)), "c_ux": (c_ux, lead + (N, m, n)), "c_uu": (c_uu, lead + (N, m, m)),
              "v_x": (v_x, lead + (n,)), "v_xx": (v_xx, lead + (n, n))}
    for name, (X, s) in expect.items():
        if _shape(X) != s:
            raise ValueError(f"{name} has shape {_shape(X)}, expected {s}")
    template = f_x
    dt = torch.float64
    dev = [arr.to_device(X, dt) for X in (f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, v_x, v_xx)]
    batch = 1
    for d in lead:
        batch *= int(d)
    dl = torch.empty(lead + (N, m), dtype=dt, device=dev[0].device)
    dL = torch.empty(lead + (N, m, n), dtype=dt, device=dev[0].device)
    rc = _lib.lib().zm_ilqr_backward_f64(*[t.data_ptr() for t in dev], dl.data_ptr(), dL.data_ptr(), batch, N, n, m,
                                         ctypes.c_void_p(arr.stream_ptr(dev[0])))
    _lib.check(rc, "backwardPass_ilqr")
    fp32_in = (arr.is_torch(template) and template.dtype == torch.float32) or \
        (not arr.is_torch(template) and np.asarray(template).dtype == np.float32)
    if fp32_in:
        dl, dL = dl.to(torch.float32), dL.to(torch.float32)
    return AffinePolicy(arr.result_like(dl, template), arr.result_like(dL, template))
