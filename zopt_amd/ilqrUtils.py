"""Drop-in for the array-level hot path of ``zopt.ilqrUtils`` on MI355X HIP kernels.

Same function names / arguments / return types as the reference (ilqrUtils.py); arrays may carry extra LEADING
batch axes.  NumPy in -> NumPy out; torch ROCm tensors in -> torch ROCm tensors out.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _arrays as arr
from . import _lib
from . import models as _models
from .pytrees import (AffineDynamics, AffinePolicy, QuadraticCostFunction, QuadraticValueFunction,  # noqa: F401
                      Trajectory)

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


LINESEARCH_ALPHAS = 0.5 ** np.arange(16)   # reference ilqrUtils.py:145


def _rollout(x0, dynFun, policy, trajPrev, alphas, costFun):
    """Shared driver of trajectoryRollout / forwardPass2 on the rollout_linesearch kernel."""
    if not hasattr(dynFun, "c_struct"):
        raise TypeError("dynFun must be a registered device model (zopt_amd.models.LinearModel / QuadcopterEuler); "
                        "arbitrary Python callables cannot run inside a HIP kernel")
    if costFun is not None and not hasattr(costFun, "c_struct"):
        raise TypeError("costFun must be a registered zopt_amd.models.QuadraticCost")
    l, L = _fields(policy)
    xPrev, uPrev = _fields(trajPrev)
    n, m = dynFun.n, dynFun.m
    shp = _shape(L)
    lead, (N, mm, nn) = shp[:-3], shp[-3:]
    if (mm, nn) != (m, n):
        raise ValueError(f"policy.L has shape {shp}, model is (n={n}, m={m})")
    for name, X, s in (("x0", x0, lead + (n,)), ("policy.l", l, lead + (N, m)), ("xPrev", xPrev, lead + (N + 1, n)),
                       ("uPrev", uPrev, lead + (N, m))):
        if _shape(X) != s:
            raise ValueError(f"{name} has shape {_shape(X)}, expected {s}")
    dt = torch.float64
    dx0, dl, dL, dxp, dup = (arr.to_device(X, dt) for X in (x0, l, L, xPrev, uPrev))
    dal = arr.to_device(np.asarray(alphas, dtype=np.float64), dt)
    batch = 1
    for d in lead:
        batch *= int(d)
    xT = torch.empty(lead + (N + 1, n), dtype=dt, device=dx0.device)
    uT = torch.empty(lead + (N, m), dtype=dt, device=dx0.device)
    J = torch.empty(lead, dtype=dt, device=dx0.device) if costFun is not None else None
    md = dynFun.c_struct()
    cs = costFun.c_struct() if costFun is not None else None
    rc = _lib.lib().zm_rollout_linesearch_f64(
        ctypes.addressof(md), ctypes.addressof(cs) if cs is not None else None, dx0.data_ptr(), dl.data_ptr(),
        dL.data_ptr(), dxp.data_ptr(), dup.data_ptr(), dal.data_ptr(), int(dal.numel()), None, xT.data_ptr(),
        uT.data_ptr(), J.data_ptr() if J is not None else None, None, batch, N, ctypes.c_void_p(arr.stream_ptr(dx0)))
    _lib.check(rc, "rollout")
    traj = Trajectory(arr.result_like(xT, L), arr.result_like(uT, L))
    return traj, (arr.result_like(J, L) if J is not None else None)


def trajectoryRollout(x0, dynFun, policy, trajPrev, alpha=1):
    """Rollout a trajectory from the initial state using the provided control policy (reference ilqrUtils.py:33-66):
    `u_k = alpha*l_k + L_k (x_k - xPrev_k) + uPrev_k`, `x_{k+1} = dynFun(x_k, u_k)`; returns Trajectory(xTraj (N+1,n)
    incl. x0, uTraj (N,m)).  `dynFun` is a registered device model."""
    traj, _ = _rollout(x0, dynFun, policy, trajPrev, [float(alpha)], None)
    return traj


def forwardPass2(x0, dynFun, costFun, policy, trajPrev):
    """Simplified iLQR forward pass (reference ilqrUtils.py:116-150): roll out the 16 step sizes 0.5**j and return the
    trajectory of minimum cost and that cost.  `costFun` is a registered QuadraticCost."""
    traj, J = _rollout(x0, dynFun, policy, trajPrev, LINESEARCH_ALPHAS, costFun)
    if not arr.is_torch(J) and np.ndim(J) == 0:
        J = float(J)
    return traj, J
