"""Batched closed-loop simulation with time-indexed gains -- the consumer of the gains the solvers return.

The reference wires a controller block and a dynamics block into its generic `Simulator` (simulator.py:124-138,
`_internalStepFunDiscrete` / `_solve_ivp_discrete`) and uses it with the two control laws below; both are one forward
recursion `u_k = law(k, x_k)`, `x_{k+1} = f(x_k, u_k)`, which is exactly the rollout kernel (`zm_rollout_linesearch_f64`
with one step size and no feed-forward term).  Every leading batch axis is an independent simulation; the model is a
registered device model (`zopt_amd.models`), optionally with a constant wind (`QuadcopterEuler(dt, wind_ned=(3, 1, 0))`,
demos/iterativeLqr.py:48).
"""
from __future__ import annotations

import numpy as np

from . import _arrays as arr
from . import ilqrUtils
from .pytrees import AffinePolicy, Trajectory

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _zeros_like(ref, shape):
    if arr.is_torch(ref):
        return torch.zeros(shape, dtype=ref.dtype, device=ref.device)
    return np.zeros(shape, dtype=np.asarray(ref).dtype)


def simulateTrackingController(dynFun, x0, LArr, traj):
    """Closed loop with the iLQR / DDP tracking law `u_k = LArr[k] (x_k - xTraj[k]) + uTraj[k]`
    (demos/iterativeLqr.py:16-17, 44-56; demos/differentialDynamicProgramming.py likewise).

    Arguments
    ---------
        dynFun : registered device model, `x_{k+1} = dynFun(x_k, u_k)`
        x0 : initial state (..., n)
        LArr : feedback gains (..., N, m, n)     (as returned by iterativeLqr / differentialDynamicProgramming)
        traj : Trajectory(xTraj (..., N+1, n), uTraj (..., N, m)) the gains were computed about

    Returns
    -------
        Trajectory(xSim (..., N+1, n) incl. x0, uSim (..., N, m))
    """
    shp = tuple(LArr.shape)
    l = _zeros_like(LArr, shp[:-1])
    return ilqrUtils.trajectoryRollout(x0, dynFun, AffinePolicy(l, LArr), traj, alpha=1)


def simulateProportionalFeedback(dynFun, x0, K, xTrim, uTrim, N=None):
    """Closed loop with `u_k = -K[k] (x_k - xTrim) + uTrim` (lqrUtils.py:266-269 proportionalFeedbackController).

    Arguments
    ---------
        dynFun : registered device model
        x0 : initial state (..., n)
        K : LQR gains, time-indexed (..., N, m, n) (discreteFiniteHorizonLqr) or constant (..., m, n)
            (discreteInfiniteHorizonLqr; `N` = number of steps is then required)
        xTrim, uTrim : operating point (n,), (m,) or with the leading batch axes
        N : number of simulation steps (taken from K when it is time-indexed)

    Returns
    -------
        Trajectory(xSim (..., N+1, n), uSim (..., N, m))
    """
    lead = tuple(x0.shape[:-1]) if hasattr(x0, "shape") else np.shape(x0)[:-1]
    n, m = dynFun.n, dynFun.m
    kshape = tuple(K.shape)
    xp = torch if arr.is_torch(K) else np
    if len(kshape) == len(lead) + 2:          # constant gain
        if N is None:
            raise ValueError("N (number of steps) is required with a constant gain")
        Kt = xp.broadcast_to(K[..., None, :, :], lead + (int(N), m, n))
    elif len(kshape) == len(lead) + 3:
        Kt = K if N is None else K[..., :int(N), :, :]
        N = Kt.shape[-3]
    else:
        raise ValueError(f"K has shape {kshape}, expected {lead + ('N', m, n)} or {lead + (m, n)}")
    L = -Kt
    as_k = (lambda v: torch.as_tensor(v, dtype=K.dtype, device=K.device)) if arr.is_torch(K) else (lambda v: np.asarray(v, dtype=np.float64))
    xr = xp.broadcast_to(as_k(xTrim)[..., None, :], lead + (N + 1, n))
    ur = xp.broadcast_to(as_k(uTrim)[..., None, :], lead + (N, m))
    l = _zeros_like(K, lead + (N, m))
    mk = (lambda a: a.contiguous()) if arr.is_torch(K) else np.ascontiguousarray
    return ilqrUtils.trajectoryRollout(x0, dynFun, AffinePolicy(l, mk(L)), Trajectory(mk(xr), mk(ur)), alpha=1)
