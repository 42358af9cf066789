"""Drop-in for the array-level hot path of ``zopt.ilqrUtils`` on MI355X HIP kernels.

Same function names / arguments / return types as the reference (ilqrUtils.py); arrays may carry extra LEADING
batch axes.  NumPy in -> NumPy out; torch ROCm tensors in -> torch ROCm tensors out.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _arrays as arr
from . import _lib
from . import models as _models
from .pytrees import (AffineDynamics, AffinePolicy, QuadraticCostFunction, QuadraticDynamics,  # noqa: F401
                      QuadraticValueFunction, Trajectory)

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class _LazyGeneric:
    """zopt_amd.generic (torch.func producers in front of the HIP sweeps for callable models) imports torch unconditionally; it is
    loaded on first use so that `import zopt_amd.ilqrUtils` works on a host without torch, as the other modules do (calls raise)."""
    _mod = None

    @staticmethod
    def is_callable_model(f):
        return callable(f) and not hasattr(f, "c_struct")

    def __getattr__(self, name):
        if _LazyGeneric._mod is None:
            from . import generic
            _LazyGeneric._mod = generic
        return getattr(_LazyGeneric._mod, name)


_generic = _LazyGeneric()


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


def _ddp_large(n, m):
    """Shapes the one-tile DDP kernels (ilqr_backward.hip MODE 2, psd.hip's contracted-dynamics kernel) do not take."""
    return n > 12 or m > 4


def _project_vf_zz(v_x, f_xx, f_ux, f_uu, eps=1e-3):
    """conditionQuadraticDynamics for LARGE shapes (reference ilqrUtils.py:237-251) on flat batches: `vf_.. = einsum('i,ijk', v_x, f_..)`
    (a torch contraction on the GPU), the stacked `[[vf_xx, vf_ux^T],[vf_ux, vf_uu]]` PD-projected by the HIP kernel (zm_psd_project_f64:
    multi-tile matrix-sign iteration beyond 16 x 16), blocks returned.  Arguments (b, n), (b, n, n, n), (b, n, m, n), (b, n, m, m)."""
    n, m = f_ux.shape[-1], f_ux.shape[-2]
    vf_xx = torch.einsum("bi,bijk->bjk", v_x, f_xx)
    vf_ux = torch.einsum("bi,bijk->bjk", v_x, f_ux)
    vf_uu = torch.einsum("bi,bijk->bjk", v_x, f_uu)
    Z = torch.cat([torch.cat([vf_xx, vf_ux.transpose(-1, -2)], dim=-1), torch.cat([vf_ux, vf_uu], dim=-1)], dim=-2).contiguous()
    if Z.shape[0]:
        rc = _lib.lib().zm_psd_project_f64(Z.data_ptr(), Z.shape[0], n + m, float(eps), ctypes.c_void_p(arr.stream_ptr(Z)))
        _lib.check(rc, "conditionQuadraticDynamics")
    return Z[:, :n, :n], Z[:, n:, :n], Z[:, n:, n:]


def _riccati_value_call(f_x, f_u, c, c_x, c_u, c_xx, c_ux, c_uu, v, v_x, v_xx):
    """zm_riccati_value_f64 (iLQR form, one time step) on flat contiguous batches; returns (v', v_x', v_xx', l, L)."""
    b, n, m = f_u.shape
    dt, dev = f_x.dtype, f_x.device
    dl, dL = torch.empty((b, m), dtype=dt, device=dev), torch.empty((b, m, n), dtype=dt, device=dev)
    ov, ovx, ovxx = torch.empty(b, dtype=dt, device=dev), torch.empty((b, n), dtype=dt, device=dev), torch.empty((b, n, n), dtype=dt, device=dev)
    ts = [t.contiguous() for t in (f_x, f_u, c, c_x, c_u, c_xx, c_ux, c_uu, v, v_x, v_xx)]
    rc = _lib.lib().zm_riccati_value_f64(ts[0].data_ptr(), ts[1].data_ptr(), None, None, None, *[t.data_ptr() for t in ts[2:]],
                                         dl.data_ptr(), dL.data_ptr(), ov.data_ptr(), ovx.data_ptr(), ovxx.data_ptr(), b, 1, n, m,
                                         ctypes.c_void_p(arr.stream_ptr(f_x)))
    _lib.check(rc, "riccatiStep")
    return ov, ovx, ovxx, dl, dL


def _ddp_step_large(f_x, f_u, f_xx, f_ux, f_uu, c, c_x, c_u, c_xx, c_ux, c_uu, v, v_x, v_xx):
    """riccatiStep_ddp for large shapes (reference ilqrUtils.py:184-206): the iLQR step (HIP tile sweep, T = 1) with the cost Hessians
    augmented by the PD-projected second-order terms -- Q_xx = c_xx + f_x^T v_xx f_x + vf_xx etc. (:196-198)."""
    vf_xx, vf_ux, vf_uu = _project_vf_zz(v_x, f_xx, f_ux, f_uu)
    return _riccati_value_call(f_x, f_u, c, c_x, c_u, c_xx + vf_xx, c_ux + vf_ux, c_uu + vf_uu, v, v_x, v_xx)


def _riccati_step(dynamics, cost, value, ddp):
    """One backward step with the value function returned (zm_riccati_value_f64, T = 1)."""
    dyn = _fields(dynamics)
    f_x, f_u = dyn[1], dyn[2]
    c, c_x, c_u, c_xx, c_ux, c_uu = _fields(cost)
    v, v_x, v_xx = _fields(value)
    shp = _shape(f_u)
    if len(shp) < 2:
        raise ValueError("f_u must have shape (..., n, m)")
    lead, (n, m) = shp[:-2], shp[-2:]
    expect = {"f_x": (f_x, lead + (n, n)), "c": (c, lead), "c_x": (c_x, lead + (n,)), "c_u": (c_u, lead + (m,)),
              "c_xx": (c_xx, lead + (n, n)), "c_ux": (c_ux, lead + (m, n)), "c_uu": (c_uu, lead + (m, m)),
              "v": (v, lead), "v_x": (v_x, lead + (n,)), "v_xx": (v_xx, lead + (n, n))}
    if ddp:
        expect.update({"f_xx": (dyn[3], lead + (n, n, n)), "f_ux": (dyn[4], lead + (n, m, n)), "f_uu": (dyn[5], lead + (n, m, m))})
    for name, (X, s_) in expect.items():
        if _shape(X) != s_:
            raise ValueError(f"{name} has shape {_shape(X)}, expected {s_}")
    template = f_x
    dt = torch.float64
    T = lambda X: arr.to_device(X, dt).contiguous()
    df_x, df_u, dc, dc_x, dc_u, dc_xx, dc_ux, dc_uu, dv, dv_x, dv_xx = (T(X) for X in (f_x, f_u, c, c_x, c_u, c_xx, c_ux, c_uu, v,
                                                                                         v_x, v_xx))
    z = [T(X).data_ptr() for X in dyn[3:6]] if ddp else [None, None, None]
    zk = [T(X) for X in dyn[3:6]] if ddp else []          # keep alive
    if ddp:
        z = [t.data_ptr() for t in zk]
    batch = 1
    for d in lead:
        batch *= int(d)
    dev = df_x.device
    if ddp and _ddp_large(n, m):
        fl = lambda t, tail: t.reshape((batch,) + tail)
        ov, ovx, ovxx, dl, dL = _ddp_step_large(fl(df_x, (n, n)), fl(df_u, (n, m)), fl(zk[0], (n, n, n)), fl(zk[1], (n, m, n)),
                                                fl(zk[2], (n, m, m)), fl(dc, ()), fl(dc_x, (n,)), fl(dc_u, (m,)), fl(dc_xx, (n, n)),
                                                fl(dc_ux, (m, n)), fl(dc_uu, (m, m)), fl(dv, ()), fl(dv_x, (n,)), fl(dv_xx, (n, n)))
        outs = [arr.result_like(o.reshape(lead + tuple(o.shape[1:])), template) for o in (ov, ovx, ovxx, dl, dL)]
        if not arr.is_torch(template) and len(lead) == 0:
            outs[0] = float(outs[0])
        return QuadraticValueFunction(*outs[:3]), AffinePolicy(*outs[3:])
    dl = torch.empty(lead + (m,), dtype=dt, device=dev)
    dL = torch.empty(lead + (m, n), dtype=dt, device=dev)
    ov = torch.empty(lead, dtype=dt, device=dev)
    ovx = torch.empty(lead + (n,), dtype=dt, device=dev)
    ovxx = torch.empty(lead + (n, n), dtype=dt, device=dev)
    rc = _lib.lib().zm_riccati_value_f64(df_x.data_ptr(), df_u.data_ptr(), z[0], z[1], z[2], dc.data_ptr(), dc_x.data_ptr(),
                                         dc_u.data_ptr(), dc_xx.data_ptr(), dc_ux.data_ptr(), dc_uu.data_ptr(), dv.data_ptr(),
                                         dv_x.data_ptr(), dv_xx.data_ptr(), dl.data_ptr(), dL.data_ptr(), ov.data_ptr(),
                                         ovx.data_ptr(), ovxx.data_ptr(), batch, 1, n, m, ctypes.c_void_p(arr.stream_ptr(df_x)))
    _lib.check(rc, "riccatiStep")
    outs = [arr.result_like(o, template) for o in (ov, ovx, ovxx, dl, dL)]
    if not arr.is_torch(template) and len(lead) == 0:
        outs[0] = float(outs[0])
    return QuadraticValueFunction(*outs[:3]), AffinePolicy(*outs[3:])


def riccatiStep_ilqr(dynamics, cost, value):
    """One step of the iLQR Riccati recursion (reference ilqrUtils.py:153-173): AffineDynamics (f (n), f_x (n,n), f_u (n,m)),
    QuadraticCostFunction and QuadraticValueFunction of ONE time step (optionally with leading batch axes) ->
    (QuadraticValueFunction(v', v_x', v_xx'), AffinePolicy(l, L))."""
    return _riccati_step(dynamics, cost, value, ddp=False)


def riccatiStep_ddp(dynamics, cost, value):
    """One step of the DDP Riccati recursion (reference ilqrUtils.py:184-206): as riccatiStep_ilqr with QuadraticDynamics
    (f, f_x, f_u, f_xx (n,n,n), f_ux (n,m,n), f_uu (n,m,m)); the second-order terms are PD-projected (:237-251)."""
    return _riccati_step(dynamics, cost, value, ddp=True)


LINESEARCH_ALPHAS = 0.5 ** np.arange(16)   # reference ilqrUtils.py:145
_TRACE = None   # diagnostics: a list; while it is one, the fused drivers run zm_ilqr_solve_trace_f64 and append ("iterations", its),
#                 ("J_trace", (its, batch) costs after every iteration's acceptance), ("alpha_trace", (its, batch) winning step-size indices)


def _rollout(x0, dynFun, policy, trajPrev, alphas, costFun):
    """Shared driver of trajectoryRollout / forwardPass2 on the rollout_linesearch kernel."""
    if _generic.is_callable_model(dynFun):
        return _rollout_generic(x0, dynFun, policy, trajPrev, alphas, costFun)
    if not hasattr(dynFun, "c_struct"):
        raise TypeError("dynFun must be a registered device model (zopt_amd.models.LinearModel / QuadcopterEuler) or a torch "
                        "callable f(x, u) -> x+ (generic path, zopt_amd/generic.py)")
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


def _torch_costs(runningCost, terminalCost):
    """torch callables (x, u) -> c, (x) -> cf from what the caller passed: torch callables as they are; a registered QuadraticCost (or
    its bound methods) as x'Qx + u'Ru, x'Qf x on device copies of its matrices"""
    for c in (runningCost, getattr(runningCost, "__self__", None)):
        if isinstance(c, _models.QuadraticCost):
            Q, R, Qf = (arr.to_device(M, torch.float64) for M in (c.Q, c.R, c.Qf))
            return (lambda x, u: x @ Q @ x + u @ R @ u), (lambda x: x @ Qf @ x)
    if not callable(runningCost) or not callable(terminalCost):
        raise TypeError("runningCost / terminalCost must be torch callables or a registered zopt_amd.models.QuadraticCost")
    return runningCost, terminalCost


def _rollout_generic(x0, dynFun, policy, trajPrev, alphas, costFun):
    """trajectoryRollout / forwardPass2 for a torch callable `dynFun(x, u) -> x+` (zopt_amd/generic.py); `costFun`: None, a
    registered QuadraticCost, or an object with torch callables `.runningCost(x, u)` / `.terminalCost(x)` (CostFunction of the
    reference, pytrees.py:27-38)."""
    l, L = _fields(policy)
    xPrev, uPrev = _fields(trajPrev)
    shp = _shape(L)
    lead, (N, m, n) = shp[:-3], shp[-3:]
    dt = torch.float64
    dx0, dl, dL, dxp, dup = (arr.to_device(X, dt) for X in (x0, l, L, xPrev, uPrev))
    dx0, dl, dL = dx0.reshape(-1, n), dl.reshape(-1, N, m), dL.reshape(-1, N, m, n)
    dxp, dup = dxp.reshape(-1, N + 1, n), dup.reshape(-1, N, m)
    if costFun is None:
        al = arr.to_device(np.asarray(alphas, dtype=np.float64), dt)
        xs, us, _ = _generic.rollout(dx0, dynFun, dl, dL, dxp, dup, al)
        xT, uT, J = xs[:, 0], us[:, 0], None
    else:
        rc, tc = _torch_costs(getattr(costFun, "runningCost", costFun), getattr(costFun, "terminalCost", costFun))
        xT, uT, J = _generic.forward_pass2(dx0, dynFun, rc, tc, dl, dL, dxp, dup)
        J = arr.result_like(J.reshape(lead), L)
    traj = Trajectory(arr.result_like(xT.reshape(lead + (N + 1, n)), L), arr.result_like(uT.reshape(lead + (N, m)), L))
    return traj, J


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


def ensurePositiveDefinite(a, eps=1e-3):
    """`w, v = eigh(a); (v * max(w, eps)) @ v.T` (reference ilqrUtils.py:217-219; eigh symmetrises its input).
    `a` is (..., k, k) with k <= 64 (one MFMA tile up to 16, NT x NT tiles beyond: psd_tiled.hip); returns a new array."""
    shp = _shape(a)
    if len(shp) < 2 or shp[-1] != shp[-2]:
        raise ValueError("a must be (..., k, k)")
    k = shp[-1]
    d = arr.to_device(a, torch.float64).clone()
    count = 1
    for s_ in shp[:-2]:
        count *= int(s_)
    rc = _lib.lib().zm_psd_project_f64(d.data_ptr(), count, k, float(eps), ctypes.c_void_p(arr.stream_ptr(d)))
    _lib.check(rc, "ensurePositiveDefinite")
    return arr.result_like(d, a)


def conditionQuadraticCost(quadratic_cost):
    """Ensure quadratic cost is strictly positive definite (reference ilqrUtils.py:222-234): project the stacked
    Hessian [[c_xx, c_ux^T],[c_ux, c_uu]] of every time step and slice it back."""
    c, c_x, c_u, c_xx, c_ux, c_uu = _fields(quadratic_cost)
    shp = _shape(c_ux)
    m, n = shp[-2:]
    lead = shp[:-2]
    if _shape(c_xx) != lead + (n, n) or _shape(c_uu) != lead + (m, m):
        raise ValueError("inconsistent cost Hessian shapes")
    dxx, dux, duu = (arr.to_device(X, torch.float64).clone() for X in (c_xx, c_ux, c_uu))
    count = 1
    for s_ in lead:
        count *= int(s_)
    rc = _lib.lib().zm_condition_cost_f64(dxx.data_ptr(), dux.data_ptr(), duu.data_ptr(), count, n, m, 1e-3,
                                          ctypes.c_void_p(arr.stream_ptr(dxx)))
    _lib.check(rc, "conditionQuadraticCost")
    return QuadraticCostFunction(c, c_x, c_u, arr.result_like(dxx, c_xx), arr.result_like(dux, c_ux),
                                 arr.result_like(duu, c_uu))


def conditionValueFunction(Vf):
    """reference ilqrUtils.py:254-257."""
    v, v_x, v_xx = _fields(Vf)
    return QuadraticValueFunction(v, v_x, ensurePositiveDefinite(v_xx))


def conditionQuadraticDynamics(quadratic_dynamics, v_x):
    """Ensure the quadratic terms of the DDP riccati step are positive definite (reference ilqrUtils.py:237-251):
    `vf_.. = einsum('i,ijk', v_x, f_..)`, the stacked `[[vf_xx, vf_ux^T],[vf_ux, vf_uu]]` PD-projected, blocks returned as
    `(vf_xx (..., n, n), vf_ux (..., m, n), vf_uu (..., m, m))`.  One time step or any leading batch / time axes."""
    dyn = _fields(quadratic_dynamics)
    f_xx, f_ux, f_uu = dyn[3], dyn[4], dyn[5]
    shp = _shape(f_ux)
    if len(shp) < 3:
        raise ValueError("f_ux must have shape (..., n, m, n)")
    lead, (n, m, n2) = shp[:-3], shp[-3:]
    if n2 != n or _shape(f_xx) != lead + (n, n, n) or _shape(f_uu) != lead + (n, m, m) or _shape(v_x) != lead + (n,):
        raise ValueError("conditionQuadraticDynamics: inconsistent shapes")
    dt = torch.float64
    dxx, dux, duu, dvx = (arr.to_device(X, dt).contiguous() for X in (f_xx, f_ux, f_uu, v_x))
    count = 1
    for d in lead:
        count *= int(d)
    if n + m > 16:      # beyond the one-tile kernel: torch contraction + the multi-tile HIP projection
        oxx, oux, ouu = _project_vf_zz(dvx.reshape(count, n), dxx.reshape(count, n, n, n), dux.reshape(count, n, m, n),
                                       duu.reshape(count, n, m, m))
        return tuple(arr.result_like(o.reshape(lead + tuple(o.shape[1:])).contiguous(), f_xx) for o in (oxx, oux, ouu))
    oxx = torch.empty(lead + (n, n), dtype=dt, device=dxx.device)
    oux = torch.empty(lead + (m, n), dtype=dt, device=dxx.device)
    ouu = torch.empty(lead + (m, m), dtype=dt, device=dxx.device)
    rc = _lib.lib().zm_condition_dynamics_f64(dxx.data_ptr(), dux.data_ptr(), duu.data_ptr(), dvx.data_ptr(), oxx.data_ptr(),
                                              oux.data_ptr(), ouu.data_ptr(), count, n, m, 1e-3,
                                              ctypes.c_void_p(arr.stream_ptr(dxx)))
    _lib.check(rc, "conditionQuadraticDynamics")
    return tuple(arr.result_like(o, f_xx) for o in (oxx, oux, ouu))


def _registered_cost(runningCost, terminalCost):
    """Accepts a zopt_amd.models.QuadraticCost handle (for both arguments) or its bound methods."""
    for c in (runningCost, getattr(runningCost, "__self__", None)):
        if isinstance(c, _models.QuadraticCost):
            other = terminalCost if isinstance(terminalCost, _models.QuadraticCost) else getattr(terminalCost, "__self__", None)
            if other is not c:
                raise TypeError("runningCost and terminalCost must come from the same QuadraticCost")
            return c
    raise TypeError("runningCost / terminalCost must be a registered zopt_amd.models.QuadraticCost (or its "
                    ".runningCost / .terminalCost methods); arbitrary Python callables cannot run inside a HIP kernel")


def backwardPass_ddp(dynamics, cost, Vf):
    """Backwards pass of the DDP algorithm (reference ilqrUtils.py:209-214, step :184-206, :237-251).

    Arguments
    ---------
        dynamics : QuadraticDynamics(f, f_x (..., N, n, n), f_u (..., N, n, m), f_xx (..., N, n, n, n),
                   f_ux (..., N, n, m, n), f_uu (..., N, n, m, m))
        cost : QuadraticCostFunction, Vf : QuadraticValueFunction   (as backwardPass_ilqr)

    Returns
    -------
        AffinePolicy(l (..., N, m), L (..., N, m, n))
    """
    _, f_x, f_u, f_xx, f_ux, f_uu = _fields(dynamics)
    c, c_x, c_u, c_xx, c_ux, c_uu = _fields(cost)
    v, v_x, v_xx = _fields(Vf)
    shp = _shape(f_u)
    if len(shp) < 3:
        raise ValueError("f_u must have shape (..., N, n, m)")
    lead, (N, n, m) = shp[:-3], shp[-3:]
    expect = {"f_x": (f_x, lead + (N, n, n)), "f_xx": (f_xx, lead + (N, n, n, n)), "f_ux": (f_ux, lead + (N, n, m, n)),
              "f_uu": (f_uu, lead + (N, n, m, m)), "c_x": (c_x, lead + (N, n)), "c_u": (c_u, lead + (N, m)),
              "c_xx": (c_xx, lead + (N, n, n)), "c_ux": (c_ux, lead + (N, m, n)), "c_uu": (c_uu, lead + (N, m, m)),
              "v_x": (v_x, lead + (n,)), "v_xx": (v_xx, lead + (n, n))}
    for name, (X, s_) in expect.items():
        if _shape(X) != s_:
            raise ValueError(f"{name} has shape {_shape(X)}, expected {s_}")
    dt = torch.float64
    dev = [arr.to_device(X, dt) for X in (f_x, f_u, f_xx, f_ux, f_uu, c_x, c_u, c_xx, c_ux, c_uu, v_x, v_xx)]
    batch = 1
    for d in lead:
        batch *= int(d)
    dl = torch.empty(lead + (N, m), dtype=dt, device=dev[0].device)
    dL = torch.empty(lead + (N, m, n), dtype=dt, device=dev[0].device)
    if _ddp_large(n, m):
        # Beyond the one-tile DDP sweep (n <= 12, m <= 4) the recursion runs step by step from the host: per step a torch contraction,
        # the HIP PD projection of the stacked (n+m)^2 matrix and the HIP tile sweep for one step (three launches per time step:
        # a coverage path for the shapes the generic-callable drivers take, not a fast one).
        if n > 48 or m > 16:
            raise ValueError(f"backwardPass_ddp: (n={n}, m={m}) not covered (need n <= 48, m <= 16)")
        fx, fu, fxx, fux, fuu, cx, cu, cxx, cux, cuu, vx, vxx = (t.reshape((batch,) + tuple(t.shape[len(lead):])) for t in dev)
        vv = torch.zeros(batch, dtype=dt, device=vx.device)
        zc = torch.zeros(batch, dtype=dt, device=vx.device)
        fl, fL = dl.reshape(batch, N, m), dL.reshape(batch, N, m, n)
        for k in range(N - 1, -1, -1):
            vv, vx, vxx, lk, Lk = _ddp_step_large(fx[:, k], fu[:, k], fxx[:, k].contiguous(), fux[:, k].contiguous(), fuu[:, k].contiguous(),
                                                  zc, cx[:, k], cu[:, k], cxx[:, k], cux[:, k], cuu[:, k], vv, vx, vxx)
            fl[:, k], fL[:, k] = lk, Lk
        return AffinePolicy(arr.result_like(dl, f_x), arr.result_like(dL, f_x))
    rc = _lib.lib().zm_ddp_backward_f64(*[t.data_ptr() for t in dev], None, 0, dl.data_ptr(), dL.data_ptr(), batch, N, n, m,
                                        ctypes.c_void_p(arr.stream_ptr(dev[0])))
    _lib.check(rc, "backwardPass_ddp")
    return AffinePolicy(arr.result_like(dl, f_x), arr.result_like(dL, f_x))


def iterativeLqr(dynamics, runningCost, terminalCost, x0, uGuess, maxIter=100, tol=1e-3):
    """Iterative LQR algorithm (reference ilqrUtils.py:260-327), batched and device-resident.

    Arguments
    ---------
        dynamics : registered discrete model `xOut = f(x,u)` (zopt_amd.models.LinearModel / QuadcopterEuler)
        runningCost, terminalCost : a registered zopt_amd.models.QuadraticCost (the handle, or its two methods)
        x0 : (..., n) initial state(s)
        uGuess : (..., N, m) initial guess for the control trajectory
        maxIter : maximum number of iLQR iterations
        tol : convergence tolerance `abs(J_prev - J) <= tol` (per trajectory)

    Returns
    -------
        traj : Trajectory(xTraj (..., N+1, n), uTraj (..., N, m))
        L : (..., N, m, n) feedback gains: `u[k] = L[k] @ (x[k]-xTraj[k]) + uTraj[k]`
        J : (...) cost
        converged : (...) bool

    Per iteration (ilqrUtils.py:305-322): linearise along the trajectory, expand the cost, PD-condition the cost and
    terminal Hessians (they are trajectory-independent for a quadratic cost: done once, shared), backward pass,
    16-way line-search rollout, `converged = |J - J_new| <= tol`.  Converged trajectories drop out of later iterations
    (the reference under vmap would run every lane to the slowest).
    """
    return _ilqr_or_ddp(dynamics, runningCost, terminalCost, x0, uGuess, maxIter, tol, ddp=False)


def differentialDynamicProgramming(dynamics, runningCost, terminalCost, x0, uGuess, maxIter=100, tol=1e-3):
    """Differential dynamic programming algorithm (reference ilqrUtils.py:330-397): `iterativeLqr` with the second-order
    expansion of the dynamics (QuadraticDynamics.from_trajectory, :365) and `backwardPass_ddp` (:373).  Same arguments
    and return values as `iterativeLqr`."""
    return _ilqr_or_ddp(dynamics, runningCost, terminalCost, x0, uGuess, maxIter, tol, ddp=True)


def _ilqr_or_ddp_generic(dynamics, runningCost, terminalCost, x0, uGuess, maxIter, tol, ddp):
    """The drivers for a torch callable `dynamics(x, u) -> x+` (and torch callables or a registered QuadraticCost for the costs):
    zopt_amd/generic.py produces the expansions and the line-search rollouts with torch.func on the GPU, the sweeps and the PD
    projections are the HIP kernels."""
    rc, tc = _torch_costs(runningCost, terminalCost)
    shp = _shape(uGuess)
    lead, (N, m) = shp[:-2], shp[-2:]
    n = _shape(x0)[-1]
    if _shape(x0) != lead + (n,):
        raise ValueError(f"x0 {_shape(x0)} / uGuess {shp} do not match")
    dt = torch.float64
    dx0 = arr.to_device(x0, dt).reshape(-1, n).contiguous()
    dug = arr.to_device(uGuess, dt).reshape(-1, N, m).contiguous()
    xT, uT, L, J, cv = _generic.solve(dynamics, rc, tc, dx0, dug, maxIter, tol, ddp)
    tmpl = uGuess
    xo, uo, Lo, Jo = (arr.result_like(o, tmpl) for o in (xT.reshape(lead + (N + 1, n)), uT.reshape(lead + (N, m)),
                                                      L.reshape(lead + (N, m, n)), J.reshape(lead)))
    co = arr.result_like(cv.reshape(lead), tmpl)
    if not arr.is_torch(tmpl) and len(lead) == 0:
        Jo, co = float(Jo), bool(co)
    return Trajectory(xo, uo), Lo, Jo, co


def _ilqr_or_ddp(dynamics, runningCost, terminalCost, x0, uGuess, maxIter, tol, ddp):
    model = dynamics
    if _generic.is_callable_model(model):
        return _ilqr_or_ddp_generic(dynamics, runningCost, terminalCost, x0, uGuess, maxIter, tol, ddp)
    if not hasattr(model, "c_struct"):
        raise TypeError("dynamics must be a registered device model (zopt_amd.models.*) or a torch callable f(x, u) -> x+")
    cost = _registered_cost(runningCost, terminalCost)
    n, m = model.n, model.m
    shp = _shape(uGuess)
    lead, (N, mm) = shp[:-2], shp[-2:]
    if mm != m or _shape(x0) != lead + (n,):
        raise ValueError(f"x0 {_shape(x0)} / uGuess {shp} do not match the model (n={n}, m={m})")
    dt = torch.float64
    lib = _lib.lib()
    dx0 = arr.to_device(x0, dt).reshape(-1, n).contiguous()
    dug = arr.to_device(uGuess, dt).reshape(-1, N, m).contiguous()
    dev = dx0.device
    B = dx0.shape[0]
    st = ctypes.c_void_p(arr.stream_ptr(dx0))
    md, cs = model.c_struct(), cost.c_struct()
    pmd, pcs = ctypes.addressof(md), ctypes.addressof(cs)
    L = torch.empty((B, N, m, n), dtype=dt, device=dev)
    xT = torch.empty((B, N + 1, n), dtype=dt, device=dev)
    uT = torch.empty((B, N, m), dtype=dt, device=dev)
    J = torch.empty(B, dtype=dt, device=dev)
    converged = torch.zeros(B, dtype=torch.int32, device=dev)
    if B:
        # The whole loop -- initial rollout (:293-298), per iteration expansions, PD-conditioned Hessians, backward pass, 16-way
        # line search, `converged = |J - J_new| <= tol` (:305-322) over the compacted list of still-active trajectories -- is ONE
        # call into the C ABI: the host side of the iteration runs in C++ (zopt_amd/csrc/ilqr_solve.hip), no Python between the
        # launches.  SYNC: host synchronisation (termination test + rebuild of the id list) every SYNC iterations; the results do
        # not depend on it (an iteration over a converged trajectory touches nothing).
        SYNC = max(1, int(os.environ.get("ZOPT_AMD_ILQR_SYNC", "4")))
        nws = int(lib.zm_ilqr_solve_workspace_f64(pmd, B, N, 1 if ddp else 0))
        if nws < 0:
            raise ValueError("iterativeLqr: cannot size the workspace for this model")
        ws = torch.empty(nws, dtype=dt, device=dev)
        iwork = torch.empty(2 * B + 2, dtype=torch.int32, device=dev)
        its = ctypes.c_int32(0)
        common = (pmd, pcs, dx0.data_ptr(), dug.data_ptr(), 1 if ddp else 0, int(maxIter), float(tol), SYNC,
                  ws.data_ptr(), nws, iwork.data_ptr(), xT.data_ptr(), uT.data_ptr(), L.data_ptr(), J.data_ptr(),
                  converged.data_ptr(), ctypes.addressof(its), B, N, st)
        if _TRACE is None:
            rc = lib.zm_ilqr_solve_f64(*common)
        else:
            Jtr = torch.full((max(int(maxIter), 1), B), float("nan"), dtype=dt, device=dev)
            atr = torch.full((max(int(maxIter), 1), B), -1, dtype=torch.int32, device=dev)
            rc = lib.zm_ilqr_solve_trace_f64(*common, Jtr.data_ptr(), atr.data_ptr())
        _lib.check(rc, "differentialDynamicProgramming" if ddp else "iterativeLqr")
        if _TRACE is not None:
            _TRACE.append(("iterations", int(its.value)))
            _TRACE.append(("J_trace", Jtr[: int(its.value)].cpu().numpy()))
            _TRACE.append(("alpha_trace", atr[: int(its.value)].cpu().numpy()))
    tmpl = uGuess
    fp32_in = (arr.is_torch(tmpl) and tmpl.dtype == torch.float32) or \
        (not arr.is_torch(tmpl) and np.asarray(tmpl).dtype == np.float32)
    outs = [xT.reshape(lead + (N + 1, n)), uT.reshape(lead + (N, m)), L.reshape(lead + (N, m, n)), J.reshape(lead)]
    if fp32_in:
        outs = [o.to(torch.float32) for o in outs]
    xo, uo, Lo, Jo = (arr.result_like(o, tmpl) for o in outs)
    co = arr.result_like(converged.bool().reshape(lead), tmpl)
    if not arr.is_torch(tmpl) and len(lead) == 0:
        Jo, co = float(Jo), bool(co)
    return Trajectory(xo, uo), Lo, Jo, co
