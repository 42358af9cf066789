"""Fixture / results interchange.  The reference has no on-disk format -- its plots (plottingTools.py:5-40,
mpcUtils.py:84-122) consume plain arrays -- so this is a thin, documented `.npz` schema for the three problem families and their
results, usable on machines without a GPU (NumPy only; nothing here computes).

    kind = "lqr"    A (..., T, n, n)  B (..., T, n, m)  Q (..., T, n, n)  R (..., T, m, m)            -> L (..., T, m, n)
    kind = "ilqr"   x0 (..., n)  uGuess (..., T, m)  Q, R, Qf  dt  model ("quadcopter" | "linear")      -> xTraj, uTraj, L, J, converged
    kind = "mpc"    A (n, n)  B (n, m)  Q  R  Qf  N  x_lb  x_ub  u_lb  u_ub  x0 (..., n)                -> xTraj, uTraj, status

Adapters onto the reference's plot functions (they take plain arrays; presentation itself stays in the reference):

    time_trajectory_inputs(results, dt, field, index)   -> (tArr (N,), xArr (N, nx))   for plottingTools.plotTimeTrajectory(tArr, xArr, ...)
    mpc_trajectory_array(xTraj_steps, index)            -> traj (N_t, N_mpc, n)        for mpcUtils.plotMpcTrajectory(traj, dt, ...)

Every file carries `schema` (this version), `kind`, the problem arrays, and -- when results are stored -- the result arrays under
the names above plus `tArr = arange(T+1) * dt` when `dt` is known (the time axis the reference's plotting helpers take).
"""
from __future__ import annotations

import numpy as np

SCHEMA = 1
_REQUIRED = {
    "lqr": ("A", "B", "Q", "R"),
    "ilqr": ("x0", "uGuess", "Q", "R", "Qf", "dt", "model"),
    "mpc": ("A", "B", "Q", "R", "N", "x_lb", "x_ub", "u_lb", "u_ub", "x0"),
}
_RESULTS = {"lqr": ("L",), "ilqr": ("xTraj", "uTraj", "L", "J", "converged"), "mpc": ("xTraj", "uTraj", "status")}


def _np(v):
    if hasattr(v, "detach"):          # torch tensor (any device)
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def save(path, kind, problem, results=None):
    """Write one problem (dict of arrays, see the module docstring) and optionally its results to `path` (.npz)."""
    if kind not in _REQUIRED:
        raise ValueError(f"kind must be one of {sorted(_REQUIRED)}")
    missing = [k for k in _REQUIRED[kind] if k not in problem]
    if missing:
        raise ValueError(f"{kind} problem lacks {missing}")
    out = {"schema": np.int64(SCHEMA), "kind": np.str_(kind)}
    out.update({k: _np(v) for k, v in problem.items()})
    if results is not None:
        unknown = [k for k in results if k not in _RESULTS[kind]]
        if unknown:
            raise ValueError(f"{kind} results do not have fields {unknown}")
        out.update({k: (_np(v).astype(np.str_) if k == "status" else _np(v)) for k, v in results.items()})
        if "dt" in problem and "xTraj" in results:
            out["tArr"] = np.arange(_np(results["xTraj"]).shape[-2]) * float(_np(problem["dt"]))
    np.savez_compressed(path, **out)
    return path


def load(path):
    """-> (kind, problem dict, results dict); refuses files of another schema version.  Uses np.load(allow_pickle=False)."""
    with np.load(path, allow_pickle=False) as z:
        if "schema" not in z or int(z["schema"]) != SCHEMA:
            raise ValueError(f"{path}: not a zopt_amd interchange file of schema {SCHEMA}")
        kind = str(z["kind"])
        data = {k: z[k] for k in z.files if k not in ("schema", "kind")}
    results = {k: data.pop(k) for k in list(data) if k in _RESULTS[kind] or k == "tArr"}
    missing = [k for k in _REQUIRED[kind] if k not in data]
    if missing:
        raise ValueError(f"{path}: {kind} problem lacks {missing}")
    return kind, data, results


def time_trajectory_inputs(results, dt, field="xTraj", index=None):
    """Arrays shaped for `zopt.plottingTools.plotTimeTrajectory(tArr, xArr, names, title)` (plottingTools.py:5-40: `tArr` (N,),
    `xArr` (N, nx), one subplot per column) from a solver result (`iterativeLqr` / `differentialDynamicProgramming` / `lqrMpc.solve`
    trajectories, NumPy or torch on any device, as returned or as loaded by `load`).  `results` is a dict or a Trajectory;
    `field` is "xTraj" ((..., T+1, n) -> N = T+1) or "uTraj" ((..., T, m) -> N = T); a batched result needs `index` (an int or a
    tuple over the leading axes)."""
    src = results[field] if isinstance(results, dict) else getattr(results, field)
    x = _np(src)
    if x.ndim < 2:
        raise ValueError(f"{field} must have shape (..., N, nx)")
    if x.ndim > 2:
        if index is None:
            raise ValueError(f"{field} has leading batch axes {x.shape[:-2]}: pass index=")
        x = x[index if isinstance(index, tuple) else (index,)]
        if x.ndim != 2:
            raise ValueError("index must address one trajectory")
    return np.arange(x.shape[0]) * float(dt), np.ascontiguousarray(x, dtype=np.float64)


def mpc_trajectory_array(xTraj_steps, index=None):
    """The array `zopt.mpcUtils.plotMpcTrajectory(traj, dt, names, title)` / `animateMpcTrajectory` take (mpcUtils.py:84-122):
    `traj[i]` = the MPC prediction at closed-loop step i, shape (N_t, N_mpc, n).  `xTraj_steps` is the sequence of `Trajectory.xTraj`
    returned by the receding-horizon loop's `lqrMpc.solve` calls (demos/lqrMpc.py:40-47 collects `xMpc[i]`), each (N_mpc, n) or
    batched (..., N_mpc, n) with `index` selecting the instance."""
    steps = [_np(x) for x in xTraj_steps]
    if not steps:
        raise ValueError("no MPC steps")
    if steps[0].ndim > 2:
        if index is None:
            raise ValueError(f"batched predictions {steps[0].shape[:-2]}: pass index=")
        steps = [x[index if isinstance(index, tuple) else (index,)] for x in steps]
    if any(x.ndim != 2 or x.shape != steps[0].shape for x in steps):
        raise ValueError("every step must hold one (N_mpc, n) prediction of the same shape")
    return np.ascontiguousarray(np.stack(steps), dtype=np.float64)
