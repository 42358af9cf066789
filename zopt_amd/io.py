"""Fixture / results interchange.  The reference has no on-disk format -- its plots (plottingTools.py:5-40,
mpcUtils.py:84-122) consume plain arrays -- so this is a thin, documented `.npz` schema for the three problem families and their
results, usable on machines without a GPU (NumPy only; nothing here computes).

    kind = "lqr"    A (..., T, n, n)  B (..., T, n, m)  Q (..., T, n, n)  R (..., T, m, m)            -> L (..., T, m, n)
    kind = "ilqr"   x0 (..., n)  uGuess (..., T, m)  Q, R, Qf  dt  model ("quadcopter" | "linear")      -> xTraj, uTraj, L, J, converged
    kind = "mpc"    A (n, n)  B (n, m)  Q  R  Qf  N  x_lb  x_ub  u_lb  u_ub  x0 (..., n)                -> xTraj, uTraj, status

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
