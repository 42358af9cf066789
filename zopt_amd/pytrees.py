"""Array-container mirror of ``zopt.pytrees`` (reference pytrees.py:6-12, 58-69, 84-98, 129-136, 165-177, 207-213).

Same NamedTuple names, field order and per-time-step slicing ``obj[k]``; the fields are NumPy arrays or
torch-ROCm tensors (optionally with leading batch axes, time axis right after them).  The JAX autodiff
constructors (``from_function`` / ``from_trajectory``) of the reference are not part of the array-level hot path
and live in ``zopt_amd.models`` for registered device models instead.
"""
from __future__ import annotations

from typing import NamedTuple


def _slice(tup, k):
    return type(tup)(*[f[k] for f in tuple.__iter__(tup)])


class Trajectory(NamedTuple):
    """Trajectory tuple: (xTraj (N+1, n), uTraj (N, m))"""
    xTraj: object
    uTraj: object

    def __getitem__(self, k):
        return _slice(self, k)


class QuadraticValueFunction(NamedTuple):
    """(v, v_x, v_xx):  V(x) = v + v_x.T x + 0.5 x.T v_xx x"""
    v: object
    v_x: object
    v_xx: object

    def __call__(self, x):
        v, v_x, v_xx = tuple.__iter__(self)
        return v + v_x.T @ x + 0.5 * x.T @ v_xx @ x


class QuadraticCostFunction(NamedTuple):
    """(c, c_x, c_u, c_xx, c_ux, c_uu)"""
    c: object
    c_x: object
    c_u: object
    c_xx: object
    c_ux: object
    c_uu: object

    def __call__(self, x, u, k=None):
        c, c_x, c_u, c_xx, c_ux, c_uu = tuple.__iter__(self)
        if k is None and getattr(c, "ndim", 0) != 0:
            raise ValueError("Must specify index for multi-dimensional cost")
        if k is not None:
            return self[k](x, u)
        return c + c_x @ x + c_u @ u + 0.5 * (x.T @ c_xx @ x + 2 * u.T @ c_ux @ x + u.T @ c_uu @ u)

    def __getitem__(self, k):
        return _slice(self, k)


class AffineDynamics(NamedTuple):
    """(f, f_x, f_u):  xOut = f + f_x x + f_u u"""
    f: object
    f_x: object
    f_u: object

    def __call__(self, x, u, k=None):
        f, f_x, f_u = tuple.__iter__(self)
        if k is None and getattr(f, "ndim", 1) != 1:
            raise ValueError("Must specify index for multi-dimensional dynamics")
        if k is not None:
            return self[k](x, u)
        return f + f_x @ x + f_u @ u

    def __getitem__(self, k):
        return _slice(self, k)


class QuadraticDynamics(NamedTuple):
    """(f, f_x, f_u, f_xx, f_ux, f_uu)"""
    f: object
    f_x: object
    f_u: object
    f_xx: object
    f_ux: object
    f_uu: object

    def __call__(self, x, u, k=None):
        f, f_x, f_u, f_xx, f_ux, f_uu = tuple.__iter__(self)
        if k is None and getattr(f, "ndim", 1) != 1:
            raise ValueError("Must specify index for trajectories")
        if k is not None:
            return self[k](x, u)
        return f + f_x @ x + f_u @ u + 0.5 * (x.T @ f_xx @ x + 2 * u.T @ f_ux @ x + u.T @ f_uu @ u)

    def __getitem__(self, k):
        return _slice(self, k)


class AffinePolicy(NamedTuple):
    """(l, L):  u = alpha * l + L x"""
    l: object
    L: object

    def __call__(self, x, k=None, alpha=1):
        l, L = tuple.__iter__(self)
        if k is None and getattr(l, "ndim", 1) != 1:
            raise ValueError("Must specify index for multi-dimensional policy")
        if k is not None:
            return self[k](x, alpha=alpha)
        return alpha * l + L @ x

    def __getitem__(self, k):
        return _slice(self, k)
