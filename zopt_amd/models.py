"""Registered device models and costs (host-side handles for the structs in include/zopt_amd.h).

The reference takes arbitrary Python callables (`dynamics`, `runningCost`, `terminalCost`) and lets JAX trace and
differentiate them (ilqrUtils.py:260-269).  A HIP kernel needs the model in device code, so the fused paths take
one of these handles wherever the reference takes a callable; the handles are also plain callables on NumPy arrays
(same math), so they can be passed to code that expects `f(x, u)`.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _arrays as arr

ZM_MODEL_LINEAR = 1
ZM_MODEL_QUADCOPTER = 2


class zm_model_t(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("n", ctypes.c_int), ("m", ctypes.c_int), ("reserved", ctypes.c_int),
                ("dt", ctypes.c_double), ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("wind_ned", ctypes.c_double * 3)]


class zm_quadcost_t(ctypes.Structure):
    _fields_ = [("Q", ctypes.c_void_p), ("R", ctypes.c_void_p), ("Qf", ctypes.c_void_p)]


class LinearModel:
    """x+ = A x + B u (time-invariant), e.g. the LQ problem of the reference's iLQR test (tests/test_ilqrUtils.py:167-196)."""

    def __init__(self, A, B):
        self.A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
        self.B = np.ascontiguousarray(np.asarray(B, dtype=np.float64))
        self.n, self.m = self.B.shape
        if self.A.shape != (self.n, self.n):
            raise ValueError("A must be (n,n) and B (n,m)")
        self._dev = None

    def __call__(self, x, u):
        return self.A @ x + self.B @ u

    def c_struct(self):
        import torch
        if self._dev is None:
            self._dev = (arr.to_device(self.A, torch.float64), arr.to_device(self.B, torch.float64))
        return zm_model_t(ZM_MODEL_LINEAR, self.n, self.m, 0, 0.0, self._dev[0].data_ptr(), self._dev[1].data_ptr())


class QuadcopterEuler:
    """x+ = x + dt * Quadcopter.inertialDynamics(x, u)  (reference quadcopter.py:116-144; demos/iterativeLqr.py:35)."""
    n, m = 12, 4
    uTrim = np.array([9.807, 0.0, 0.0, 0.0])   # hover: thrust = g (quadcopter.py:15, tests/test_quadcopter.py:55-58)

    def __init__(self, dt: float = 0.1, wind_ned=(0.0, 0.0, 0.0)):
        """dt: Euler step; wind_ned: constant wind in the NED frame (quadcopter.py:117; demos/iterativeLqr.py:48)"""
        self.dt = float(dt)
        self.wind_ned = tuple(float(w) for w in wind_ned)
        if len(self.wind_ned) != 3:
            raise ValueError("wind_ned must have 3 components")

    def c_struct(self):
        return zm_model_t(ZM_MODEL_QUADCOPTER, 12, 4, 0, self.dt, None, None, (ctypes.c_double * 3)(*self.wind_ned))


class QuadraticCost:
    """runningCost(x,u) = x'Qx + u'Ru, terminalCost(x) = x'Qf x  (demos/iterativeLqr.py:12-13,37; no 1/2)."""

    def __init__(self, Q, R, Qf=None):
        self.Q = np.ascontiguousarray(np.asarray(Q, dtype=np.float64))
        self.R = np.ascontiguousarray(np.asarray(R, dtype=np.float64))
        self.Qf = self.Q if Qf is None else np.ascontiguousarray(np.asarray(Qf, dtype=np.float64))
        self.n, self.m = self.Q.shape[0], self.R.shape[0]
        if self.Q.shape != (self.n, self.n) or self.R.shape != (self.m, self.m) or self.Qf.shape != (self.n, self.n):
            raise ValueError("QuadraticCost: Q (n,n), R (m,m), Qf (n,n) expected")
        self._dev = None

    def __call__(self, traj, k=None):
        """Cost of a trajectory, J = terminalCost(x_N) + sum_k runningCost(x_k, u_k) -- CostFunction.__call__ of the reference
        (pytrees.py:40-55); with `k`, the running cost at step k only.  Leading batch axes allowed; evaluated on the GPU."""
        from .pytrees import _expand
        c, _, _, _, _, _ = _expand("cost", self, traj[0] if not hasattr(traj, "xTraj") else traj.xTraj,
                                   traj[1] if not hasattr(traj, "uTraj") else traj.uTraj)
        if k is not None:
            return c[..., k]
        xT = traj.xTraj if hasattr(traj, "xTraj") else traj[0]
        v, _, _ = _expand("terminal", self, xT[..., -2:, :], None)
        J = c.sum(-1) + v[..., 0]
        return float(J) if (not hasattr(J, "device") and getattr(J, "ndim", 1) == 0) else J

    def runningCost(self, x, u):
        return x @ self.Q @ x + u @ self.R @ u

    def terminalCost(self, x):
        return x @ self.Qf @ x

    def c_struct(self):
        import torch
        if self._dev is None:
            self._dev = tuple(arr.to_device(X, torch.float64) for X in (self.Q, self.R, self.Qf))
        return zm_quadcost_t(*[t.data_ptr() for t in self._dev])
