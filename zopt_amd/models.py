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
ZM_MODEL_QUADCOPTER_RB = 3


class zm_model_t(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("n", ctypes.c_int), ("m", ctypes.c_int), ("reserved", ctypes.c_int),
                ("dt", ctypes.c_double), ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("wind_ned", ctypes.c_double * 3)]


class zm_quadcost_t(ctypes.Structure):
    _fields_ = [("Q", ctypes.c_void_p), ("R", ctypes.c_void_p), ("Qf", ctypes.c_void_p), ("diagonal", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


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
        st = zm_model_t(ZM_MODEL_LINEAR, self.n, self.m, 0, 0.0, self._dev[0].data_ptr(), self._dev[1].data_ptr())
        st._owner = self      # the struct holds raw device addresses: keep the arrays behind them alive with it
        return st


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
        isdiag = lambda M: not np.any(M - np.diag(np.diagonal(M)))          # exact: also False for NaN off the diagonal
        diag = int(isdiag(self.Q) and isdiag(self.R) and isdiag(self.Qf))
        st = zm_quadcost_t(*[t.data_ptr() for t in self._dev], diag, 0)
        st._owner = self      # the struct holds raw device addresses: keep the arrays behind them alive with it
        return st


class QuadcopterRigidBody:
    """The 8-state model the reference trims and linearises: `rigidBodyDynamics(state, control, wind_body)`
    (quadcopter.py:70-113), state [u,v,w,p,q,r,phi,theta].  dt = 0: the continuous derivative; dt > 0: one Euler step."""
    n, m = 8, 4

    def __init__(self, dt: float = 0.0, wind_body=(0.0, 0.0, 0.0)):
        self.dt = float(dt)
        self.wind_body = tuple(float(w) for w in wind_body)
        if len(self.wind_body) != 3:
            raise ValueError("wind_body must have 3 components")

    def c_struct(self):
        return zm_model_t(ZM_MODEL_QUADCOPTER_RB, 8, 4, 0, self.dt, None, None, (ctypes.c_double * 3)(*self.wind_body))


class Quadcopter:
    """Mirror of `zopt.quadcopter.Quadcopter` (quadcopter.py:10-201) on the device kernels: same method names and argument order,
    every array may carry leading batch axes (a family of operating points in one call)."""
    g, m, I = 9.807, 2.5, np.eye(3)         # quadcopter.py:15-18 (compiled into the device model)

    def rigidBodyDynamics(self, state, control, wind_body=(0.0, 0.0, 0.0)):
        """xDot (..., 8) of the rigid-body model (quadcopter.py:70-113)."""
        from .pytrees import AffineDynamics
        return AffineDynamics.from_function(QuadcopterRigidBody(0.0, wind_body), state, control).f

    def inertialDynamics(self, state, control, wind_ned=(0.0, 0.0, 0.0)):
        """xDot (..., 12) with position states (quadcopter.py:116-144)."""
        from .pytrees import AffineDynamics
        return AffineDynamics.from_function(QuadcopterEuler(0.0, wind_ned), state, control).f

    def trim(self, uvwTrim, tol=1e-9):
        """Trim at the body velocities uvwTrim (..., 3) (quadcopter.py:146-177) -> (xTrim (..., 8), uTrim (..., 4)).
        Raises RuntimeError("Trim failed") like the reference when an instance does not reach |rigidBodyDynamics| <= 1e-3."""
        import torch
        from . import _arrays as arr
        from . import _lib
        v = arr.to_device(uvwTrim, torch.float64)
        lead = tuple(v.shape[:-1])
        if v.shape[-1] != 3:
            raise ValueError("uvwTrim must have shape (..., 3)")
        v = v.reshape(-1, 3).contiguous()
        b = v.shape[0]
        xT = torch.empty((b, 8), dtype=torch.float64, device=v.device)
        uT = torch.empty((b, 4), dtype=torch.float64, device=v.device)
        res = torch.empty(b, dtype=torch.float64, device=v.device)
        rc = _lib.lib().zm_quadcopter_trim_f64(v.data_ptr(), None, xT.data_ptr(), uT.data_ptr(), res.data_ptr(), None, b, float(tol),
                                               ctypes.c_void_p(arr.stream_ptr(v)))
        _lib.check(rc, "Quadcopter.trim")
        if b and float(res.max()) > 1e-3:
            raise RuntimeError("Trim failed")
        return arr.result_like(xT.reshape(lead + (8,)), uvwTrim), arr.result_like(uT.reshape(lead + (4,)), uvwTrim)

    def linearize(self, x0, u0, dt=0):
        """Linearise the rigid-body dynamics about (x0 (..., 8), u0 (..., 4)) (quadcopter.py:179-201): continuous (A, B) for
        dt = 0, forward-Euler discretised `I + dt A`, `dt B` otherwise."""
        from .pytrees import AffineDynamics
        lin = AffineDynamics.from_function(QuadcopterRigidBody(float(dt)), x0, u0)
        return lin.f_x, lin.f_u
