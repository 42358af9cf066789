"""Array-container mirror of ``zopt.pytrees`` (reference pytrees.py:6-12, 58-69, 84-98, 129-136, 165-177, 207-213).

Same NamedTuple names, field order and per-time-step slicing ``obj[k]``; the fields are NumPy arrays or
torch-ROCm tensors (optionally with leading batch axes, time axis right after them).  The Taylor-expansion
constructors (``from_function`` / ``from_trajectory`` / ``fromTerminalCostFunction``, reference pytrees.py:72-81,
100-115, 139-153, 180-194) keep their names and argument order; where the reference differentiates an arbitrary
JAX callable, these take a *registered device model* / cost (``zopt_amd.models``) and run the dual-number kernels
(``zm_linearize_dynamics_f64``, ``zm_quadratic_dynamics_f64``, ``zm_quadratize_cost_f64``).
"""
from __future__ import annotations

import ctypes
from typing import NamedTuple

import numpy as np


def _expand(kind, fun, xTraj, uTraj):
    """Run one of the expansion kernels along (xTraj (...,N+1,n), uTraj (...,N,m)); returns the output tensors shaped
    like the inputs' leading axes.  kind: 'affine' | 'quadratic' | 'cost' | 'terminal'."""
    import torch
    from . import _arrays as arr
    from . import _lib
    arr.require_gpu()
    if not hasattr(fun, "c_struct"):
        return _expand_callable(kind, fun, xTraj, uTraj)
    n, m = fun.n, fun.m
    dt = torch.float64
    x = arr.to_device(xTraj, dt)
    lead = tuple(x.shape[:-2])
    N = x.shape[-2] - 1
    x = x.reshape(-1, N + 1, n).contiguous()
    if uTraj is None:
        u = torch.zeros((x.shape[0], max(N, 1), m), dtype=dt, device=x.device)
    else:
        u = arr.to_device(uTraj, dt).reshape(-1, N, m).contiguous()
        if u.shape[0] != x.shape[0]:
            raise ValueError("xTraj / uTraj batch axes differ")
    B, dev = x.shape[0], x.device
    st = ctypes.c_void_p(arr.stream_ptr(x))
    cs = fun.c_struct()
    pc = ctypes.addressof(cs)
    lib = _lib.lib()
    E = lambda *shape: torch.empty(shape, dtype=dt, device=dev)
    if kind in ("affine", "quadratic"):
        f, f_x, f_u = E(B, N, n), E(B, N, n, n), E(B, N, n, m)
        _lib.check(lib.zm_linearize_dynamics_f64(pc, x.data_ptr(), u.data_ptr(), None, f.data_ptr(), f_x.data_ptr(),
                                                 f_u.data_ptr(), B, N, st), "AffineDynamics.from_trajectory")
        outs = [f, f_x, f_u]
        if kind == "quadratic":
            f_xx, f_ux, f_uu = E(B, N, n, n, n), E(B, N, n, m, n), E(B, N, n, m, m)
            _lib.check(lib.zm_quadratic_dynamics_f64(pc, x.data_ptr(), u.data_ptr(), None, f_xx.data_ptr(),
                                                     f_ux.data_ptr(), f_uu.data_ptr(), B, N, st),
                       "QuadraticDynamics.from_trajectory")
            outs += [f_xx, f_ux, f_uu]
        outs = [o.reshape(lead + tuple(o.shape[1:])) for o in outs]
    else:
        c, c_x, c_u, v, v_x = E(B, N), E(B, N, n), E(B, N, m), E(B), E(B, n)
        c_xx, c_ux, c_uu, v_xx = E(n, n), E(m, n), E(m, m), E(n, n)
        _lib.check(lib.zm_quadratize_cost_f64(pc, n, m, x.data_ptr(), u.data_ptr(), None, c.data_ptr(), c_x.data_ptr(),
                                              c_u.data_ptr(), v.data_ptr(), v_x.data_ptr(), c_xx.data_ptr(),
                                              c_ux.data_ptr(), c_uu.data_ptr(), v_xx.data_ptr(), B, N, st),
                   "QuadraticCostFunction.from_trajectory")
        if kind == "cost":
            outs = [c, c_x, c_u] + [h.expand((B, N) + tuple(h.shape)).contiguous() for h in (c_xx, c_ux, c_uu)]
            outs = [o.reshape(lead + tuple(o.shape[1:])) for o in outs]
        else:     # terminal cost at x_N; a dummy time axis keeps _point's squeeze uniform
            outs = [v[:, None], v_x[:, None], v_xx.expand((B, 1, n, n)).contiguous()]
            outs = [o.reshape(lead + tuple(o.shape[1:])) for o in outs]
    return [arr.result_like(o, xTraj) for o in outs]


def _expand_callable(kind, fun, xTraj, uTraj):
    """The expansions for torch callables (zopt_amd/generic.py: torch.func on the GPU, the part jax.jacobian / jax.hessian play in
    the reference).  `fun`: a dynamics callable f(x, u) -> x+ for 'affine' / 'quadratic'; for 'cost' / 'terminal' a callable or an
    object with `.runningCost(x, u)` / `.terminalCost(x)` (CostFunction, reference pytrees.py:27-38)."""
    import torch
    from . import _arrays as arr
    from . import generic
    if kind in ("affine", "quadratic") and not callable(fun):
        raise TypeError("expected a registered device model or a torch callable f(x, u) -> x+, got " + type(fun).__name__)
    x = arr.to_device(xTraj, torch.float64)
    lead, (N1, n) = tuple(x.shape[:-2]), x.shape[-2:]
    x = x.reshape(-1, N1, n)
    if kind == "terminal":
        tc = getattr(fun, "terminalCost", fun)
        outs = [o[:, None] for o in generic.expand_terminal(tc, x[:, -1])]
    else:
        u = arr.to_device(uTraj, torch.float64)
        u = u.reshape(-1, N1 - 1, u.shape[-1])
        if kind == "cost":
            outs = generic.expand_cost(getattr(fun, "runningCost", fun), x, u)
        else:
            outs = generic.expand_dynamics(fun, x, u, kind == "quadratic")
    outs = [o.reshape(lead + tuple(o.shape[1:])) for o in outs]
    return [arr.result_like(o, xTraj) for o in outs]


def _point(kind, fun, x0, u0):
    """Expansion about a single point (or a batch of points): a length-1 trajectory with the time axis squeezed."""
    shp = tuple(x0.shape) if hasattr(x0, "shape") else np.shape(x0)
    xs = _stack2(x0)
    us = None if u0 is None else _unsq(u0)
    outs = _expand(kind, fun, xs, us)
    k = len(shp) - 1
    return [o[(slice(None),) * k + (0,)] for o in outs]


def _unsq(a):
    return a.unsqueeze(-2) if hasattr(a, "unsqueeze") else np.asarray(a)[..., None, :]


def _stack2(x):
    """(..., n) -> (..., 2, n): x_0 = x, x_1 = x (the kernels take N+1 states for N steps)"""
    if hasattr(x, "unsqueeze"):
        return x.unsqueeze(-2).expand(tuple(x.shape[:-1]) + (2, x.shape[-1]))
    x = np.asarray(x)
    return np.broadcast_to(x[..., None, :], x.shape[:-1] + (2, x.shape[-1]))


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

    @classmethod
    def fromTerminalCostFunction(cls, costFun, xf):
        """Quadratic value function of the terminal cost about the final state xf (..., n)   (pytrees.py:72-81)"""
        v, v_x, v_xx = _point("terminal", costFun, xf, None)
        return cls(v, v_x, v_xx)


class QuadraticCostFunction(NamedTuple):
    """(c, c_x, c_u, c_xx, c_ux, c_uu)"""
    c: object
    c_x: object
    c_u: object
    c_xx: object
    c_ux: object
    c_uu: object

    @classmethod
    def from_function(cls, costFun, x0, u0):
        """Second-order expansion of the running cost about (x0, u0)   (pytrees.py:100-107)"""
        return cls(*_point("cost", costFun, x0, u0))

    @classmethod
    def from_trajectory(cls, costFun, traj):
        """Second-order expansion of the running cost about (xTraj[:-1], uTraj)   (pytrees.py:109-115)"""
        return cls(*_expand("cost", costFun, tuple.__getitem__(traj, 0), tuple.__getitem__(traj, 1)))

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

    @classmethod
    def from_function(cls, dynFun, x0, u0):
        """First-order expansion of the registered model about (x0, u0)   (pytrees.py:139-145)"""
        return cls(*_point("affine", dynFun, x0, u0))

    @classmethod
    def from_trajectory(cls, dynFun, traj):
        """First-order expansion about (xTraj[:-1], uTraj); xTraj (..., N+1, n), uTraj (..., N, m)   (pytrees.py:147-153)"""
        return cls(*_expand("affine", dynFun, tuple.__getitem__(traj, 0), tuple.__getitem__(traj, 1)))

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

    @classmethod
    def from_function(cls, dynFun, x0, u0):
        """Second-order expansion of the registered model about (x0, u0)   (pytrees.py:180-186)"""
        return cls(*_point("quadratic", dynFun, x0, u0))

    @classmethod
    def from_trajectory(cls, dynFun, traj):
        """Second-order expansion about (xTraj[:-1], uTraj)   (pytrees.py:188-194)"""
        return cls(*_expand("quadratic", dynFun, tuple.__getitem__(traj, 0), tuple.__getitem__(traj, 1)))

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


class QuadraticDeltaCost(NamedTuple):
    """(dJ_lin, dJ_quad):  dJ_exp(alpha) = alpha * (dJ_lin + alpha * dJ_quad)   (reference pytrees.py:226-236; only its unused
    `forwardPass` line search consumes it)"""
    dJ_lin: object
    dJ_quad: object

    def __call__(self, alpha):
        dJ_lin, dJ_quad = tuple.__iter__(self)
        return alpha * (dJ_lin + alpha * dJ_quad)
