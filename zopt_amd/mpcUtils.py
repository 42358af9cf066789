"""Drop-in for the solve path of ``zopt.mpcUtils`` (class ``lqrMpc``) on MI355X HIP kernels.

Same constructor and ``solve`` signature as the reference (mpcUtils.py:14-26, 61-81).  New: ``x0`` may carry leading
batch axes -- every initial state is an independent QP instance solved by one GPU lane.  The plotting / animation helpers
of the reference module (mpcUtils.py:84-202) are presentation code and not part of this package.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _arrays as arr
from . import _lib
from .pytrees import Trajectory

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

# cvxpy's status strings (mpcUtils.py:74,78).  "optimal_inaccurate" = OSQP's "solved inaccurate": the iteration limit was reached with
# both residuals within 10x their tolerances.  "unbounded" cannot arise for this QP (Q, Qf >= 0, R > 0); an instance that is neither
# solved nor certified infeasible at the limit is "user_limit" (cvxpy's name for OSQP's "maximum iterations reached").
_STATUS = {1: "optimal", 2: "infeasible", 3: "user_limit", 4: "optimal_inaccurate"}


class lqrMpc():

    def __init__(self, A, B, Q, R, N, x_lb, x_ub, u_lb, u_ub, Qf=None):
        """
        Setup an LQR MPC problem (reference mpcUtils.py:14-59)

        Arguments
        ---------
            A : Dynamics matrix; shape = (n,n)
            B : Input matrix; shape = (n,m)
            Q : State cost matrix; shape = (n,n)
            R : Control cost matrix; shape = (m,m)
            N : MPC horizon
            x_lb, x_ub : State lower / upper bound (+-inf allowed)
            u_lb, u_ub : Control lower / upper bound
            Qf : Terminal cost matrix, optional; shape = (n,n).  Defaults to Q
        """
        if Qf is None:
            Qf = Q
        f64 = lambda X: np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        self.A, self.B, self.Q, self.R, self.Qf = f64(A), f64(B), f64(Q), f64(R), f64(Qf)
        self.n, self.m = self.B.shape
        self.N = int(N)
        self.x_lb, self.x_ub, self.u_lb, self.u_ub = f64(x_lb), f64(x_ub), f64(u_lb), f64(u_ub)
        if self.A.shape != (self.n, self.n) or self.Q.shape != (self.n, self.n) or self.R.shape != (self.m, self.m) \
                or self.x_lb.shape != (self.n,) or self.u_lb.shape != (self.m,) or self.N < 1:
            raise ValueError("inconsistent lqrMpc problem shapes")
        # cvxpy refuses the reference's problem (DCPError at solve time) unless every quad_form weight is positive semidefinite; here the
        # ADMM's Hessians 2Q + rho I would hide a slightly indefinite weight and return the stationary point of a non-convex problem
        for name, W in (("Q", self.Q), ("R", self.R), ("Qf", self.Qf)):
            w = np.linalg.eigvalsh(0.5 * (W + W.T))
            if w[0] < -1e-10 * max(1.0, abs(w[-1])):
                raise ValueError(f"lqrMpc: {name} is not positive semidefinite (smallest eigenvalue {w[0]:.3g}): the problem is not "
                                 f"convex (cvxpy raises DCPError for the reference's quad_form)")
        self._dev = None
        self._tables = {}
        self._ws = None   # ((batch, device, rho), ADMM workspace) of the last solve: warm start
        # one penalty for every instance: the geometric mean of the cost curvatures keeps both blocks of the
        # w-update Hessian (2Q + rho I, 2R + rho I) comparably conditioned
        self.rho = float(np.sqrt(max(np.trace(2 * self.Q) / self.n, 1e-12) * max(np.trace(2 * self.R) / self.m, 1e-12)))
        # The solve kernels are compiled for a few (n, m); any other n <= 24, m <= 8 is embedded in the next one: the extra
        # states follow x+ = 0 from x = 0 with unit weight and no bound, the extra controls act on nothing and cost u^2 --
        # they stay exactly zero and are sliced off the results.
        self._n_user, self._m_user = self.n, self.m
        fit = [(ns, mc) for (ns, mc) in self._COMPILED if ns >= self.n and mc >= self.m]
        if not fit:
            raise ValueError(f"lqrMpc: (n={self.n}, m={self.m}) outside the compiled kernels (n <= 24, m <= 8)")
        ns, mc = min(fit, key=lambda t: (t[0] * t[1], t[0]))
        if (ns, mc) != (self.n, self.m):
            n0, m0 = self.n, self.m
            pad2 = lambda X, r, c, d: np.block([[X, np.zeros((X.shape[0], c - X.shape[1]))],
                                                [np.zeros((r - X.shape[0], X.shape[1])), d * np.eye(r - X.shape[0], c - X.shape[1])]])
            self.A = pad2(self.A, ns, ns, 0.0)
            self.B = pad2(self.B, ns, mc, 0.0)
            self.Q, self.Qf = pad2(self.Q, ns, ns, 1.0), pad2(self.Qf, ns, ns, 1.0)
            self.R = pad2(self.R, mc, mc, 1.0)
            inf = np.inf
            self.x_lb = np.concatenate([self.x_lb, np.full(ns - n0, -inf)])
            self.x_ub = np.concatenate([self.x_ub, np.full(ns - n0, inf)])
            self.u_lb = np.concatenate([self.u_lb, np.full(mc - m0, -inf)])
            self.u_ub = np.concatenate([self.u_ub, np.full(mc - m0, inf)])
            self.n, self.m = ns, mc

    # (24, 8): beyond the 16-index tile of the 16-lanes-per-instance kernel -- the lane-per-instance kernel with a fixed penalty (no
    # tabulated levels): a coverage path, an order of magnitude slower per instance than the (12, 4) kernels
    _COMPILED = ((24, 8), (12, 4), (8, 4), (4, 2), (4, 1), (2, 2), (2, 1), (1, 1))

    N_LEVELS, RHO_STEP = 7, 5.0      # adaptive penalty: rho * 5^(l - 3), l = 0..6  (OSQP changes rho only by factors >= 5)

    def _device_problem(self, rho, adaptive):
        arr.require_gpu()
        if self._dev is None:
            self._dev = {k: arr.to_device(getattr(self, k), torch.float64)
                         for k in ("A", "B", "Q", "R", "Qf", "x_lb", "x_ub", "u_lb", "u_ub")}
        key = (rho, bool(adaptive))
        if key not in self._tables:
            d = self._dev
            nl = self.N_LEVELS if adaptive else 1
            l0 = nl // 2
            K = torch.empty((nl, self.N, self.m, self.n), dtype=torch.float64, device=d["A"].device)
            Mi = torch.empty((nl, self.N, self.m, self.m), dtype=torch.float64, device=d["A"].device)
            for l in range(nl):
                rc = _lib.lib().zm_mpc_setup_f64(d["A"].data_ptr(), d["B"].data_ptr(), d["Q"].data_ptr(), d["R"].data_ptr(),
                                                 d["Qf"].data_ptr(), float(rho) * self.RHO_STEP ** (l - l0), self.N, self.n,
                                                 self.m, K[l].data_ptr(), Mi[l].data_ptr(),
                                                 ctypes.c_void_p(arr.stream_ptr(K)))
                _lib.check(rc, "lqrMpc setup")
            self._tables[key] = (K, Mi, nl, l0)
        return self._dev, self._tables[key]

    def solve(self, x0, **kwargs):
        """
        Solve the MPC step at state x0 (reference mpcUtils.py:61-81)

        Arguments
        ---------
            x0 : Initial state (n,) -- or (..., n): a batch of independent instances
            **kwargs : solver options, named as the OSQP options the reference forwards through cvxpy
                (demos/lqrMpc.py:32): eps_abs, eps_rel (default 1e-5, cvxpy's OSQP default), max_iter (default 10000),
                rho, adaptive_rho (default True: the penalty moves between 7 tabulated levels rho * 5^l as OSQP's does),
                alpha (over-relaxation in (0, 2); default 1.6, OSQP's default, i.e. what the reference's solve runs with),
                eps_prim_inf (default 1e-4), warm_start (default True, as cvxpy: a solve for the same batch shape
                starts from the previous solve's ADMM iterates; `warm_start="shift"` (extension) advances them by one
                horizon step first, the right guess inside the receding-horizon loop of demos/lqrMpc.py:41-48);
                `solver` may be None or "OSQP" (the build has one solver); eps_dual_inf / verbose / polish are accepted
                and ignored.

        Returns
        -------
            u : Optimal control at current time step (…, m)
            traj : Trajectory tuple (xTraj (…, N+1, n), uTraj (…, N, m))
            status : problem status, one of [optimal, optimal_inaccurate, infeasible, user_limit] (a list of them for a batch)
        """
        solver = kwargs.pop("solver", None)
        if solver not in (None, "OSQP"):
            raise ValueError(f"solver {solver!r} is not available in zopt_amd (ADMM only; pass solver='OSQP' or None)")
        eps_abs = float(kwargs.pop("eps_abs", 1e-5))
        eps_rel = float(kwargs.pop("eps_rel", 1e-5))
        max_iter = int(kwargs.pop("max_iter", 10000))
        rho = float(kwargs.pop("rho", self.rho))
        adaptive = bool(kwargs.pop("adaptive_rho", True))        # OSQP / cvxpy default
        eps_pinf = float(kwargs.pop("eps_prim_inf", 1e-4))
        alpha = float(kwargs.pop("alpha", 1.6))
        if not (0.0 < alpha < 2.0):
            raise ValueError("alpha must lie in (0, 2)")
        warm = kwargs.pop("warm_start", kwargs.pop("warm_starting", True))
        shift = isinstance(warm, str) and warm == "shift"     # extension: previous iterates advanced by one horizon step
        warm = bool(warm)
        for k in ("eps_dual_inf", "verbose", "polish", "polishing"):
            kwargs.pop(k, None)
        if kwargs:
            raise TypeError(f"unknown solver options {sorted(kwargs)}")
        shp = tuple(x0.shape) if hasattr(x0, "shape") else tuple(np.shape(x0))
        if len(shp) < 1 or shp[-1] != self._n_user:
            raise ValueError(f"x0 has shape {shp}, expected (..., {self._n_user})")
        lead = shp[:-1]
        d, (K, Mi, n_levels, level0) = self._device_problem(rho, adaptive)
        dx0 = arr.to_device(x0, torch.float64).reshape(-1, self._n_user)
        if self.n != self._n_user:
            dx0 = torch.nn.functional.pad(dx0, (0, self.n - self._n_user))
        dx0 = dx0.contiguous()
        Bn = dx0.shape[0]
        dev = dx0.device
        N, n, m = self.N, self.n, self.m
        key = (Bn, str(dev), rho, adaptive)
        warm = warm and self._ws is not None and self._ws[0] == key
        if not warm:
            self._ws = (key, torch.empty(4 * Bn * N * (n + m), dtype=torch.float64, device=dev))
        ws = self._ws[1]
        xT = torch.empty((Bn, N + 1, n), dtype=torch.float64, device=dev)
        uT = torch.empty((Bn, N, m), dtype=torch.float64, device=dev)
        st = torch.empty(Bn, dtype=torch.int32, device=dev)
        its = torch.empty(Bn, dtype=torch.int32, device=dev)
        res = torch.empty((Bn, 2), dtype=torch.float64, device=dev)
        rc = _lib.lib().zm_mpc_solve_relaxed_f64(d["A"].data_ptr(), d["B"].data_ptr(), K.data_ptr(), Mi.data_ptr(), n_levels,
                                                  level0, self.RHO_STEP, alpha, d["x_lb"].data_ptr(), d["x_ub"].data_ptr(),
                                                  d["u_lb"].data_ptr(), d["u_ub"].data_ptr(), dx0.data_ptr(), rho, eps_abs,
                                                  eps_rel, eps_pinf, max_iter, (2 if shift else 1) if warm else 0,
                                                  ws.data_ptr(), xT.data_ptr(), uT.data_ptr(), st.data_ptr(), its.data_ptr(),
                                                  res.data_ptr(), Bn, N, n, m, ctypes.c_void_p(arr.stream_ptr(dx0)))
        _lib.check(rc, "lqrMpc.solve")
        self.last_iterations = its.reshape(lead).cpu().numpy()
        self.last_residuals = res.reshape(lead + (2,)).cpu().numpy()
        codes = st.cpu().numpy().reshape(lead)
        xo = arr.result_like(xT.reshape(lead + (N + 1, n))[..., :self._n_user], x0)
        uo = arr.result_like(uT.reshape(lead + (N, m))[..., :self._m_user], x0)
        if len(lead) == 0:
            status = _STATUS[int(codes)]
        else:
            status = np.vectorize(_STATUS.get, otypes=[object])(codes)
        return uo[..., 0, :], Trajectory(xo, uo), status
