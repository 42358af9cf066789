"""CPU checkers for the box-constrained LQ-MPC path (TEST INFRASTRUCTURE ONLY).

The reference (zopt/mpcUtils.py:12-81) states the QP and hands it to cvxpy -> OSQP 1.0.4 (requirements.txt:13,52);
that solver is not part of the reference tree and not installed here, and the reference's own test pins only
`status == "optimal"` (tests/test_mpcUtils.py:23): NUMERIC PARITY IS UNPINNED for this row.  What is checked instead:

  * `kkt_residuals`      -- solver-independent optimality certificate of a candidate (x, u) for the QP of
                            mpcUtils.py:48-59 (dynamics, bounds, and the reduced-gradient / complementarity
                            condition in the condensed variables);
  * `solve_reference`    -- an independent high-accuracy solve (condensed QP, SciPy trust-constr) for small problems;
  * `admm`               -- a NumPy restatement of the build's own ADMM (zopt_amd/csrc/mpc.hip) for iterate-level checks.
"""
from __future__ import annotations

import numpy as np
import scipy.optimize as spo

CHECK_EVERY = 8   # zopt_amd/csrc/mpc_common.h ZM_MPC_CHK: iterations between infeasibility-certificate checks


def condense(A, B, Q, R, Qf, N, x0):
    """x_k = Phi_k x0 + sum_j Gam[k,j] u_j ;  cost = u'Hu + 2 g'u + c  in the stacked control vector u (N*m)."""
    n, m = B.shape
    Phi = [np.eye(n)]
    for _ in range(N):
        Phi.append(A @ Phi[-1])
    Gam = np.zeros((N + 1, n, N * m))
    for k in range(1, N + 1):
        for j in range(k):
            Gam[k][:, j * m:(j + 1) * m] = Phi[k - 1 - j] @ B
    H = np.zeros((N * m, N * m))
    g = np.zeros(N * m)
    c = 0.0
    for k in range(N + 1):
        W = Qf if k == N else Q
        xk0 = Phi[k] @ x0
        H += Gam[k].T @ W @ Gam[k]
        g += Gam[k].T @ (0.5 * (W + W.T)) @ xk0
        c += xk0 @ W @ xk0
    for k in range(N):
        H[k * m:(k + 1) * m, k * m:(k + 1) * m] += R
    return Phi, Gam, H, g, c


def rollout(A, B, x0, u):
    x = [np.asarray(x0, dtype=np.float64)]
    for k in range(u.shape[0]):
        x.append(A @ x[-1] + B @ u[k])
    return np.stack(x)


def cost(Q, R, Qf, x, u):
    """mpcUtils.py:52-54."""
    return sum(x[k] @ Q @ x[k] + u[k] @ R @ u[k] for k in range(u.shape[0])) + x[-1] @ Qf @ x[-1]


def solve_reference(A, B, Q, R, Qf, N, x_lb, x_ub, u_lb, u_ub, x0):
    """Independent reference: condensed QP in u with linear inequality constraints on the states, SciPy trust-constr."""
    n, m = B.shape
    Phi, Gam, H, g, c = condense(A, B, Q, R, Qf, N, x0)
    Hs = H + H.T
    fun = lambda u: u @ H @ u + 2 * g @ u + c
    jac = lambda u: Hs @ u + 2 * g
    rows, lo, hi = [], [], []
    for k in range(1, N + 1):
        for i in range(n):
            if np.isfinite(x_lb[i]) or np.isfinite(x_ub[i]):
                rows.append(Gam[k][i])
                off = (Phi[k] @ x0)[i]
                lo.append(x_lb[i] - off)
                hi.append(x_ub[i] - off)
    cons = [spo.LinearConstraint(np.array(rows), np.array(lo), np.array(hi))] if rows else []
    bounds = spo.Bounds(np.tile(u_lb, N), np.tile(u_ub, N))
    res = spo.minimize(fun, np.zeros(N * m), jac=jac, hess=lambda u: Hs, method="trust-constr", bounds=bounds,
                       constraints=cons, options=dict(gtol=1e-12, xtol=1e-14, barrier_tol=1e-14, maxiter=5000))
    u = res.x.reshape(N, m)
    return rollout(A, B, x0, u), u, res.fun


def kkt_residuals(A, B, Q, R, Qf, N, x_lb, x_ub, u_lb, u_ub, x0, x, u, act_tol=1e-6):
    """Solver-independent certificate for a candidate (x (N+1,n), u (N,m)).

    Returns dict(dyn, bound, stat): max dynamics defect, max bound violation, and the stationarity defect: the norm
    of the projection of the reduced gradient (condensed variables) onto the cone of feasible directions of the
    constraints that are inactive / active at the candidate -- computed by a small non-negative least squares."""
    n, m = B.shape
    dyn = max(np.max(np.abs(x[k + 1] - (A @ x[k] + B @ u[k]))) for k in range(N))
    dyn = max(dyn, np.max(np.abs(x[0] - x0)))
    viol = max(np.max(np.maximum(x_lb - x, 0)), np.max(np.maximum(x - x_ub, 0)), np.max(np.maximum(u_lb - u, 0)),
               np.max(np.maximum(u - u_ub, 0)))
    Phi, Gam, H, g, c = condense(A, B, Q, R, Qf, N, x0)
    uv = u.reshape(-1)
    grad = (H + H.T) @ uv + 2 * g
    # active constraint normals (outward): grad + sum mu_i a_i = 0 with mu >= 0
    normals = []
    for k in range(N):
        for j in range(m):
            e = np.zeros(N * m)
            e[k * m + j] = 1.0
            if u[k, j] >= u_ub[j] - act_tol:
                normals.append(e)
            if u[k, j] <= u_lb[j] + act_tol:
                normals.append(-e)
    for k in range(1, N + 1):
        for i in range(n):
            if x[k, i] >= x_ub[i] - act_tol:
                normals.append(Gam[k][i])
            if x[k, i] <= x_lb[i] + act_tol:
                normals.append(-Gam[k][i])
    if normals:
        Nm = np.array(normals).T
        mu, rnorm = spo.nnls(Nm, -grad)
        stat = rnorm
    else:
        stat = np.linalg.norm(grad)
    return dict(dyn=dyn, bound=viol, stat=stat / max(1.0, np.linalg.norm(grad)))


def admm(A, B, Q, R, Qf, N, x_lb, x_ub, u_lb, u_ub, x0, rho=1.0, eps_abs=1e-5, eps_rel=1e-5, max_iter=10000,
         eps_prim_inf=1e-4, alpha=1.0):
    """NumPy restatement of zopt_amd/csrc/mpc.hip for ONE instance (fixed penalty).  `alpha` is OSQP's over-relaxation: the
    relaxed iterate alpha w + (1 - alpha) y_prev enters the projection and the dual update.  Returns (x, u, status, iters)."""
    n, m = B.shape
    if np.any(x0 < x_lb) or np.any(x0 > x_ub):
        return rollout(A, B, x0, np.zeros((N, m))), np.zeros((N, m)), "infeasible", 0
    Hx, Hu = 2 * Q + rho * np.eye(n), 2 * R + rho * np.eye(m)
    P = 2 * Qf + rho * np.eye(n)
    K, Mi = [None] * N, [None] * N
    for k in range(N - 1, -1, -1):
        Suu = Hu + B.T @ P @ B
        Sux = B.T @ P @ A
        Mi[k] = np.linalg.inv(Suu)
        K[k] = Mi[k] @ Sux
        P = Hx + A.T @ P @ A - Sux.T @ K[k]
    yx, yu = np.zeros((N, n)), np.zeros((N, m))      # yx[k] is the copy of x_{k+1}
    lx, lu = np.zeros((N, n)), np.zeros((N, m))
    status, it = "user_limit", 0
    x, u = None, None
    for it in range(1, max_iter + 1):
        chk = (it % CHECK_EVERY) == 0
        zx, zu = yx - lx, yu - lu
        p = -rho * zx[N - 1]
        kf = np.zeros((N, m))
        for k in range(N - 1, -1, -1):
            qu = -rho * zu[k] + B.T @ p
            kf[k] = Mi[k] @ qu
            p = (-rho * zx[k - 1] if k >= 1 else 0.0) + A.T @ p - K[k].T @ qu
        xs = [x0]
        us = []
        for k in range(N):
            us.append(-K[k] @ xs[-1] - kf[k])
            xs.append(A @ xs[-1] + B @ us[-1])
        x, u = np.stack(xs), np.stack(us)
        xh, uh = alpha * x[1:] + (1.0 - alpha) * yx, alpha * u + (1.0 - alpha) * yu     # relaxed iterates (alpha = 1: x, u)
        yxn = np.clip(xh + lx, x_lb, x_ub)
        yun = np.clip(uh + lu, u_lb, u_ub)
        rp = max(np.max(np.abs(x[1:] - yxn)), np.max(np.abs(u - yun)))                 # primal residual of the actual iterate
        rx, ru = xh - yxn, uh - yun                                                     # dual step
        rd = rho * max(np.max(np.abs(yxn - yx)), np.max(np.abs(yun - yu)))
        lx, lu = lx + rx, lu + ru
        yx, yu = yxn, yun
        ep = eps_abs + eps_rel * max(np.max(np.abs(x[1:])), np.max(np.abs(u)), np.max(np.abs(yx)), np.max(np.abs(yu)))
        ed = eps_abs + eps_rel * rho * max(np.max(np.abs(lx)), np.max(np.abs(lu)))
        if rp <= ep and rd <= ed:
            status = "optimal"
            break
        if chk:   # OSQP-style primal infeasibility certificate on v = w - y
            s = rx[N - 1].copy()
            gmax = 0.0
            for k in range(N - 1, -1, -1):
                gmax = max(gmax, np.max(np.abs(ru[k] + B.T @ s)))
                s = (rx[k - 1] if k >= 1 else 0.0) + A.T @ s
            sup = 0.0
            for r_, lo_, hi_ in ((rx, x_lb, x_ub), (ru, u_lb, u_ub)):
                lo_b, hi_b = np.broadcast_to(lo_, r_.shape), np.broadcast_to(hi_, r_.shape)
                pos, neg = r_ > 0, r_ < 0
                sup += np.sum(r_[pos] * hi_b[pos]) + np.sum(r_[neg] * lo_b[neg])
            dn = max(np.max(np.abs(rx)), np.max(np.abs(ru)))
            if gmax <= eps_prim_inf * dn and (s @ x0 - sup) > eps_prim_inf * dn:
                status = "infeasible"
                break
    return x, u, status, it
