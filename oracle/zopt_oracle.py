"""CPU oracle: NumPy restatement of the zopt hot path (TEST INFRASTRUCTURE ONLY).

This file restates, in plain NumPy (fp64 or fp32, following the input dtype), the
algorithms of the reference ``zprihoda/zopt`` that SURVEY.md section 8(a) puts on the hot
path.  Every function cites the reference ``file:line`` it follows and keeps the
reference's operation order (association of the matrix products, Joseph-form update,
``solve`` = LU with partial pivoting via LAPACK) so that it is as close to the
reference's own arithmetic as a NumPy program can be.

Batch extension (new in the build): every array may carry extra LEADING axes; with no
leading axes the functions take and return exactly the reference's shapes.

Pinning status (see DESIGN.md "Oracle"): the reference itself cannot run here (``jax`` /
``cvxpy`` are not installed: an ordinary ModuleNotFoundError, not a refusal) and it ships
no fixture files, so this oracle is pinned by the reference's own known-answer tests
(``tests/test_lqrUtils.py``, ``tests/test_ilqrUtils.py``, ``tests/test_quadcopter.py``,
``tests/test_pytrees.py`` -- transcribed as data in ``tests/golden/reference_kats.json``)
and cross-checked against SciPy's DARE solver (``lqrUtils.py:202-203`` is two lines of
real SciPy).  ``lqrMpc`` (cvxpy -> OSQP 1.0.4) has NO numeric pin in the reference:
"parity unpinned" -- see ``mpc_*`` below.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import
this module.  The product (``zopt_amd``) never does.
"""
from __future__ import annotations

from typing import Callable, NamedTuple

import numpy as np


def _T(x):
    """Matrix transpose of the last two axes (the reference's ``.T`` on 2-D arrays)."""
    return np.swapaxes(x, -1, -2)


def _mv(M, v):
    """Batched matrix @ vector."""
    return (M @ v[..., None])[..., 0]


# ----------------------------------------------------------------------------------------
# A1  lqrUtils.discreteFiniteHorizonLqr                      reference lqrUtils.py:144-173
# ----------------------------------------------------------------------------------------
def discreteFiniteHorizonLqr(A, B, Q, R, N):
    """Backward Riccati recursion, Joseph-form value update.

    Follows ``lqrUtils.py:167-172``:
        V <- Q[-1]                                              (:172, quirk Q1)
        for k = N-1 .. 0:
            L_k = solve(R_k + B_k^T V B_k,  B_k^T V A_k)        (:168)
            V   = Q_k + L_k^T R_k L_k + (A_k-B_k L_k)^T V (A_k-B_k L_k)   (:169)
    Returns L with shape (..., N, m, n); control law u = -L x (:151).
    """
    A, B, Q, R = (np.asarray(x) for x in (A, B, Q, R))
    V = Q[..., -1, :, :]
    m = B.shape[-1]
    n = B.shape[-2]
    L = np.empty(A.shape[:-3] + (N, m, n), dtype=A.dtype)
    for k in range(N - 1, -1, -1):
        Ak, Bk, Qk, Rk = A[..., k, :, :], B[..., k, :, :], Q[..., k, :, :], R[..., k, :, :]
        BtV = _T(Bk) @ V
        Lk = np.linalg.solve(Rk + BtV @ Bk, BtV @ Ak)
        Acl = Ak - Bk @ Lk
        V = Qk + (_T(Lk) @ Rk) @ Lk + (_T(Acl) @ V) @ Acl
        L[..., k, :, :] = Lk
    return L


# ----------------------------------------------------------------------------------------
# A2  lqrUtils.bilinearAffineLqr                             reference lqrUtils.py:207-262
# ----------------------------------------------------------------------------------------
def bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, N):
    """Finite-horizon LQR with cross term H, affine dynamics d, linear costs q, r, q0.

    Follows ``lqrUtils.py:242-261``; carry (V, v, v0) <- (Q[-1], q[-1], q0[-1]) (:261).
    Returns (L (..., N, m, n), l (..., N, m)); law u = -L x - l.
    """
    A, B, d, Q, R, H, q, r, q0 = (np.asarray(x) for x in (A, B, d, Q, R, H, q, r, q0))
    n, m = B.shape[-2:]
    V = Q[..., -1, :, :]
    v = q[..., -1, :]
    v0 = q0[..., -1]
    LArr = np.empty(A.shape[:-3] + (N, m, n), dtype=A.dtype)
    lArr = np.empty(A.shape[:-3] + (N, m), dtype=A.dtype)
    for k in range(N - 1, -1, -1):
        Ak, Bk, dk = A[..., k, :, :], B[..., k, :, :], d[..., k, :]
        Qk, Rk, Hk = Q[..., k, :, :], R[..., k, :, :], H[..., k, :, :]
        qk, rk, q0k = q[..., k, :], r[..., k, :], q0[..., k]
        Vd = _mv(V, dk)
        Su = rk + _mv(_T(Bk), v) + _mv(_T(Bk), _mv(_T(V), dk))   # r + v^T B + d^T V B      (:244)
        Suu = Rk + (_T(Bk) @ V) @ Bk                              # (:245)
        Sux = Hk + (_T(Bk) @ V) @ Ak                              # (:246)
        L = np.linalg.solve(Suu, Sux)                             # (:248)
        l = np.linalg.solve(Suu, Su[..., None])[..., 0]           # (:249)
        VNew = Qk + (_T(Ak) @ V) @ Ak - (_T(L) @ Suu) @ L         # (:251)
        vNew = qk + _mv(_T(Ak), v + Vd) - _mv(_T(Sux), l)         # (:252)
        v0New = (v0 + q0k + np.sum(dk * v, -1) + 0.5 * np.sum(dk * Vd, -1)
                 - 0.5 * np.sum(l * Su, -1))                      # (:253)
        V, v, v0 = VNew, vNew, v0New
        LArr[..., k, :, :] = L
        lArr[..., k, :] = l
    return LArr, lArr


# ----------------------------------------------------------------------------------------
# A12  layout types                                 reference pytrees.py:6-12,58-69,84-98,
#                                                   129-136,165-177,207-213
# ----------------------------------------------------------------------------------------
class Trajectory(NamedTuple):
    xTraj: np.ndarray
    uTraj: np.ndarray


class QuadraticValueFunction(NamedTuple):
    v: np.ndarray
    v_x: np.ndarray
    v_xx: np.ndarray


class QuadraticCostFunction(NamedTuple):
    c: np.ndarray
    c_x: np.ndarray
    c_u: np.ndarray
    c_xx: np.ndarray
    c_ux: np.ndarray
    c_uu: np.ndarray


class AffineDynamics(NamedTuple):
    f: np.ndarray
    f_x: np.ndarray
    f_u: np.ndarray


class QuadraticDynamics(NamedTuple):
    f: np.ndarray
    f_x: np.ndarray
    f_u: np.ndarray
    f_xx: np.ndarray
    f_ux: np.ndarray
    f_uu: np.ndarray


class AffinePolicy(NamedTuple):
    l: np.ndarray
    L: np.ndarray


# ----------------------------------------------------------------------------------------
# A5  ensurePositiveDefinite & friends                    reference ilqrUtils.py:217-257
# ----------------------------------------------------------------------------------------
def ensurePositiveDefinite(a, eps=1e-3):
    """``w, v = eigh(a); (v * max(w, eps)) @ v.T``  (ilqrUtils.py:217-219).

    ``jnp.linalg.eigh`` symmetrises its input by default (symmetrize_input=True):
    (a + a^H)/2 -- restated explicitly because ``numpy.linalg.eigh`` instead reads only
    the lower triangle.
    """
    a = np.asarray(a)
    a = 0.5 * (a + _T(a))
    w, v = np.linalg.eigh(a)
    return (v * np.maximum(w, eps)[..., None, :]) @ _T(v)


def conditionQuadraticCost(cost: QuadraticCostFunction) -> QuadraticCostFunction:
    """ilqrUtils.py:222-234 -- PD-project the stacked (n+m) cost Hessian per time step."""
    c, c_x, c_u, c_xx, c_ux, c_uu = cost
    n = c_xx.shape[-1]
    m = c_uu.shape[-1]
    top = np.concatenate([c_xx, _T(c_ux)], axis=-1)
    bot = np.concatenate([c_ux, c_uu], axis=-1)
    c_zz = ensurePositiveDefinite(np.concatenate([top, bot], axis=-2))
    return QuadraticCostFunction(c, c_x, c_u, c_zz[..., :n, :n], c_zz[..., n:, :n], c_zz[..., n:, n:])


def conditionValueFunction(Vf: QuadraticValueFunction) -> QuadraticValueFunction:
    """ilqrUtils.py:254-257."""
    return QuadraticValueFunction(Vf.v, Vf.v_x, ensurePositiveDefinite(Vf.v_xx))


def conditionQuadraticDynamics(dyn: QuadraticDynamics, v_x):
    """ilqrUtils.py:237-251 -- one time step (leading batch axes allowed)."""
    _, _, _, f_xx, f_ux, f_uu = dyn
    vf_xx = np.einsum('...i,...ijk->...jk', v_x, f_xx)
    vf_uu = np.einsum('...i,...ijk->...jk', v_x, f_uu)
    vf_ux = np.einsum('...i,...ijk->...jk', v_x, f_ux)
    n = vf_xx.shape[-1]
    top = np.concatenate([vf_xx, _T(vf_ux)], axis=-1)
    bot = np.concatenate([vf_ux, vf_uu], axis=-1)
    vf_zz = ensurePositiveDefinite(np.concatenate([top, bot], axis=-2))
    return vf_zz[..., :n, :n], vf_zz[..., n:, :n], vf_zz[..., n:, n:]


# ----------------------------------------------------------------------------------------
# A3  riccatiStep_ilqr / backwardPass_ilqr                reference ilqrUtils.py:153-181
# ----------------------------------------------------------------------------------------
def _riccati_tail(Q, Q_x, Q_u, Q_xx, Q_uu, Q_ux):
    """ilqrUtils.py:167-170 (shared by the iLQR and DDP steps, :200-203)."""
    l = -np.linalg.solve(Q_uu, Q_u[..., None])[..., 0]
    L = -np.linalg.solve(Q_uu, Q_ux)
    Quul = _mv(Q_uu, l)
    v = Q - 0.5 * np.sum(l * Quul, -1)
    v_x = Q_x - _mv(_T(L), Quul)
    v_xx = Q_xx - (_T(L) @ Q_uu) @ L
    return QuadraticValueFunction(v, v_x, v_xx), AffinePolicy(l, L)


def riccatiStep_ilqr(dynamics, cost, value):
    """One backward step (ilqrUtils.py:153-173); ``dynamics.f`` is unused (:156)."""
    _, f_x, f_u = dynamics[:3]
    c, c_x, c_u, c_xx, c_ux, c_uu = (np.asarray(t) for t in cost)
    v, v_x, v_xx = (np.asarray(t) for t in value)
    f_x, f_u = np.asarray(f_x), np.asarray(f_u)
    Q = c + v
    Q_x = c_x + _mv(_T(f_x), v_x)
    Q_u = c_u + _mv(_T(f_u), v_x)
    Q_xx = c_xx + (_T(f_x) @ v_xx) @ f_x
    Q_uu = c_uu + (_T(f_u) @ v_xx) @ f_u
    Q_ux = c_ux + (_T(f_u) @ v_xx) @ f_x
    return _riccati_tail(Q, Q_x, Q_u, Q_xx, Q_uu, Q_ux)


def backwardPass_ilqr(dynamics: AffineDynamics, cost: QuadraticCostFunction, Vf: QuadraticValueFunction):
    """Reverse scan of riccatiStep_ilqr (ilqrUtils.py:176-181).

    Time axis: -3 for matrices, -2 for vectors, -1 for scalars c (leading batch axes allowed).
    Returns AffinePolicy(l (..., N, m), L (..., N, m, n)).
    """
    f, f_x, f_u = (np.asarray(t) for t in dynamics[:3])
    c, c_x, c_u, c_xx, c_ux, c_uu = (np.asarray(t) for t in cost)
    N = c.shape[-1]
    n, m = f_u.shape[-2:]
    V = QuadraticValueFunction(*(np.asarray(t) for t in Vf))
    lArr = np.empty(f_u.shape[:-3] + (N, m), dtype=f_x.dtype)
    LArr = np.empty(f_u.shape[:-3] + (N, m, n), dtype=f_x.dtype)
    for k in range(N - 1, -1, -1):
        dyn_k = (None, f_x[..., k, :, :], f_u[..., k, :, :])
        cost_k = (c[..., k], c_x[..., k, :], c_u[..., k, :], c_xx[..., k, :, :], c_ux[..., k, :, :],
                  c_uu[..., k, :, :])
        V, pol = riccatiStep_ilqr(dyn_k, cost_k, V)
        lArr[..., k, :] = pol.l
        LArr[..., k, :, :] = pol.L
    return AffinePolicy(lArr, LArr)


# ----------------------------------------------------------------------------------------
# A4  riccatiStep_ddp / backwardPass_ddp                  reference ilqrUtils.py:184-214
# ----------------------------------------------------------------------------------------
def riccatiStep_ddp(dynamics, cost, value):
    """ilqrUtils.py:184-206."""
    c, c_x, c_u, c_xx, c_ux, c_uu = (np.asarray(t) for t in cost)
    v, v_x, v_xx = (np.asarray(t) for t in value)
    dyn = QuadraticDynamics(*[None if t is None else np.asarray(t) for t in dynamics])
    f_x, f_u = dyn.f_x, dyn.f_u
    vf_xx, vf_ux, vf_uu = conditionQuadraticDynamics(dyn, v_x)
    Q = c + v
    Q_x = c_x + _mv(_T(f_x), v_x)
    Q_u = c_u + _mv(_T(f_u), v_x)
    Q_xx = c_xx + (_T(f_x) @ v_xx) @ f_x + vf_xx
    Q_uu = c_uu + (_T(f_u) @ v_xx) @ f_u + vf_uu
    Q_ux = c_ux + (_T(f_u) @ v_xx) @ f_x + vf_ux
    return _riccati_tail(Q, Q_x, Q_u, Q_xx, Q_uu, Q_ux)


def backwardPass_ddp(dynamics: QuadraticDynamics, cost: QuadraticCostFunction, Vf: QuadraticValueFunction):
    """Reverse scan of riccatiStep_ddp (ilqrUtils.py:209-214)."""
    f, f_x, f_u, f_xx, f_ux, f_uu = (np.asarray(t) for t in dynamics)
    c, c_x, c_u, c_xx, c_ux, c_uu = (np.asarray(t) for t in cost)
    N = c.shape[-1]
    n, m = f_u.shape[-2:]
    V = QuadraticValueFunction(*(np.asarray(t) for t in Vf))
    lArr = np.empty(f_u.shape[:-3] + (N, m), dtype=f_x.dtype)
    LArr = np.empty(f_u.shape[:-3] + (N, m, n), dtype=f_x.dtype)
    for k in range(N - 1, -1, -1):
        dyn_k = (None, f_x[..., k, :, :], f_u[..., k, :, :], f_xx[..., k, :, :, :], f_ux[..., k, :, :, :],
                 f_uu[..., k, :, :, :])
        cost_k = (c[..., k], c_x[..., k, :], c_u[..., k, :], c_xx[..., k, :, :], c_ux[..., k, :, :],
                  c_uu[..., k, :, :])
        V, pol = riccatiStep_ddp(dyn_k, cost_k, V)
        lArr[..., k, :] = pol.l
        LArr[..., k, :, :] = pol.L
    return AffinePolicy(lArr, LArr)


# ----------------------------------------------------------------------------------------
# A6  trajectoryRollout + AffinePolicy.__call__   reference ilqrUtils.py:33-66, pytrees.py:215-220
# ----------------------------------------------------------------------------------------
def trajectoryRollout(x0, dynFun: Callable, policy: AffinePolicy, trajPrev: Trajectory, alpha=1):
    """for k: u_k = alpha*l_k + L_k (x_k - xPrev_k) + uPrev_k ; x_{k+1} = dynFun(x_k, u_k).

    Single trajectory (the reference's shapes).  ``policy`` may also be a callable
    ``policy(dx, k=k, alpha=alpha)`` as in the reference's own test (test_ilqrUtils.py:7-22).
    """
    xPrev, uPrev = trajPrev
    N = np.shape(uPrev)[0]
    x = np.asarray(x0)
    xs, us = [x], []
    for k in range(N):
        dx = x - xPrev[k]
        if callable(policy):
            u = policy(dx, k=k, alpha=alpha) + uPrev[k]
        else:
            u = alpha * policy.l[k] + policy.L[k] @ dx + uPrev[k]
        x = dynFun(x, u)
        xs.append(x)
        us.append(u)
    return Trajectory(np.stack(xs), np.stack(us))


def trajectoryCost(runningCost: Callable, terminalCost: Callable, traj: Trajectory):
    """CostFunction.__call__ with k=None (pytrees.py:49-52): sum_k c(x_k,u_k) + c_f(x_N)."""
    xTraj, uTraj = traj
    J = 0.0
    for k in range(uTraj.shape[0]):
        J = J + runningCost(xTraj[k], uTraj[k])
    return J + terminalCost(xTraj[-1])


# ----------------------------------------------------------------------------------------
# A7  forwardPass2                                         reference ilqrUtils.py:116-150
# ----------------------------------------------------------------------------------------
LINESEARCH_ALPHAS = 0.5 ** np.arange(16)       # ilqrUtils.py:145


def forwardPass2(x0, dynFun, runningCost, terminalCost, policy, trajPrev, return_index=False):
    """16 rollouts at alpha = 0.5**j, take argmin of J (NaN wins, as in NumPy/JAX argmin).
    `return_index` (test diagnostics): also the winning index into LINESEARCH_ALPHAS and all 16 costs."""
    Js, trajs = [], []
    for alpha in LINESEARCH_ALPHAS:
        t = trajectoryRollout(x0, dynFun, policy, trajPrev, alpha=alpha)
        trajs.append(t)
        Js.append(trajectoryCost(runningCost, terminalCost, t))
    idx = int(np.argmin(np.asarray(Js)))
    if return_index:
        return trajs[idx], Js[idx], idx, np.asarray(Js)
    return trajs[idx], Js[idx]


# ----------------------------------------------------------------------------------------
# A10  Quadcopter model                                   reference quadcopter.py:10-144
# ----------------------------------------------------------------------------------------
QUAD_G = 9.807          # quadcopter.py:15
QUAD_MASS = 2.5         # :16
QUAD_FORCE_LIN = np.array([-0.2, -0.2, -0.3])      # :59
QUAD_FORCE_QUAD = np.array([-0.05, -0.05, -0.1])   # :60
QUAD_MOMENT_LIN = np.array([-0.1, -0.1, -0.05])    # :61


def quad_bodyToInertialRotationMatrix(phi, theta, psi):
    """quadcopter.py:23-38, reproduced as written (quirk Q4: entry [0][2] is cphi*sth*cpsi - sphi*spsi)."""
    cphi, sphi = np.cos(phi), np.sin(phi)
    cth, sth = np.cos(theta), np.sin(theta)
    cpsi, spsi = np.cos(psi), np.sin(psi)
    return np.array([
        [cth * cpsi, sphi * sth * cpsi - cphi * spsi, cphi * sth * cpsi - sphi * spsi],
        [cth * spsi, sphi * sth * spsi + cphi * cpsi, cphi * sth * spsi - sphi * cpsi],
        [-sth, sphi * cth, cphi * cth],
    ])


def quad_bodyRatesToEulerRatesRotationMatrix(phi, theta):
    """quadcopter.py:41-48."""
    sphi, cphi = np.sin(phi), np.cos(phi)
    cth, tth = np.cos(theta), np.tan(theta)
    return np.array([[1, sphi * tth, cphi * tth], [0, cphi, -sphi], [0, sphi / cth, cphi / cth]])


def quad_rigidBodyDynamics(state, control, wind_body=np.zeros(3)):
    """quadcopter.py:70-113: state [u,v,w,p,q,r,phi,theta(,...)] -> 8 derivatives (quirk Q5: reads only [0:8])."""
    state = np.asarray(state)      # (complex allowed: the oracle differentiates by the complex-step method)
    control = np.asarray(control)
    uvw, pqr = state[0:3], state[3:6]
    phi, theta = state[6:8]
    thrust, mxyz = control[0], control[1:4]
    d2xyz = np.array([-np.sin(theta), np.sin(phi) * np.cos(theta), np.cos(phi) * np.cos(theta)])
    R_rates2Eul = quad_bodyRatesToEulerRatesRotationMatrix(phi, theta)
    uvw_aero = uvw - wind_body                                                        # :64
    force_aero = QUAD_FORCE_LIN * uvw_aero + QUAD_FORCE_QUAD * uvw_aero ** 2         # :65
    moment_aero = QUAD_MOMENT_LIN * pqr                                               # :66
    force_total = QUAD_MASS * np.array([0, 0, -thrust]) + force_aero + QUAD_MASS * QUAD_G * d2xyz   # :98-100
    moment_total = mxyz + moment_aero                                                 # :102-103 (I = eye(3))
    uvwDot = (1 / QUAD_MASS) * (-np.cross(pqr, uvw) + force_total)                    # :106
    pqrDot = -np.cross(pqr, pqr) + moment_total                                       # :107
    phiThetaDot = R_rates2Eul[0:2, :] @ pqr                                           # :108
    return np.concatenate([uvwDot, pqrDot, phiThetaDot])


def quad_inertialDynamics(state, control, wind_ned=np.zeros(3)):
    """quadcopter.py:116-144: 12-state xDot = f(x, u); state [u,v,w,p,q,r,phi,theta,psi,x,y,z]."""
    state = np.asarray(state)
    uvw, pqr = state[0:3], state[3:6]
    phi, theta, psi = state[6:9]
    R_b2i = quad_bodyToInertialRotationMatrix(phi, theta, psi)
    R_rates2Eul = quad_bodyRatesToEulerRatesRotationMatrix(phi, theta)
    wind_body = R_b2i.T @ wind_ned
    xDot_rb = quad_rigidBodyDynamics(state[:9], control, wind_body=wind_body)
    psiDot = np.array([R_rates2Eul[2, :] @ pqr])
    xyzDot = R_b2i @ uvw
    return np.concatenate([xDot_rb, psiDot, xyzDot])


def quad_euler_step(dt):
    """The discrete map of the iLQR / DDP / MPC demos: x+ = x + dt * inertialDynamics(x, u) (demos/iterativeLqr.py:35)."""
    return lambda x, u: x + dt * quad_inertialDynamics(x, u)


def quadratic_costs(Q, R, Qf):
    """runningCost = x'Qx + u'Ru, terminalCost = x'Qf x (demos/iterativeLqr.py:12-13,37; no 1/2, quirk Q7)."""
    Q, R, Qf = (np.asarray(t, dtype=np.float64) for t in (Q, R, Qf))
    return (lambda x, u: x @ Q @ x + u @ R @ u), (lambda x: x @ Qf @ x)


# ----------------------------------------------------------------------------------------
# A9  linearisers                      reference pytrees.py:72-81, 100-115, 139-153
# ----------------------------------------------------------------------------------------
def jacobians(dynFun, x, u, h=1e-30):
    """f, d f/dx, d f/du at (x, u) by the complex-step method (exact to rounding for analytic dynFun) -- the role
    jax.jacobian plays in AffineDynamics.from_function (pytrees.py:139-145)."""
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    n, m = x.shape[0], u.shape[0]
    f = np.real(dynFun(x, u))
    f_x = np.empty((n, n))
    f_u = np.empty((n, m))
    for j in range(n):
        xp = x.astype(np.complex128)
        xp[j] += 1j * h
        f_x[:, j] = np.imag(dynFun(xp, u.astype(np.complex128))) / h
    for j in range(m):
        up = u.astype(np.complex128)
        up[j] += 1j * h
        f_u[:, j] = np.imag(dynFun(x.astype(np.complex128), up)) / h
    return f, f_x, f_u


def affine_dynamics_from_trajectory(dynFun, traj: Trajectory) -> AffineDynamics:
    """AffineDynamics.from_trajectory (pytrees.py:147-153): expansion at (xTraj[:-1], uTraj)."""
    xTraj, uTraj = traj
    out = [jacobians(dynFun, xTraj[k], uTraj[k]) for k in range(uTraj.shape[0])]
    return AffineDynamics(np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), np.stack([o[2] for o in out]))


def quadratic_cost_from_trajectory(Q, R, traj: Trajectory) -> QuadraticCostFunction:
    """QuadraticCostFunction.from_trajectory (pytrees.py:109-115) for c = x'Qx + u'Ru: the values autodiff returns,
    c_x = (Q+Q')x, c_u = (R+R')u, c_xx = Q+Q', c_ux = 0 (m,n), c_uu = R+R'."""
    xTraj, uTraj = traj
    N = uTraj.shape[0]
    n, m = xTraj.shape[1], uTraj.shape[1]
    x = xTraj[:-1]
    c = np.einsum('ki,ij,kj->k', x, Q, x) + np.einsum('ki,ij,kj->k', uTraj, R, uTraj)
    return QuadraticCostFunction(c, x @ (Q + Q.T).T, uTraj @ (R + R.T).T, np.tile(Q + Q.T, (N, 1, 1)),
                                 np.zeros((N, m, n)), np.tile(R + R.T, (N, 1, 1)))


def terminal_value_function(Qf, xf) -> QuadraticValueFunction:
    """QuadraticValueFunction.fromTerminalCostFunction (pytrees.py:72-81) for c_f = x'Qf x."""
    return QuadraticValueFunction(xf @ Qf @ xf, (Qf + Qf.T) @ xf, Qf + Qf.T)


# ----------------------------------------------------------------------------------------
# A8  iterativeLqr                                         reference ilqrUtils.py:260-327
# ----------------------------------------------------------------------------------------
def iterativeLqr(dynFun, Q, R, Qf, x0, uGuess, maxIter=100, tol=1e-3, return_iters=False, trace=None):
    """iLQR loop of ilqrUtils.py:290-327 for an analytic `dynFun` and the quadratic cost (Q, R, Qf).
    Returns (Trajectory, L (N,m,n), J, converged).  `trace` (a list, test diagnostics): receives per iteration
    (J_new, winning step-size index, the 16 costs of the line search)."""
    Q, R, Qf = (np.asarray(t, dtype=np.float64) for t in (Q, R, Qf))
    x0 = np.asarray(x0, dtype=np.float64)
    uGuess = np.asarray(uGuess, dtype=np.float64)
    n = x0.shape[0]
    N, m = uGuess.shape
    runningCost, terminalCost = quadratic_costs(Q, R, Qf)
    policy = AffinePolicy(uGuess, np.zeros((N, m, n)))                                   # :293
    traj_prev = Trajectory(np.zeros((N + 1, n)), np.zeros((N, m)))                        # :294
    traj = trajectoryRollout(x0, dynFun, policy, traj_prev)                               # :297
    J = trajectoryCost(runningCost, terminalCost, traj)                                   # :298
    converged, it = False, 0
    while (not converged) and it < maxIter:                                               # :301-303
        dyn = affine_dynamics_from_trajectory(dynFun, traj)                               # :308
        cost = quadratic_cost_from_trajectory(Q, R, traj)                                 # :309
        Vf = terminal_value_function(Qf, traj.xTraj[-1])                                  # :310
        cost = conditionQuadraticCost(cost)                                               # :312
        Vf = conditionValueFunction(Vf)                                                   # :313
        policy = backwardPass_ilqr(dyn, cost, Vf)                                         # :315
        traj_new, J_new, idx, Js = forwardPass2(x0, dynFun, runningCost, terminalCost, policy, traj, return_index=True)   # :316
        if trace is not None:
            trace.append((J_new, idx, Js))
        converged = bool(abs(J - J_new) <= tol)                                           # :318
        traj, J = traj_new, J_new
        it += 1
    out = (traj, policy.L, J, converged)
    return out + (it,) if return_iters else out


# ----------------------------------------------------------------------------------------
# A9 (second order) QuadraticDynamics.from_trajectory            reference pytrees.py:180-194
# ----------------------------------------------------------------------------------------
def quad_euler_step_torch(dt):
    """torch (CPU, fp64) restatement of x+ = x + dt*inertialDynamics(x,u) (quadcopter.py:23-144), used only to obtain exact
    second derivatives by autograd -- the role jax.hessian plays in QuadraticDynamics.from_function (pytrees.py:180-186)."""
    import torch

    def f(x, u):
        uvw, pqr = x[0:3], x[3:6]
        phi, theta, psi = x[6], x[7], x[8]
        cphi, sphi = torch.cos(phi), torch.sin(phi)
        cth, sth, tth = torch.cos(theta), torch.sin(theta), torch.tan(theta)
        cpsi, spsi = torch.cos(psi), torch.sin(psi)
        R = torch.stack([
            torch.stack([cth * cpsi, sphi * sth * cpsi - cphi * spsi, cphi * sth * cpsi - sphi * spsi]),
            torch.stack([cth * spsi, sphi * sth * spsi + cphi * cpsi, cphi * sth * spsi - sphi * cpsi]),
            torch.stack([-sth, sphi * cth, cphi * cth])])
        one, zero = torch.ones_like(phi), torch.zeros_like(phi)
        E = torch.stack([torch.stack([one, sphi * tth, cphi * tth]), torch.stack([zero, cphi, -sphi]),
                         torch.stack([zero, sphi / cth, cphi / cth])])
        flin = torch.tensor(QUAD_FORCE_LIN, dtype=x.dtype, device=x.device)
        fquad = torch.tensor(QUAD_FORCE_QUAD, dtype=x.dtype, device=x.device)
        mlin = torch.tensor(QUAD_MOMENT_LIN, dtype=x.dtype, device=x.device)
        force_aero = flin * uvw + fquad * uvw ** 2
        d2xyz = torch.stack([-sth, sphi * cth, cphi * cth])
        force_total = QUAD_MASS * torch.stack([zero, zero, -u[0]]) + force_aero + QUAD_MASS * QUAD_G * d2xyz
        uvwDot = (1 / QUAD_MASS) * (-torch.linalg.cross(pqr, uvw) + force_total)
        pqrDot = u[1:4] + mlin * pqr
        eul = E @ pqr
        xyzDot = R @ uvw
        xd = torch.cat([uvwDot, pqrDot, eul, xyzDot])
        return x + dt * xd
    return f


def quadratic_dynamics_from_trajectory(f_torch, traj: Trajectory) -> QuadraticDynamics:
    """QuadraticDynamics.from_trajectory (pytrees.py:188-194) by torch autograd on CPU:
    f_xx[i,j,k] = d2f_i/dx_j dx_k, f_ux[i,j,k] = d2f_i/du_j dx_k, f_uu[i,j,k] = d2f_i/du_j du_k."""
    import torch
    from torch.func import hessian, jacfwd, vmap
    xT, uT = (torch.as_tensor(np.asarray(t), dtype=torch.float64) for t in traj)
    xs = xT[:uT.shape[0]]
    # vmap over the time steps, as the reference does (pytrees.py:193: jax.vmap(QuadraticDynamics.from_function, ...))
    fs = vmap(f_torch)(xs, uT)
    fx, fu = vmap(jacfwd(f_torch, argnums=(0, 1)))(xs, uT)
    (fxx, _fxu), (fux, fuu) = vmap(hessian(f_torch, argnums=(0, 1)))(xs, uT)
    return QuadraticDynamics(*(t.numpy() for t in (fs, fx, fu, fxx, fux, fuu)))


# ----------------------------------------------------------------------------------------
# A8 (DDP) differentialDynamicProgramming                 reference ilqrUtils.py:330-397
# ----------------------------------------------------------------------------------------
def differentialDynamicProgramming(dynFun, f_torch, Q, R, Qf, x0, uGuess, maxIter=100, tol=1e-3, return_iters=False, trace=None):
    """DDP loop of ilqrUtils.py:360-397: as iterativeLqr with QuadraticDynamics (:365) and backwardPass_ddp (:373).
    `dynFun` (NumPy) rolls out, `f_torch` (the same map in torch) supplies the first and second derivatives."""
    Q, R, Qf = (np.asarray(t, dtype=np.float64) for t in (Q, R, Qf))
    x0 = np.asarray(x0, dtype=np.float64)
    uGuess = np.asarray(uGuess, dtype=np.float64)
    n = x0.shape[0]
    N, m = uGuess.shape
    runningCost, terminalCost = quadratic_costs(Q, R, Qf)
    policy = AffinePolicy(uGuess, np.zeros((N, m, n)))
    traj = trajectoryRollout(x0, dynFun, policy, Trajectory(np.zeros((N + 1, n)), np.zeros((N, m))))
    J = trajectoryCost(runningCost, terminalCost, traj)
    converged, it = False, 0
    while (not converged) and it < maxIter:
        dyn = quadratic_dynamics_from_trajectory(f_torch, traj)
        cost = conditionQuadraticCost(quadratic_cost_from_trajectory(Q, R, traj))
        Vf = conditionValueFunction(terminal_value_function(Qf, traj.xTraj[-1]))
        policy = backwardPass_ddp(dyn, cost, Vf)
        traj_new, J_new, idx, Js = forwardPass2(x0, dynFun, runningCost, terminalCost, policy, traj, return_index=True)
        if trace is not None:
            trace.append((J_new, idx, Js))
        converged = bool(abs(J - J_new) <= tol)
        traj, J = traj_new, J_new
        it += 1
    out = (traj, policy.L, J, converged)
    return out + (it,) if return_iters else out


def discreteInfiniteHorizonLqr(A, B, Q, R):
    """lqrUtils.py:202-203 verbatim in meaning: V = scipy.linalg.solve_discrete_are(A, B, Q, R);
    L = solve(R + B^T V B, B^T V A).  SciPy is the library the reference itself calls (pinned 1.16.2; 1.15.3 here)."""
    import scipy.linalg as spl
    A, B, Q, R = (np.asarray(x, dtype=np.float64) for x in (A, B, Q, R))
    V = spl.solve_discrete_are(A, B, Q, R)
    return np.linalg.solve(R + B.T @ V @ B, B.T @ V @ A), V


def infiniteHorizonLqr(A, B, Q, R):
    """lqrUtils.py:34-36 verbatim in meaning: P = scipy.linalg.solve_continuous_are(A, B, Q, R); K = solve(R, B^T P).
    SciPy is the library the reference itself calls."""
    import scipy.linalg as spl
    A, B, Q, R = (np.asarray(x, dtype=np.float64) for x in (A, B, Q, R))
    P = spl.solve_continuous_are(A, B, Q, R)
    return np.linalg.solve(R, B.T @ P), P


def infiniteHorizonIntegralLqr(A, B, Q, R, Qi, Ci):
    """lqrUtils.py:125-141: integral-augmented system [[0, Ci], [0, A]], [[0], [B]], blkdiag(Qi, Q) through infiniteHorizonLqr."""
    import scipy.linalg as spl
    A, B, Q, R, Qi, Ci = (np.asarray(x, dtype=np.float64) for x in (A, B, Q, R, Qi, Ci))
    n_i = Qi.shape[0]
    n_x, n_u = B.shape
    Aw = np.block([[np.zeros((n_i, n_i)), Ci], [np.zeros((n_x, n_i)), A]])
    Bw = np.vstack([np.zeros((n_i, n_u)), B])
    Qw = spl.block_diag(Qi, Q)
    K, _ = infiniteHorizonLqr(Aw, Bw, Qw, R)
    return K[:, :n_i], K[:, n_i:]


def lqrHjb(t, V, A, B, Q, R_inv, n):
    """lqrUtils.py:39-52 (_lqrHjb): dV = -Q + V B R_inv B^T V - V A - A^T V, flattened."""
    V = np.asarray(V, dtype=np.float64).reshape((n, n))
    dV = -Q(t) + V @ B(t) @ R_inv(t) @ B(t).T @ V - V @ A(t) - A(t).T @ V
    return dV.reshape(-1)


def finiteHorizonLqr(A, B, Q, R_inv, Qf, T, N=50, rtol=1e-12, atol=1e-14):
    """lqrUtils.py:85-97: integrate dV/ds = -lqrHjb(T - s, V) from V(0) = Qf over s in linspace(0, T, N), reverse, and
    interpolate linearly (jaxUtils.py:7-24: jnp.interp per component, clipped at the ends).  The reference integrates with
    jax.experimental.ode.odeint (Dormand-Prince 5(4), rtol = atol = 1.4e-8); jax is absent here, so this restatement
    integrates the same ODE with SciPy's DOP853 at tight tolerances -- the two agree to the reference's own tolerance.
    Returns (K, t, V) with K(t) = R_inv(t) @ B(t).T @ V(t), t (N,), V (N, n, n)."""
    from scipy.integrate import solve_ivp
    Qf = np.asarray(Qf, dtype=np.float64)
    n = np.asarray(A(0)).shape[0]
    t = np.linspace(0, T, num=N)
    sol = solve_ivp(lambda s, V: -lqrHjb(T - s, V, A, B, Q, R_inv, n), (0.0, float(T)), Qf.reshape(-1), method="DOP853",
                    t_eval=t, rtol=rtol, atol=atol)
    assert sol.success
    V = sol.y.T[::-1].reshape(N, n, n)          # V[j] is the value at time t[j]

    def Vfun(tq):
        return np.stack([np.interp(tq, t, V[:, i, j]) for i in range(n) for j in range(n)]).reshape(n, n)

    return (lambda tq: R_inv(tq) @ B(tq).T @ Vfun(tq)), t, V
