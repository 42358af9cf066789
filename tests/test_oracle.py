"""Pin the CPU oracle: the reference's own known-answer tests + SciPy DARE + C-vs-NumPy agreement.

CPU only.  Every expected value here comes from the reference's tests (file:line in each
docstring; data in tests/golden/reference_kats.json) or from SciPy, never from the oracle itself.
"""
import json
import os

import numpy as np
import pytest
import scipy.linalg as spl

from oracle import c_oracle
from oracle import zopt_oracle as zo
from tests import problems

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
I2 = np.eye(2)


def _tile(x, N):
    return np.repeat(np.asarray(x, dtype=np.float64)[None], N, axis=0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kat_discreteFiniteHorizonLqr(dtype):
    """reference tests/test_lqrUtils.py:61-69 -- L[1] = 0.5 I, L[0] = 0.6 I."""
    k = KATS["A1_discreteFiniteHorizonLqr"]
    N = k["N"]
    A = B = Q = R = _tile(I2, N).astype(dtype)
    L = zo.discreteFiniteHorizonLqr(A, B, Q, R, N)
    assert L.dtype == dtype and L.shape == (N, 2, 2)
    assert L == pytest.approx(np.array(k["L"]), rel=1e-6)


def test_kat_discreteFiniteHorizonLqr_c_oracle():
    k = KATS["A1_discreteFiniteHorizonLqr"]
    N = k["N"]
    A = B = Q = R = _tile(I2, N)[None]
    L = c_oracle.lqr_backward(A, B, Q, R)
    assert L[0] == pytest.approx(np.array(k["L"]), rel=1e-12)


def test_kat_bilinearAffineLqr():
    """reference tests/test_lqrUtils.py:82-98 -- L = I both steps, l[1] = 1.5, l[0] = 1."""
    k = KATS["A2_bilinearAffineLqr"]
    N = k["N"]
    A = B = Q = R = H = _tile(I2, N)
    d = q = r = _tile(np.ones(2), N)
    q0 = np.ones(N)
    L, l = zo.bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, N)
    assert L == pytest.approx(np.array(k["L"]), rel=1e-6)
    assert l == pytest.approx(np.array(k["l"]), rel=1e-6)


def _kat_step_inputs():
    dyn = (np.zeros(2), I2, I2)
    cost = (0.0, np.zeros(2), np.zeros(2), I2, np.zeros((2, 2)), I2)
    value = (0.0, np.zeros(2), I2)
    return dyn, cost, value


def test_kat_riccatiStep_ilqr():
    """reference tests/test_ilqrUtils.py:56-81 -- exact ==."""
    k = KATS["A3_riccatiStep_ilqr"]
    dyn, cost, value = _kat_step_inputs()
    vout, pol = zo.riccatiStep_ilqr(dyn, cost, value)
    assert vout.v == k["out_v"]
    assert np.all(vout.v_x == np.array(k["out_v_x"]))
    assert np.all(vout.v_xx == np.array(k["out_v_xx"]))
    assert np.all(pol.l == np.array(k["l"]))
    assert np.all(pol.L == np.array(k["L"]))


def test_kat_riccatiStep_ddp():
    """reference tests/test_ilqrUtils.py:110-135 -- rel 1e-3 (the PD clamp adds 1e-3 I)."""
    k = KATS["A4_riccatiStep_ddp"]
    dyn, cost, value = _kat_step_inputs()
    z = np.zeros((2, 2, 2))
    vout, pol = zo.riccatiStep_ddp(dyn + (z, z, z), cost, value)
    assert vout.v == k["out_v"]
    assert vout.v_x == pytest.approx(np.array(k["out_v_x"]))
    assert vout.v_xx == pytest.approx(np.array(k["out_v_xx"]), rel=1e-3)
    assert pol.l == pytest.approx(np.array(k["l"]))
    assert pol.L == pytest.approx(np.array(k["L"]), rel=1e-3)
    # the clamp really is there: v_xx != 1.5 I exactly
    assert not np.all(vout.v_xx == 1.5 * I2)


def test_kat_trajectoryRollout():
    """reference tests/test_ilqrUtils.py:7-22 -- integer dynamics, incl. alpha = 0.5."""
    k = KATS["A6_trajectoryRollout"]
    N = k["N"]
    dynFun = lambda x, u: x + u
    policy = lambda x, k, alpha: np.array([alpha * k])
    trajPrev = (np.zeros(N), np.zeros(N))
    x0 = np.array(k["x0"])
    xT, uT = zo.trajectoryRollout(x0, dynFun, policy, trajPrev)
    assert np.all(xT == np.array(k["alpha1"]["xTraj"])[:, None])
    assert np.all(uT == np.array(k["alpha1"]["uTraj"])[:, None])
    xT, uT = zo.trajectoryRollout(x0, dynFun, policy, trajPrev, alpha=0.5)
    assert np.all(xT == np.array(k["alpha0.5"]["xTraj"])[:, None])
    assert np.all(uT == np.array(k["alpha0.5"]["uTraj"])[:, None])


def test_backward_passes_match_single_steps():
    """backwardPass_* is a reverse scan of the step (ilqrUtils.py:176-181, 209-214)."""
    rng = np.random.default_rng(5)
    N, n, m = 4, 3, 2
    f_x = rng.standard_normal((N, n, n)); f_u = rng.standard_normal((N, n, m))
    M = rng.standard_normal((N, n + m, n + m)); H = M @ np.swapaxes(M, -1, -2) + np.eye(n + m)
    cost = zo.QuadraticCostFunction(rng.standard_normal(N), rng.standard_normal((N, n)), rng.standard_normal((N, m)),
                                    H[:, :n, :n], H[:, n:, :n], H[:, n:, n:])
    Vf = zo.QuadraticValueFunction(0.3, rng.standard_normal(n), np.eye(n))
    pol = zo.backwardPass_ilqr(zo.AffineDynamics(np.zeros((N, n)), f_x, f_u), cost, Vf)
    V = Vf
    for k in range(N - 1, -1, -1):
        V, p = zo.riccatiStep_ilqr((None, f_x[k], f_u[k]), tuple(t[k] for t in cost), V)
        assert np.array_equal(p.L, pol.L[k]) and np.array_equal(p.l, pol.l[k])


def test_ensurePositiveDefinite_semantics():
    """ilqrUtils.py:217-219: spectral clamp at eps; jnp.linalg.eigh symmetrises its input."""
    a = np.diag([2.0, -1.0, 1e-5])
    out = zo.ensurePositiveDefinite(a)
    assert out == pytest.approx(np.diag([2.0, 1e-3, 1e-3]), abs=1e-15)
    rng = np.random.default_rng(0)
    M = rng.standard_normal((5, 5))
    assert zo.ensurePositiveDefinite(M) == pytest.approx(zo.ensurePositiveDefinite(0.5 * (M + M.T)), abs=1e-13)
    S = M @ M.T + np.eye(5)      # already PD with min eig >= 1: unchanged
    assert zo.ensurePositiveDefinite(S) == pytest.approx(S, rel=1e-12)


def test_dare_known_answer_and_limit():
    """tests/test_lqrUtils.py:72-79 (golden-ratio gain) + long-horizon A1 converges to the SciPy DARE gain."""
    g = (1 + np.sqrt(5)) / (3 + np.sqrt(5))
    N = 60
    A = B = Q = R = _tile(I2, N)
    L = zo.discreteFiniteHorizonLqr(A, B, Q, R, N)
    assert L[0] == pytest.approx(g * I2, rel=1e-12)
    for (n, m, T) in [(12, 4, 50), (4, 1, 50), (8, 4, 100)]:
        A1, B1, Q1, R1 = problems.random_lti_systems(3, n, m, seed=11)
        At, Bt, Qt, Rt = problems.tile_over_horizon(A1, B1, Q1, R1, 400)
        L = zo.discreteFiniteHorizonLqr(At, Bt, Qt, Rt, 400)
        for i in range(3):
            V = spl.solve_discrete_are(A1[i], B1[i], Q1[i], R1[i])
            Ld = np.linalg.solve(R1[i] + B1[i].T @ V @ B1[i], B1[i].T @ V @ A1[i])
            assert L[i, 0] == pytest.approx(Ld, rel=1e-9, abs=1e-11)


@pytest.mark.parametrize("n,m,T", [(12, 4, 50), (4, 1, 50), (8, 4, 100), (2, 2, 3), (7, 3, 9)])
def test_c_oracle_matches_numpy_oracle(n, m, T):
    A, B, Q, R = problems.random_time_varying(6, T, n, m, seed=n * 100 + m)
    L_np = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
    L_c = c_oracle.lqr_backward(A, B, Q, R)
    scale = np.max(np.abs(L_np))
    assert np.max(np.abs(L_c - L_np)) <= 1e-12 * scale


def test_batch_axes_equal_loop_over_trajectories():
    A, B, Q, R = problems.random_time_varying(4, 10, 5, 2, seed=3)
    Lb = zo.discreteFiniteHorizonLqr(A, B, Q, R, 10)
    for i in range(4):
        Li = zo.discreteFiniteHorizonLqr(A[i], B[i], Q[i], R[i], 10)
        assert np.allclose(Li, Lb[i], rtol=1e-13, atol=1e-15)


def test_kat_quadcopter_model():
    """reference tests/test_quadcopter.py:12-86 (rotation matrices, rigid-body and inertial dynamics)."""
    assert zo.quad_bodyToInertialRotationMatrix(0, 0, 0) == pytest.approx(np.eye(3))
    th = np.pi / 6
    cth, sth, tth = np.cos(th), np.sin(th), np.tan(th)
    assert zo.quad_bodyToInertialRotationMatrix(th, 0, 0) == pytest.approx(np.array([[1, 0, 0], [0, cth, -sth], [0, sth, cth]]))
    assert zo.quad_bodyToInertialRotationMatrix(0, th, 0) == pytest.approx(np.array([[cth, 0, sth], [0, 1, 0], [-sth, 0, cth]]))
    assert zo.quad_bodyToInertialRotationMatrix(0, 0, th) == pytest.approx(np.array([[cth, -sth, 0], [sth, cth, 0], [0, 0, 1]]))
    assert zo.quad_bodyRatesToEulerRatesRotationMatrix(0, 0) == pytest.approx(np.eye(3))
    assert zo.quad_bodyRatesToEulerRatesRotationMatrix(th, 0) == pytest.approx(np.array([[1, 0, 0], [0, cth, -sth], [0, sth, cth]]))
    assert zo.quad_bodyRatesToEulerRatesRotationMatrix(0, th) == pytest.approx(np.array([[1, 0, tth], [0, 1, 0], [0, 0, 1 / cth]]))
    k = KATS["A10_quadcopter"]
    assert zo.quad_rigidBodyDynamics(np.zeros(9), np.zeros(4)) == pytest.approx(np.array(k["rigidBody_rest_zero_thrust"]["xDot"]))
    hover = np.array([9.807, 0, 0, 0])
    assert zo.quad_rigidBodyDynamics(np.zeros(9), hover) == pytest.approx(np.zeros(8))
    assert zo.quad_inertialDynamics(np.zeros(12), hover) == pytest.approx(np.zeros(12))
    s = np.zeros(12); s[0:3] = [0.1, 0.2, 0.3]
    assert zo.quad_inertialDynamics(s, hover)[9:] == pytest.approx(np.array([0.1, 0.2, 0.3]))
    s[8] = np.pi / 2
    assert zo.quad_inertialDynamics(s, hover)[9:] == pytest.approx(np.array([-0.2, 0.1, 0.3]))
    # quirk Q4 is reproduced: entry [0][2] of the body-to-inertial matrix for a general attitude
    R = zo.quad_bodyToInertialRotationMatrix(0.3, 0.2, 0.1)
    assert R[0, 2] == pytest.approx(np.cos(0.3) * np.sin(0.2) * np.cos(0.1) - np.sin(0.3) * np.sin(0.1))


def test_complex_step_jacobians_match_finite_differences():
    """The oracle's lineariser (complex step) against central differences on the quadcopter Euler map."""
    rng = np.random.default_rng(3)
    step = zo.quad_euler_step(0.1)
    x = 0.3 * rng.standard_normal(12)
    u = np.array([9.807, 0, 0, 0]) + 0.3 * rng.standard_normal(4)
    f, f_x, f_u = zo.jacobians(step, x, u)
    assert f == pytest.approx(step(x, u), rel=1e-15)
    h = 1e-6
    for j in range(12):
        e = np.zeros(12); e[j] = h
        assert f_x[:, j] == pytest.approx((step(x + e, u) - step(x - e, u)) / (2 * h), abs=1e-8)
    for j in range(4):
        e = np.zeros(4); e[j] = h
        assert f_u[:, j] == pytest.approx((step(x, u + e) - step(x, u - e)) / (2 * h), abs=1e-8)


def test_kat_iterativeLqr_converges_to_riccati():
    """reference tests/test_ilqrUtils.py:167-181 (A=B=Q=R=I, N=3, x0=(2,1)) asserts `converged`; the problem is LQ, so
    the iLQR trajectory must in addition be the Riccati-optimal one (hand check via the LQR gains)."""
    I = np.eye(2)
    f = lambda x, u: I @ x + I @ u
    x0 = np.array([2.0, 1.0])
    traj, L, J, converged, iters = zo.iterativeLqr(f, I, I, I, x0, np.zeros((3, 2)), return_iters=True)
    assert converged and iters <= 3
    K = zo.discreteFiniteHorizonLqr(_tile(I2, 4), _tile(I2, 4), _tile(I2, 4), _tile(I2, 4), 4)[1:]   # stage gains with terminal Q
    x = x0.copy()
    for k in range(3):
        u = -K[k] @ x
        assert traj.uTraj[k] == pytest.approx(u, abs=1e-9)
        x = x + u
    assert traj.xTraj[-1] == pytest.approx(x, abs=1e-9)
    assert L == pytest.approx(-K, abs=1e-9)


def test_dare_oracle_known_answer():
    """reference tests/test_lqrUtils.py:72-79 (the oracle is the same SciPy call the reference makes, lqrUtils.py:202-203)."""
    I = np.eye(2)
    L, V = zo.discreteInfiniteHorizonLqr(I, I, I, I)
    assert L == pytest.approx((1 + np.sqrt(5)) / (3 + np.sqrt(5)) * np.eye(2))
    assert V == pytest.approx((1 + np.sqrt(5)) / 2 * np.eye(2))


def test_continuous_lqr_oracle_known_answers():
    """reference tests/test_lqrUtils.py:8-15 (K = (1 + sqrt 2) I), :18-28 (_lqrHjb), :31-44 (finiteHorizonLqr: K(T) = I and the
    analytic scalar Riccati solution at t = 0), :47-58 (integral LQR: Ki = [[1],[0]], Kp = diag(3, 1 + sqrt 2))."""
    K, P = zo.infiniteHorizonLqr(I2, I2, I2, I2)
    assert K == pytest.approx((1 + np.sqrt(2)) * I2) and P == pytest.approx((1 + np.sqrt(2)) * I2)
    k = KATS["CARE_infiniteHorizonIntegralLqr"]
    Ki, Kp = zo.infiniteHorizonIntegralLqr(I2, I2, I2, I2, np.array(k["Qi"]), np.array(k["Ci"]))
    assert Ki == pytest.approx(np.array(k["Ki"]), abs=1e-12)
    assert Kp == pytest.approx(np.diag([3, 1 + np.sqrt(2)]))
    c = lambda t: I2
    assert zo.lqrHjb(KATS["ODE_lqrHjb"]["t"], I2, c, c, c, c, 2) == pytest.approx(np.array(KATS["ODE_lqrHjb"]["dV_flat"]))
    f = KATS["ODE_finiteHorizonLqr"]
    Kf, t, V = zo.finiteHorizonLqr(c, c, c, c, I2, f["T"], N=f["N"])
    assert Kf(f["T"]) == pytest.approx(I2)
    s2 = np.sqrt(2)
    K_exp = lambda tq: ((1 + s2) * np.exp(2 * s2) - (s2 - 1) * np.exp(2 * s2 * tq)) / (np.exp(2 * s2 * tq) + np.exp(2 * s2))
    assert Kf(0) == pytest.approx(K_exp(0) * I2, rel=f["K(0)_rel"])
    assert Kf(0) == pytest.approx(K_exp(0) * I2, rel=1e-10)
    assert V[1] == pytest.approx(K_exp(t[1]) * I2, rel=1e-10)
