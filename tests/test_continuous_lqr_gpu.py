"""GPU parity tests for the continuous-time LQR designs:
infiniteHorizonLqr / infiniteHorizonIntegralLqr (reference lqrUtils.py:13-36, 101-141) and finiteHorizonLqr (:39-98).

zopt_amd.lqrUtils -> ctypes -> C ABI -> HIP kernels (care.hip); the oracle (SciPy: solve_continuous_are -- the library call
the reference itself makes -- and DOP853 for the Riccati ODE) is only the checker.
Tolerances: CARE 1e-9 relative on well-conditioned designs (measured ~1e-13); Riccati ODE 2e-6 relative against a 1e-12
integration (the kernel controls its local error at the reference integrator's rtol = atol = 1.4e-8)."""
import json
import os

import numpy as np
import pytest

from oracle import zopt_oracle as zo

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
I2 = np.eye(2)


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def lqr():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import lqrUtils
    return lqrUtils


def _designs(batch, n, m, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    A = scale * rng.standard_normal((batch, n, n))
    B = rng.standard_normal((batch, n, m))
    M = rng.standard_normal((batch, n, n))
    Q = M @ np.swapaxes(M, -1, -2) / n + 0.1 * np.eye(n)
    M = rng.standard_normal((batch, m, m))
    R = M @ np.swapaxes(M, -1, -2) / m + 0.5 * np.eye(m)
    return A, B, Q, R


def test_kat_infiniteHorizonLqr(lqr):
    """reference tests/test_lqrUtils.py:8-15: K = (1 + sqrt 2) I."""
    assert "1+sqrt(2)" == KATS["CARE_infiniteHorizonLqr"]["K_scale_of_I2"]
    K = lqr.infiniteHorizonLqr(I2, I2, I2, I2)
    assert K.shape == (2, 2)
    assert K == pytest.approx((1 + np.sqrt(2)) * I2, rel=1e-13, abs=1e-14)


def test_kat_infiniteHorizonIntegralLqr(lqr):
    """reference tests/test_lqrUtils.py:47-58 (Ci passed 1-D as there): Ki = [[1],[0]], Kp = diag(3, 1 + sqrt 2)."""
    k = KATS["CARE_infiniteHorizonIntegralLqr"]
    Ki, Kp = lqr.infiniteHorizonIntegralLqr(I2, I2, I2, I2, np.array(k["Qi"]), np.array(k["Ci"]))
    assert Ki.shape == (2, 1) and Kp.shape == (2, 2)
    assert Ki == pytest.approx(np.array(k["Ki"]), abs=1e-12)
    assert Kp == pytest.approx(np.diag([3, 1 + np.sqrt(2)]), abs=1e-12)


@pytest.mark.parametrize("n,m,scale", [(12, 4, 0.3), (8, 4, 1.0), (2, 2, 1.0), (1, 1, 2.0), (16, 16, 0.5), (16, 4, 0.3), (5, 3, 1.0),
                                       (15, 6, 0.3), (9, 2, 0.5), (3, 3, 5.0)])
def test_care_parity_random(lqr, n, m, scale):
    batch = 6
    A, B, Q, R = _designs(batch, n, m, seed=17 * n + m, scale=scale)
    K, P, it = lqr.infiniteHorizonLqr(A, B, Q, R, return_value=True)
    assert K.shape == (batch, m, n) and P.shape == (batch, n, n)
    for b in range(batch):
        Kr, Pr = zo.infiniteHorizonLqr(A[b], B[b], Q[b], R[b])
        assert _rel(P[b], Pr) <= 1e-9 and _rel(K[b], Kr) <= 1e-9
        res = A[b].T @ P[b] + P[b] @ A[b] - P[b] @ B[b] @ np.linalg.solve(R[b], B[b].T) @ P[b] + Q[b]
        assert np.max(np.abs(res)) <= 1e-10 * max(np.max(np.abs(P[b])), 1.0) * max(np.max(np.abs(A[b])), 1.0)
        assert np.all(np.linalg.eigvals(A[b] - B[b] @ K[b]).real < 0)        # the stabilising solution
    assert np.all(it > 0) and np.all(it <= 30)


def test_care_ill_conditioned_single_input(lqr):
    """Weakly controllable single-input designs (|P| ~ 1e6): P itself is only determined to ~cond * eps, so the check is
    the Riccati residual, which must be as small as SciPy's own (within a factor 10), plus a loose direct comparison."""
    batch, n, m = 6, 9, 1
    A, B, Q, R = _designs(batch, n, m, seed=17 * n + m, scale=0.1)
    K, P, it = lqr.infiniteHorizonLqr(A, B, Q, R, return_value=True)
    for b in range(batch):
        Kr, Pr = zo.infiniteHorizonLqr(A[b], B[b], Q[b], R[b])
        G = B[b] @ np.linalg.solve(R[b], B[b].T)
        res = lambda X: np.max(np.abs(A[b].T @ X + X @ A[b] - X @ G @ X + Q[b])) / np.max(np.abs(X))
        assert res(P[b]) <= 10 * res(Pr) + 1e-12
        assert _rel(P[b], Pr) <= 1e-6 and _rel(K[b], Kr) <= 1e-6
        assert np.all(np.linalg.eigvals(A[b] - B[b] @ K[b]).real < 0)


def test_care_quadcopter_hover_and_integral(lqr):
    """The reference's demo designs (demos/infiniteHorizonLqrControl.py:20-26, demos/integralLqrControl.py:25-38): the hover
    linearisation of the 8-state rigid-body model (zero eigenvalues), plain and with three integral states."""
    from zopt_amd import models
    ac = models.Quadcopter()
    xTrim, uTrim = ac.trim(np.zeros(3))
    A, B = ac.linearize(xTrim, uTrim)
    A, B = np.asarray(A), np.asarray(B)
    n, m = B.shape
    Q, R = np.eye(n), np.eye(m)
    K = lqr.infiniteHorizonLqr(A, B, Q, R)
    Kr, _ = zo.infiniteHorizonLqr(A, B, Q, R)
    assert _rel(K, Kr) <= 1e-9
    Ci = np.zeros((3, n))
    Ci[0, 0] = Ci[1, 1] = Ci[2, 2] = 1.0
    Qi = 0.5 * np.eye(3)
    Ki, Kp = lqr.infiniteHorizonIntegralLqr(A, B, Q, R, Qi, Ci)
    Kir, Kpr = zo.infiniteHorizonIntegralLqr(A, B, Q, R, Qi, Ci)
    assert Ki.shape == (m, 3) and Kp.shape == (m, n)
    assert _rel(Ki, Kir) <= 1e-8 and _rel(Kp, Kpr) <= 1e-8


def test_care_errors_and_torch(lqr):
    import torch
    A, B, Q, R = _designs(3, 4, 2, seed=5)
    with pytest.raises(ValueError):
        lqr.infiniteHorizonLqr(A, B, Q + np.triu(np.ones((4, 4)), 1), R)        # SciPy: "should be symmetric/hermitian"
    with pytest.raises(ValueError):
        lqr.infiniteHorizonLqr(A, B[:, :3], Q, R)
    with pytest.raises(ValueError):
        lqr.infiniteHorizonLqr(np.zeros((17, 17)), np.zeros((17, 1)), np.eye(17), np.eye(1))
    with pytest.raises(np.linalg.LinAlgError):                                  # unstable and uncontrollable: no solution
        lqr.infiniteHorizonLqr(np.eye(2), np.zeros((2, 1)), np.eye(2), np.eye(1))
    t = [torch.as_tensor(X, device="cuda") for X in (A, B, Q, R)]
    K = lqr.infiniteHorizonLqr(*t)
    assert K.is_cuda and K.shape == (3, 2, 4)
    assert _rel(K[1].cpu().numpy(), zo.infiniteHorizonLqr(A[1], B[1], Q[1], R[1])[0]) <= 1e-9
    assert lqr.infiniteHorizonLqr(A[:0], B[:0], Q[:0], R[:0]).shape == (0, 2, 4)


def test_kat_finiteHorizonLqr(lqr):
    """reference tests/test_lqrUtils.py:31-44: K(T) = I from V(T) = Qf; K(0) matches the analytic scalar Riccati solution."""
    c = lambda t: I2
    K = lqr.finiteHorizonLqr(c, c, c, c, I2, 1, N=4)
    assert K(1) == pytest.approx(I2, abs=1e-15)
    s2 = np.sqrt(2)
    K_exp = lambda t: ((1 + s2) * np.exp(2 * s2) - (s2 - 1) * np.exp(2 * s2 * t)) / (np.exp(2 * s2 * t) + np.exp(2 * s2))
    assert K(0) == pytest.approx(K_exp(0) * I2, rel=KATS["ODE_finiteHorizonLqr"]["K(0)_rel"])
    assert K(0) == pytest.approx(K_exp(0) * I2, rel=1e-7, abs=1e-9)            # the grid point itself: integrator accuracy
    for j, tj in enumerate(K.t):                                                  # every output time is exact up to rtol
        assert K.V[j] == pytest.approx(K_exp(tj) * I2, rel=1e-7, abs=1e-9)
    # between grid points: linear interpolation of V, clipped outside [0, T]   (jaxUtils.py:7-24)
    assert K(0.5) == pytest.approx(0.5 * (K.V[1] + K.V[2]), rel=1e-14)
    assert K(-1.0) == pytest.approx(K(0.0)) and K(7.0) == pytest.approx(K(1.0))
    assert zo.lqrHjb(0, I2, c, c, c, c, 2) == pytest.approx(np.array(KATS["ODE_lqrHjb"]["dV_flat"]))


def test_riccati_ode_quadcopter_demo(lqr):
    """demos/finiteHorizonLqrControl.py:10-31: hover linearisation (n = 8, m = 4), Q = I, R = I, Qf = 10 I, T = 5, N = 50."""
    from zopt_amd import models
    ac = models.Quadcopter()
    xTrim, uTrim = ac.trim(np.zeros(3))
    A, B = (np.asarray(X) for X in ac.linearize(xTrim, uTrim))
    n, m = B.shape
    At, Bt, Qt, Rt = (lambda t: A), (lambda t: B), (lambda t: np.eye(n)), (lambda t: np.eye(m))
    Qf = 10 * np.eye(n)
    K = lqr.finiteHorizonLqr(At, Bt, Qt, Rt, Qf, 5)
    Kr, tr, Vr = zo.finiteHorizonLqr(At, Bt, Qt, Rt, Qf, 5)
    assert K.V.shape == (50, n, n) and np.allclose(K.t, tr)
    assert _rel(K.V, Vr) <= 2e-6
    for tq in (0.0, 0.37, 2.5, 4.99, 5.0):
        assert K(tq).shape == (m, n)
        assert _rel(K(tq), Kr(tq)) <= 2e-6
    assert int(K.info) > 0


def test_riccati_ode_time_varying_batch(lqr):
    """Time-varying coefficients (piecewise linear in t: the kernel's interpolation is exact) and leading batch axes;
    nonsymmetric Qf exercises the reference's formula as written (V A + A^T V, V B R_inv B^T V with a general V)."""
    rng = np.random.default_rng(3)
    batch, n, m, T, N = 5, 6, 2, 2.0, 9
    A0, A1 = 0.5 * rng.standard_normal((2, batch, n, n))
    B0, B1 = rng.standard_normal((2, batch, n, m))
    Qc = np.broadcast_to(np.eye(n), (batch, n, n)).copy()
    Ri0 = np.broadcast_to(np.eye(m), (batch, m, m)).copy()
    At = lambda t: A0 + (t / T) * (A1 - A0)
    Bt = lambda t: B0 + (t / T) * (B1 - B0)
    Qt = lambda t: (1.0 + t) * Qc
    Rt = lambda t: (2.0 - 0.5 * t) * Ri0
    Qf = np.eye(n) + 0.1 * rng.standard_normal((batch, n, n))
    K = lqr.finiteHorizonLqr(At, Bt, Qt, Rt, Qf, T, N=N)
    assert K.V.shape == (batch, N, n, n) and K(0.3).shape == (batch, m, n)
    for b in range(batch):
        pick = lambda f: (lambda t: f(t)[b])
        Kr, _, Vr = zo.finiteHorizonLqr(pick(At), pick(Bt), pick(Qt), pick(Rt), Qf[b], T, N=N)
        assert _rel(K.V[b], Vr) <= 2e-6
        assert _rel(K(0.3)[b], Kr(0.3)) <= 2e-6
    assert np.all(K.info > 0)


def test_riccati_ode_smooth_time_varying_coefficients_are_refined(lqr):
    """Coefficients that are NOT piecewise linear in t (the reference evaluates the callables inside the integrator's right-hand side,
    lqrUtils.py:39-52, 88-97): the host-side sampling is refined until the value function stops changing, and the result agrees with
    the oracle's integration of the exact callables to the integrator's tolerance -- where the unrefined default grid does not."""
    rng = np.random.default_rng(8)
    n, m, T, N = 5, 2, 3.0, 7
    A0, A1 = 0.4 * rng.standard_normal((2, n, n))
    B0 = rng.standard_normal((n, m))
    At = lambda t: A0 * np.cos(2.5 * t) + A1 * np.sin(1.7 * t) ** 2
    Bt = lambda t: B0 * (1.0 + 0.5 * np.sin(3.0 * t))
    Qt = lambda t: np.eye(n) * np.exp(0.4 * t)
    Rt = lambda t: np.eye(m) / (1.0 + 0.3 * t * t)
    Qf = 2.0 * np.eye(n)
    Kr, _, Vr = zo.finiteHorizonLqr(At, Bt, Qt, Rt, Qf, T, N=N)
    K = lqr.finiteHorizonLqr(At, Bt, Qt, Rt, Qf, T, N=N)
    assert K.n_samples > 8 * (N - 1) + 1 and K.coef_change <= 1e-6 * np.max(np.abs(K.V))
    err = _rel(K.V, Vr)
    assert err <= 2e-6 and _rel(K(0.8), Kr(0.8)) <= 5e-5            # (K(t) interpolates V linearly between the N output times, as the reference does)
    K0 = lqr.finiteHorizonLqr(At, Bt, Qt, Rt, Qf, T, N=N, n_samples=8 * (N - 1) + 1)     # refinement off: the round-2 behaviour
    assert _rel(K0.V, Vr) > 10 * err


def test_riccati_ode_edge_cases(lqr):
    import torch
    c = lambda t: I2
    K = lqr.finiteHorizonLqr(c, c, c, c, I2, 1.0, N=1)                # a single output time is T... linspace(0, T, 1) = [0]
    assert K.V.shape == (1, 2, 2)
    K = lqr.finiteHorizonLqr(c, c, c, c, I2, 1.0, N=2)
    assert K.V[1] == pytest.approx(I2)
    with pytest.raises(ValueError):
        lqr.finiteHorizonLqr(c, c, c, c, I2, -1.0)
    with pytest.raises(ValueError):
        lqr.finiteHorizonLqr(c, c, c, c, np.eye(3), 1.0)
    tc = lambda t: torch.eye(2, dtype=torch.float64, device="cuda")
    K = lqr.finiteHorizonLqr(tc, tc, tc, tc, torch.eye(2, dtype=torch.float64, device="cuda"), 1.0, N=4)
    assert K.V.is_cuda and K(0.2).is_cuda
    s2 = np.sqrt(2)
    k0 = ((1 + s2) * np.exp(2 * s2) - (s2 - 1)) / (1 + np.exp(2 * s2))
    assert K(0.0).cpu().numpy() == pytest.approx(k0 * I2, rel=1e-7)


def test_riccati_ode_failures_are_reported(lqr):
    """A step cap that is too small and a Riccati flow with a finite escape time (Q = -100 I: dV/ds = Q - V^2 reaches -inf at
    s = atan(0.1)/10 + pi/20 ~ 0.167) end with info < 0 and NaN at the output times that were not reached; the terminal
    value (t = T) is always delivered."""
    c = lambda t: I2
    K = lqr.finiteHorizonLqr(c, c, c, c, I2, 1.0, N=6, max_steps=3)
    assert int(K.info) == -1
    assert K.V[-1] == pytest.approx(I2) and np.all(np.isnan(K.V[0]))
    zero = lambda t: np.zeros((2, 2))
    K = lqr.finiteHorizonLqr(zero, c, lambda t: -100.0 * I2, c, I2, 1.0, N=6)
    assert int(K.info) < 0
    assert K.V[-1] == pytest.approx(I2) and np.all(np.isnan(K.V[0]))
    # a batch in which only one design fails: the others are unaffected
    Qb = lambda t: np.stack([I2, -100.0 * I2, I2])
    cb = lambda t: np.stack([I2, I2, I2])
    Ab = lambda t: np.stack([I2, 0 * I2, I2])
    K = lqr.finiteHorizonLqr(Ab, cb, Qb, cb, np.stack([I2, I2, I2]), 1.0, N=4)
    assert K.info[0] > 0 and K.info[1] < 0 and K.info[2] > 0
    s2 = np.sqrt(2)
    k0 = ((1 + s2) * np.exp(2 * s2) - (s2 - 1)) / (1 + np.exp(2 * s2))
    assert K.V[0, 0] == pytest.approx(k0 * I2, rel=1e-7) and K.V[2, 0] == pytest.approx(k0 * I2, rel=1e-7)
