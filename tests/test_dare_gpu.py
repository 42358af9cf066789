"""GPU parity tests for discreteInfiniteHorizonLqr (reference lqrUtils.py:176-204).  The reference's implementation is two
SciPy calls, and SciPy runs here: the oracle is that same library call, so parity is pinned by the reference's own
arithmetic (plus its known-answer test, tests/test_lqrUtils.py:72-79)."""
import json
import os

import numpy as np
import pytest

from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


@pytest.fixture(scope="module")
def lqr():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import lqrUtils
    return lqrUtils


def test_kat_golden_ratio(lqr):
    """tests/test_lqrUtils.py:72-79: A = B = Q = R = I2  ->  K = v/(v+1) I, v = (1+sqrt 5)/2."""
    I = np.eye(2)
    K = lqr.discreteInfiniteHorizonLqr(I, I, I, I)
    K_exp = (1 + np.sqrt(5)) / (3 + np.sqrt(5)) * np.eye(2)
    assert K.shape == (2, 2) and np.max(np.abs(K - K_exp)) <= 1e-13


@pytest.mark.parametrize("n,m,rho", [(12, 4, 0.95), (12, 4, 1.3), (8, 4, 0.9), (4, 1, 1.1), (7, 3, 0.99), (2, 2, 2.0), (1, 1, 0.5),
                                     (11, 2, 1.05)])
def test_matches_scipy_dare(lqr, n, m, rho):
    """stable and unstable (rho > 1: stabilisable, B is dense) random systems against solve_discrete_are."""
    batch = 9
    A, B, Q, R = problems.random_lti_systems(batch, n, m, seed=7 + n, rho=rho)
    L, V, its = lqr.discreteInfiniteHorizonLqr(A, B, Q, R, return_value=True)
    assert L.shape == (batch, m, n) and V.shape == (batch, n, n) and its.shape == (batch,)
    for i in range(batch):
        Lr, Vr = zo.discreteInfiniteHorizonLqr(A[i], B[i], Q[i], R[i])
        assert np.max(np.abs(L[i] - Lr)) <= 1e-10 * max(np.max(np.abs(Lr)), 1.0), (i, its[i])
        assert np.max(np.abs(V[i] - Vr)) <= 1e-10 * np.max(np.abs(Vr))
        assert 1 <= its[i] < 200000
    # the finite-horizon sweep approaches the same gain (the cross-check the reference's test suite uses)
    T = 400
    Af, Bf, Qf, Rf = problems.tile_over_horizon(A[:2], B[:2], Q[:2], R[:2], T)
    Lf = lqr.discreteFiniteHorizonLqr(Af, Bf, Qf, Rf, T)
    if rho <= 1.0:
        assert np.max(np.abs(Lf[:, 0] - L[:2])) <= 1e-9 * np.max(np.abs(L[:2]))


def test_torch_batch_axes_and_iteration_cap(lqr):
    import torch
    A, B, Q, R = problems.random_lti_systems(6, 12, 4, seed=3)
    tA, tB, tQ, tR = (torch.as_tensor(x.reshape((2, 3) + x.shape[1:]), device="cuda") for x in (A, B, Q, R))
    L = lqr.discreteInfiniteHorizonLqr(tA, tB, tQ, tR)
    assert isinstance(L, torch.Tensor) and L.is_cuda and L.shape == (2, 3, 4, 12)
    Lr, _ = zo.discreteInfiniteHorizonLqr(A[4], B[4], Q[4], R[4])
    assert np.max(np.abs(L[1, 1].cpu().numpy() - Lr)) <= 1e-10 * np.max(np.abs(Lr))
    Lc, Vc, its = lqr.discreteInfiniteHorizonLqr(A, B, Q, R, maxIter=5, return_value=True)    # capped: 5 Riccati steps
    assert np.all(its == -5)                            # explicit status: the cap ended the loop, after 5 iterations
    Af, Bf, Qf, Rf = problems.tile_over_horizon(A, B, Q, R, 6)
    Lf = lqr.discreteFiniteHorizonLqr(Af, Bf, Qf, Rf, 6)
    assert np.max(np.abs(Lc - Lf[:, 0])) <= 1e-12 * np.max(np.abs(Lc))      # = L_0 of the 6-step horizon from V = Q
    with pytest.raises(np.linalg.LinAlgError):          # unconverged at the cap: SciPy raises LinAlgError, never a silent gain
        lqr.discreteInfiniteHorizonLqr(A, B, Q, R, maxIter=5)
    with pytest.raises(np.linalg.LinAlgError):          # not stabilizable: unstable mode the input cannot reach
        lqr.discreteInfiniteHorizonLqr(np.diag([2.0, 0.5]), np.array([[0.0], [1.0]]), np.eye(2), np.eye(1), maxIter=2000)
    with pytest.raises(ValueError):
        lqr.discreteInfiniteHorizonLqr(np.eye(65), np.ones((65, 2)), np.eye(65), np.eye(2))
    # convergence EXACTLY on the last allowed iteration is convergence (the kernel reports it, the count alone cannot tell):
    # a design that needs k iterations is accepted with maxIter = k and refused with maxIter = k - 4 (the test runs every 4th)
    L1, _, k = lqr.discreteInfiniteHorizonLqr(A[:1], B[:1], Q[:1], R[:1], return_value=True)
    k = int(k[0])
    assert k > 8 and k % 4 == 0
    L2, _, k2 = lqr.discreteInfiniteHorizonLqr(A[:1], B[:1], Q[:1], R[:1], maxIter=k, return_value=True)
    assert int(k2[0]) == k and np.array_equal(L1, L2)
    assert np.array_equal(lqr.discreteInfiniteHorizonLqr(A[:1], B[:1], Q[:1], R[:1], maxIter=k), L1)      # no LinAlgError
    with pytest.raises(np.linalg.LinAlgError):
        lqr.discreteInfiniteHorizonLqr(A[:1], B[:1], Q[:1], R[:1], maxIter=k - 4)


@pytest.mark.parametrize("n,m,rho", [(16, 4, 0.95), (13, 5, 1.1), (24, 8, 0.9), (33, 7, 1.05), (48, 16, 0.95), (64, 16, 0.9), (57, 3, 1.02), (12, 6, 1.2)])
def test_matches_scipy_dare_large_states(lqr, n, m, rho):
    """`discreteInfiniteHorizonLqr` beyond the tile-16 shapes (12 < n <= 64 or 4 < m <= 16): the fp64 tile kernel's Joseph-form step
    iterated on time-invariant operands until the gain stops changing; stable and unstable random systems against SciPy's
    solve_discrete_are -- the library the reference itself calls (lqrUtils.py:202-203)."""
    batch = 3
    A, B, Q, R = problems.random_lti_systems(batch, n, m, seed=70 + n, rho=rho)
    L, V, its = lqr.discreteInfiniteHorizonLqr(A, B, Q, R, return_value=True)
    assert L.shape == (batch, m, n) and V.shape == (batch, n, n) and its.shape == (batch,)
    for i in range(batch):
        Lr, Vr = zo.discreteInfiniteHorizonLqr(A[i], B[i], Q[i], R[i])
        assert np.max(np.abs(L[i] - Lr)) <= 1e-10 * max(np.max(np.abs(Lr)), 1.0), (i, its[i])
        assert np.max(np.abs(V[i] - Vr)) <= 1e-9 * np.max(np.abs(Vr)), (i, its[i])
        assert 1 <= its[i] < 200000
    assert np.array_equal(lqr.discreteInfiniteHorizonLqr(A, B, Q, R), L)
    with pytest.raises(np.linalg.LinAlgError):          # unconverged at the cap
        lqr.discreteInfiniteHorizonLqr(A, B, Q, R, maxIter=3)
    Lc, _, itc = lqr.discreteInfiniteHorizonLqr(A, B, Q, R, maxIter=3, return_value=True)
    assert np.all(itc == -3)
    Af, Bf, Qf, Rf = problems.tile_over_horizon(A, B, Q, R, 4)
    Lf = lqr.discreteFiniteHorizonLqr(Af, Bf, Qf, Rf, 4)
    assert np.max(np.abs(Lc - Lf[:, 1])) <= 1e-11 * np.max(np.abs(Lc))     # 3 iterations from V = Q = the gain at step T-3 of a 4-step horizon
