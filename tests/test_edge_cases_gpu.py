"""Edge cases the reference's array semantics imply: empty batches, single-step horizons, batch axes of
several dimensions, non-contiguous and over-long inputs (`N` shorter than the arrays), singular solves propagating
inf/NaN instead of raising."""
import numpy as np
import pytest

from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import ilqrUtils, lqrUtils, models, mpcUtils, pytrees
    return lqrUtils, ilqrUtils, mpcUtils, models, pytrees


def test_empty_batches(mods):
    lqr, ilqr, mpc, models, pt = mods
    for dt, n, m in ((np.float64, 12, 4), (np.float32, 64, 16), (np.float64, 20, 5)):
        z = lambda *s: np.zeros((0,) + s, dtype=dt)
        L = lqr.discreteFiniteHorizonLqr(z(3, n, n), z(3, n, m), z(3, n, n), z(3, m, m), 3)
        assert L.shape == (0, 3, m, n) and L.dtype == dt
    z = lambda *s: np.zeros((0,) + s)
    L, l = lqr.bilinearAffineLqr(z(4, 8, 8), z(4, 8, 4), z(4, 8), z(4, 8, 8), z(4, 4, 4), z(4, 4, 8), z(4, 8), z(4, 4), z(4), 4)
    assert L.shape == (0, 4, 4, 8) and l.shape == (0, 4, 4)
    K = lqr.discreteInfiniteHorizonLqr(z(6, 6), z(6, 2), z(6, 6), z(2, 2))
    assert K.shape == (0, 2, 6)
    pol = ilqr.backwardPass_ilqr(pt.AffineDynamics(z(5, 3), z(5, 3, 3), z(5, 3, 2)),
                                 pt.QuadraticCostFunction(z(5), z(5, 3), z(5, 2), z(5, 3, 3), z(5, 2, 3), z(5, 2, 2)),
                                 pt.QuadraticValueFunction(z(), z(3), z(3, 3)))
    assert pol.l.shape == (0, 5, 2) and pol.L.shape == (0, 5, 2, 3)
    model = models.QuadcopterEuler(0.1)
    cost = models.QuadraticCost(np.eye(12), np.eye(4))
    traj, Lg, J, conv = ilqr.iterativeLqr(model, cost, cost, z(12), z(7, 4))
    assert traj.xTraj.shape == (0, 8, 12) and traj.uTraj.shape == (0, 7, 4) and Lg.shape == (0, 7, 4, 12)
    assert J.shape == (0,) and conv.shape == (0,)
    prob = mpc.lqrMpc(np.eye(2), np.eye(2), np.eye(2), np.eye(2), 3, -np.ones(2), np.ones(2), -np.ones(2), np.ones(2))
    u, tr, st = prob.solve(z(2))
    assert u.shape == (0, 2) and tr.xTraj.shape == (0, 4, 2) and tr.uTraj.shape == (0, 3, 2) and len(st) == 0
    assert ilqr.ensurePositiveDefinite(z(5, 5)).shape == (0, 5, 5)


def test_multi_axis_batches_and_over_long_inputs(mods):
    lqr, ilqr, mpc, models, pt = mods
    A, B, Q, R = problems.random_time_varying(6, 9, 12, 4, seed=31)
    sh = lambda X: X.reshape((2, 3) + X.shape[1:])
    L = lqr.discreteFiniteHorizonLqr(sh(A), sh(B), sh(Q), sh(R), 9)
    assert L.shape == (2, 3, 9, 4, 12)
    Lr = zo.discreteFiniteHorizonLqr(A, B, Q, R, 9)
    assert np.max(np.abs(L.reshape(6, 9, 4, 12) - Lr)) <= 1e-10 * np.max(np.abs(Lr))
    # N shorter than the arrays: the reference scans xs = arange(N), terminal value Q[-1] of the FULL array (lqrUtils.py:172)
    L5 = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 5)
    Lr5 = zo.discreteFiniteHorizonLqr(A, B, Q, R, 5)
    assert L5.shape == (6, 5, 4, 12)
    assert np.max(np.abs(L5 - Lr5)) <= 1e-10 * np.max(np.abs(Lr5))
    # non-contiguous views (Fortran-ordered / strided inputs)
    Af = np.asfortranarray(A)
    Ls = lqr.discreteFiniteHorizonLqr(Af, B[:, ::1], Q, R, 9)
    assert np.array_equal(Ls, L.reshape(6, 9, 4, 12))


def test_singular_solve_propagates_nonfinite(mods):
    """`jnp.linalg.solve` on a singular Suu returns inf/NaN silently (JAX semantics): no exception here either,
    and the other trajectories of the batch are unaffected."""
    lqr, *_ = mods
    for dt, n, m in ((np.float64, 12, 4), (np.float32, 32, 16), (np.float64, 20, 6)):
        A, B, Q, R = problems.random_time_varying(3, 4, n, m, seed=8, dtype=dt)
        B[1] = 0
        R[1] = 0                                   # Suu = R + B^T V B = 0 for trajectory 1
        L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 4)
        assert not np.all(np.isfinite(L[1]))
        Lr = zo.discreteFiniteHorizonLqr(*(x[[0, 2]].astype(np.float64) for x in (A, B, Q, R)), 4)
        tol = 1e-10 if dt == np.float64 else 2e-4
        assert np.max(np.abs(L[[0, 2]] - Lr)) <= tol * np.max(np.abs(Lr))


def test_single_step_horizons(mods):
    lqr, ilqr, mpc, models, pt = mods
    A, B, Q, R = problems.random_time_varying(2, 1, 12, 4, seed=4)
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 1)
    assert np.max(np.abs(L - zo.discreteFiniteHorizonLqr(A, B, Q, R, 1))) <= 1e-12
    model = models.LinearModel(np.eye(2), np.eye(2))
    cost = models.QuadraticCost(np.eye(2), np.eye(2))
    traj, Lg, J, conv = ilqr.iterativeLqr(model, cost, cost, np.array([2.0, 1.0]), np.zeros((1, 2)))
    assert traj.xTraj.shape == (2, 2) and Lg.shape == (1, 2, 2) and isinstance(conv, bool)
