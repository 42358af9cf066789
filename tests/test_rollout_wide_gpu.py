"""GPU parity tests of the large-linear-model rollouts (rollout_wide.hip): `trajectoryRollout` / `forwardPass2` (reference
ilqrUtils.py:33-66, 116-150; pytrees.py:49-52, 215-220) for LinearModel beyond the lane-per-rollout kernels' n <= 12, m <= 4 -- one
wave per rollout, n <= 64, m <= 16 -- against the oracle, including the NaN-wins argmin and a line search whose winners differ."""
import numpy as np
import pytest

from oracle import zopt_oracle as zo

pytestmark = pytest.mark.gpu
SHAPES = [(16, 4, 12, 5), (24, 8, 9, 3), (48, 16, 6, 2), (64, 16, 5, 3), (13, 5, 8, 4), (33, 1, 5, 2), (12, 5, 6, 2), (57, 9, 7, 2), (3, 7, 4, 2)]


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import ilqrUtils, models, pytrees
    return ilqrUtils, models, pytrees


def _problem(rng, batch, N, n, m):
    l = 0.3 * rng.standard_normal((batch, N, m))
    L = 0.2 * rng.standard_normal((batch, N, m, n)) / np.sqrt(n)
    xPrev = rng.standard_normal((batch, N + 1, n))
    uPrev = 0.3 * rng.standard_normal((batch, N, m))
    x0 = rng.standard_normal((batch, n))
    return x0, l, L, xPrev, uPrev


@pytest.mark.parametrize("n,m,N,batch", SHAPES)
def test_trajectoryRollout_large_linear_models(mods, n, m, N, batch):
    ilqr, models, pt = mods
    rng = np.random.default_rng(100 * n + m)
    model = models.LinearModel(rng.standard_normal((n, n)) * (0.9 / np.sqrt(n)), rng.standard_normal((n, m)))
    x0, l, L, xPrev, uPrev = _problem(rng, batch, N, n, m)
    for alpha in (1, 0.25):
        t = ilqr.trajectoryRollout(x0, model, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev), alpha=alpha)
        assert t.xTraj.shape == (batch, N + 1, n) and t.uTraj.shape == (batch, N, m)
        assert np.array_equal(t.xTraj[:, 0], x0)
        for b in range(batch):
            r = zo.trajectoryRollout(x0[b], model, zo.AffinePolicy(l[b], L[b]), zo.Trajectory(xPrev[b], uPrev[b]), alpha=alpha)
            assert _rel(t.xTraj[b], r.xTraj) <= 1e-11 and _rel(t.uTraj[b], r.uTraj) <= 1e-11


@pytest.mark.parametrize("n,m,N,batch", SHAPES)
def test_forwardPass2_large_linear_models(mods, n, m, N, batch):
    ilqr, models, pt = mods
    rng = np.random.default_rng(11 + n)
    model = models.LinearModel(rng.standard_normal((n, n)) * (1.05 / np.sqrt(n)), rng.standard_normal((n, m)))
    x0, l, L, xPrev, uPrev = _problem(rng, batch, N, n, m)
    l *= 4.0 * rng.uniform(0.05, 3.0, (batch, 1, 1))      # step lengths from timid to far too big: different alphas win
    Mq, Mr = rng.standard_normal((n, n)), rng.standard_normal((m, m))
    cost = models.QuadraticCost(Mq @ Mq.T / n + np.eye(n), Mr @ Mr.T / m + np.eye(m) + 0.1 * rng.standard_normal((m, m)), 10 * np.eye(n))
    traj, J = ilqr.forwardPass2(x0, model, cost, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev))
    assert J.shape == (batch,)
    for b in range(batch):
        rt, rJ = zo.forwardPass2(x0[b], model, cost.runningCost, cost.terminalCost, zo.AffinePolicy(l[b], L[b]),
                                 zo.Trajectory(xPrev[b], uPrev[b]))
        assert abs(J[b] - rJ) <= 1e-10 * abs(rJ)
        assert _rel(traj.xTraj[b], rt.xTraj) <= 1e-9 and _rel(traj.uTraj[b], rt.uTraj) <= 1e-9


def test_forwardPass2_large_picks_different_step_sizes_and_nan_wins(mods):
    ilqr, models, pt = mods
    rng = np.random.default_rng(5)
    n, m, N, batch = 20, 6, 15, 12
    model = models.LinearModel(rng.standard_normal((n, n)) * (1.05 / np.sqrt(n)), rng.standard_normal((n, m)))
    x0, l, L, xPrev, uPrev = _problem(rng, batch, N, n, m)
    l *= np.logspace(-1, 1.5, batch)[:, None, None]
    cost = models.QuadraticCost(np.eye(n), np.eye(m), 10 * np.eye(n))
    traj, J = ilqr.forwardPass2(x0, model, cost, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev))
    picked = set()
    for b in range(batch):
        Js = [zo.trajectoryCost(cost.runningCost, cost.terminalCost,
                                zo.trajectoryRollout(x0[b], model, zo.AffinePolicy(l[b], L[b]), zo.Trajectory(xPrev[b], uPrev[b]), alpha=a))
              for a in zo.LINESEARCH_ALPHAS]
        picked.add(int(np.argmin(Js)))
        assert abs(J[b] - min(Js)) <= 1e-10 * abs(min(Js))
    assert len(picked) > 2
    # NaN wins the argmin (ilqrUtils.py:147): alpha = 1 overflows to inf - inf = NaN, smaller steps give finite or -inf costs
    n = m = 14
    A = np.eye(n)
    A[0, 1] = -1.0
    model = models.LinearModel(A, np.eye(n))
    Qf = np.zeros((n, n))
    Qf[0, 0], Qf[1, 1] = 1.0, -1.0
    cost = models.QuadraticCost(np.zeros((n, n)), np.zeros((m, m)), Qf)
    N = 3
    l = np.zeros((1, N, m))
    l[..., :2] = 1.5e308
    prev = pt.Trajectory(np.zeros((1, N + 1, n)), np.zeros((1, N, m)))
    with np.errstate(all="ignore"):
        traj, J = ilqr.forwardPass2(np.zeros((1, n)), model, cost, pt.AffinePolicy(l, np.zeros((1, N, m, n))), prev)
        rt, rJ = zo.forwardPass2(np.zeros(n), model, cost.runningCost, cost.terminalCost, zo.AffinePolicy(l[0], np.zeros((N, m, n))),
                                 zo.Trajectory(prev.xTraj[0], prev.uTraj[0]))
    assert np.isnan(rJ) and np.isnan(J[0])
    assert np.array_equal(traj.uTraj[0], rt.uTraj, equal_nan=True)


def test_shapes_beyond_the_wide_kernel_are_refused(mods):
    ilqr, models, pt = mods
    n, m, N = 65, 2, 3
    model = models.LinearModel(np.eye(n), np.ones((n, m)))
    with pytest.raises(ValueError):
        ilqr.trajectoryRollout(np.zeros(n), model, pt.AffinePolicy(np.zeros((N, m)), np.zeros((N, m, n))),
                               pt.Trajectory(np.zeros((N + 1, n)), np.zeros((N, m))))
