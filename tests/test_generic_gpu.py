"""The generic-callable path (zopt_amd/generic.py): the reference's drivers take arbitrary callables `dynamics(x, u)`, `runningCost(x, u)`,
`terminalCost(x)` (ilqrUtils.py:260-269, 330-338) and differentiate them by JAX autodiff; here torch callables are differentiated and
rolled out with torch.func ON THE GPU and the sweeps / PD projections are the HIP kernels.  Checked against the CPU oracle loops and
against the registered-model (fused) path on the same problems."""
import json
import os

import numpy as np
import pytest

from oracle import zopt_oracle as zo

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import ilqrUtils, models, pytrees
    return ilqrUtils, models, pytrees


def _quad_torch(dt):
    """the quadcopter as a torch callable of single points: the oracle's torch restatement (quadcopter.py:23-144)"""
    return zo.quad_euler_step_torch(dt)


def test_kat_ilqr_and_ddp_with_callables(mods):
    """reference tests/test_ilqrUtils.py:167-196: A = B = Q = R = I, N = 3, x0 = (2, 1), given as plain callables"""
    import torch
    ilqr, _, _ = mods
    I = np.eye(2)
    x0 = np.array([2.0, 1.0])
    for solver, ref in ((ilqr.iterativeLqr, zo.iterativeLqr), (ilqr.differentialDynamicProgramming, None)):
        traj, L, J, conv = solver(lambda x, u: x + u, lambda x, u: x @ x + u @ u, lambda x: x @ x, x0, np.zeros((3, 2)))
        assert conv is True and isinstance(J, float)
        if ref is not None:
            rt, rL, rJ, rc = ref(lambda x, u: x + u, I, I, I, x0, np.zeros((3, 2)))
        else:
            rt, rL, rJ, rc = zo.differentialDynamicProgramming(lambda x, u: x + u, lambda x, u: x + u, I, I, I, x0, np.zeros((3, 2)))
        assert rc and _rel(traj.uTraj, rt.uTraj) <= 1e-9 and _rel(L, rL) <= 1e-9 and J == pytest.approx(rJ, rel=1e-10)


@pytest.mark.parametrize("ddp", [False, True])
def test_quadcopter_as_a_callable_matches_the_registered_model_and_the_oracle(mods, ddp):
    """demos/iterativeLqr.py:22-39 / differentialDynamicProgramming.py:22-39 (N = 25 here): the same problems through the generic path
    (torch callables), the fused path (registered model) and the oracle loop"""
    ilqr, models, _ = mods
    N = 25
    Q, R, Qf = np.eye(12), (0.2 if ddp else 1.0) * np.eye(4), 10 * np.eye(12)
    cost = models.QuadraticCost(Q, R, Qf)
    x0 = np.zeros((3, 12))
    x0[:, 9:12] = [[0, 5, 0], [1, -2, 3], [-4, 1, 2]]
    ug = np.tile(models.QuadcopterEuler.uTrim, (3, N, 1))
    solve = ilqr.differentialDynamicProgramming if ddp else ilqr.iterativeLqr
    ft = _quad_torch(0.1)
    import torch
    tQ, tR, tQf = (torch.as_tensor(M, device="cuda") for M in (Q, R, Qf))
    gen = solve(ft, lambda x, u: x @ tQ @ x + u @ tR @ u, lambda x: x @ tQf @ x, x0, ug)
    mixed = solve(ft, cost.runningCost, cost.terminalCost, x0, ug)          # callable dynamics, registered quadratic cost
    fused = solve(models.QuadcopterEuler(0.1), cost, cost, x0, ug)
    fn = zo.quad_euler_step(0.1)
    for i in range(3):
        if ddp:
            rt, rL, rJ, rc = zo.differentialDynamicProgramming(fn, ft, Q, R, Qf, x0[i], ug[i])
        else:
            rt, rL, rJ, rc = zo.iterativeLqr(fn, Q, R, Qf, x0[i], ug[i])
        for (traj, L, J, conv) in (gen, mixed, fused):
            assert bool(conv[i]) == rc and abs(J[i] - rJ) <= 1e-7 * abs(rJ)
            assert _rel(traj.xTraj[i], rt.xTraj) <= 1e-6 and _rel(traj.uTraj[i], rt.uTraj) <= 1e-6 and _rel(L[i], rL) <= 1e-5


def test_expansions_of_callables_match_the_registered_model(mods):
    """AffineDynamics / QuadraticDynamics / QuadraticCostFunction.from_trajectory and fromTerminalCostFunction for callables
    (pytrees.py:72-81, 100-115, 139-153, 180-194) against the device model's kernels"""
    ilqr, models, pt = mods
    rng = np.random.default_rng(3)
    N = 6
    xT = 0.3 * rng.standard_normal((2, N + 1, 12))
    uT = models.QuadcopterEuler.uTrim + 0.3 * rng.standard_normal((2, N, 4))
    traj = pt.Trajectory(xT, uT)
    ft, md = _quad_torch(0.1), models.QuadcopterEuler(0.1)
    a, b = pt.QuadraticDynamics.from_trajectory(ft, traj), pt.QuadraticDynamics.from_trajectory(md, traj)
    for fa, fb in zip(tuple.__iter__(a), tuple.__iter__(b)):
        assert fa.shape == fb.shape and np.max(np.abs(fa - fb)) <= 1e-12
    Q, R, Qf = rng.standard_normal((12, 12)), rng.standard_normal((4, 4)), rng.standard_normal((12, 12))
    cost = models.QuadraticCost(Q, R, Qf)
    import torch
    tQ, tR, tQf = (torch.as_tensor(M, device="cuda") for M in (Q, R, Qf))

    class Costs:                                  # CostFunction(runningCost, terminalCost) of the reference, torch callables
        runningCost = staticmethod(lambda x, u: x @ tQ @ x + u @ tR @ u)
        terminalCost = staticmethod(lambda x: x @ tQf @ x)

    qa, qb = pt.QuadraticCostFunction.from_trajectory(Costs, traj), pt.QuadraticCostFunction.from_trajectory(cost, traj)
    for fa, fb in zip(tuple.__iter__(qa), tuple.__iter__(qb)):
        assert fa.shape == fb.shape and np.max(np.abs(fa - fb)) <= 1e-11
    va, vb = pt.QuadraticValueFunction.fromTerminalCostFunction(Costs, xT[0, -1]), pt.QuadraticValueFunction.fromTerminalCostFunction(cost, xT[0, -1])
    for fa, fb in zip(tuple.__iter__(va), tuple.__iter__(vb)):
        assert np.shape(fa) == np.shape(fb) and np.max(np.abs(np.asarray(fa) - np.asarray(fb))) <= 1e-11


def test_forwardPass2_with_callables(mods):
    """16-way line search of a callable model with callable costs against the oracle (ilqrUtils.py:116-150), incl. NaN-wins argmin"""
    ilqr, models, pt = mods
    import torch
    rng = np.random.default_rng(8)
    batch, N, n, m = 5, 12, 3, 2
    A, Bm = rng.standard_normal((n, n)) * 0.5, rng.standard_normal((n, m))
    tA, tB = torch.as_tensor(A, device="cuda"), torch.as_tensor(Bm, device="cuda")
    f_t = lambda x, u: tA @ torch.sin(x) + tB @ u
    f_np = lambda x, u: A @ np.sin(x) + Bm @ u
    x0, l, L = rng.standard_normal((batch, n)), 2.0 * rng.standard_normal((batch, N, m)), 0.2 * rng.standard_normal((batch, N, m, n))
    xPrev, uPrev = 0.3 * rng.standard_normal((batch, N + 1, n)), 0.3 * rng.standard_normal((batch, N, m))

    class Costs:
        runningCost = staticmethod(lambda x, u: (x * x).sum() + 0.1 * (u ** 4).sum())
        terminalCost = staticmethod(lambda x: 3.0 * (x * x).sum())

    traj, J = ilqr.forwardPass2(x0, f_t, Costs, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev))
    rcost, tcost = (lambda x, u: (x * x).sum() + 0.1 * (u ** 4).sum()), (lambda x: 3.0 * (x * x).sum())
    for b in range(batch):
        rt, rJ = zo.forwardPass2(x0[b], f_np, rcost, tcost, zo.AffinePolicy(l[b], L[b]), zo.Trajectory(xPrev[b], uPrev[b]))
        assert abs(J[b] - rJ) <= 1e-10 * abs(rJ) and _rel(traj.xTraj[b], rt.xTraj) <= 1e-10 and _rel(traj.uTraj[b], rt.uTraj) <= 1e-10
    from zopt_amd import generic
    Jn = torch.tensor([[3.0, float("nan"), -float("inf"), float("nan")], [2.0, 1.0, 1.0, 5.0]])
    assert generic.argmin_nan_wins(Jn).tolist() == [1, 1]
