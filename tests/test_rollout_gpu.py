"""GPU parity tests for K6 rollout_linesearch (trajectoryRollout / forwardPass2, reference ilqrUtils.py:33-66, 116-150).

Tolerance: trajectories are T-step recursions through the model; device sin/cos/tan differ from libm by <= a few ulp,
so max|err| <= 1e-9 * max|ref| (measured ~1e-13) and J to 1e-10 relative."""
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


def test_kat_trajectoryRollout(mods):
    """reference tests/test_ilqrUtils.py:7-22: dynFun = x+u, policy = alpha*k, zero previous trajectory; exact."""
    ilqr, models, pt = mods
    k = KATS["A6_trajectoryRollout"]
    N = k["N"]
    model = models.LinearModel(np.eye(1), np.eye(1))
    policy = pt.AffinePolicy(np.arange(N, dtype=np.float64)[:, None], np.zeros((N, 1, 1)))
    trajPrev = pt.Trajectory(np.zeros((N + 1, 1)), np.zeros((N, 1)))
    x0 = np.array(k["x0"])
    t = ilqr.trajectoryRollout(x0, model, policy, trajPrev)
    assert np.all(t.xTraj == np.array(k["alpha1"]["xTraj"])[:, None])
    assert np.all(t.uTraj == np.array(k["alpha1"]["uTraj"])[:, None])
    t = ilqr.trajectoryRollout(x0, model, policy, trajPrev, alpha=0.5)
    assert np.all(t.xTraj == np.array(k["alpha0.5"]["xTraj"])[:, None])
    assert np.all(t.uTraj == np.array(k["alpha0.5"]["uTraj"])[:, None])


def _random_policy_problem(rng, batch, N, n, m, x_scale=1.0):
    l = 0.3 * rng.standard_normal((batch, N, m))
    L = 0.2 * rng.standard_normal((batch, N, m, n)) / np.sqrt(n)
    xPrev = x_scale * rng.standard_normal((batch, N + 1, n))
    uPrev = 0.3 * rng.standard_normal((batch, N, m))
    x0 = x_scale * rng.standard_normal((batch, n))
    return x0, l, L, xPrev, uPrev


@pytest.mark.parametrize("n,m,N,batch", [(12, 4, 30, 9), (2, 2, 3, 5), (4, 1, 50, 3), (8, 4, 20, 70), (5, 3, 7, 2), (1, 1, 4, 1)])
def test_linear_model_rollout_parity(mods, n, m, N, batch):
    ilqr, models, pt = mods
    rng = np.random.default_rng(100 * n + m)
    A = rng.standard_normal((n, n)) * (0.9 / np.sqrt(n))
    B = rng.standard_normal((n, m))
    model = models.LinearModel(A, B)
    x0, l, L, xPrev, uPrev = _random_policy_problem(rng, batch, N, n, m)
    for alpha in (1, 0.25):
        t = ilqr.trajectoryRollout(x0, model, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev), alpha=alpha)
        assert t.xTraj.shape == (batch, N + 1, n) and t.uTraj.shape == (batch, N, m)
        for b in range(batch):
            r = zo.trajectoryRollout(x0[b], model, zo.AffinePolicy(l[b], L[b]), zo.Trajectory(xPrev[b], uPrev[b]), alpha=alpha)
            assert _rel(t.xTraj[b], r.xTraj) <= 1e-11 and _rel(t.uTraj[b], r.uTraj) <= 1e-11


def _quad_problem(rng, batch, N):
    """Small perturbations around hover: the open-loop quadcopter is unstable (and its quadratic drag blows up in finite
    time for large negative speeds), so keep the excursion moderate over the horizon."""
    x0, l, L, xPrev, uPrev = _random_policy_problem(rng, batch, N, 12, 4, x_scale=0.05)
    l, L = 0.05 * l, 0.1 * L
    uPrev = 0.1 * uPrev + np.array([9.807, 0, 0, 0])   # around hover
    return x0, l, L, xPrev, uPrev


def test_quadcopter_rollout_parity(mods):
    """Device restatement of quadcopter.py:116-144 (incl. quirk Q4) + Euler step vs the NumPy oracle, T = 60 (excursions up to |x| ~ 25)."""
    ilqr, models, pt = mods
    rng = np.random.default_rng(7)
    batch, N = 6, 60
    x0, l, L, xPrev, uPrev = _quad_problem(rng, batch, N)
    model = models.QuadcopterEuler(dt=0.1)
    step = zo.quad_euler_step(0.1)
    t = ilqr.trajectoryRollout(x0, model, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev), alpha=0.5)
    for b in range(batch):
        r = zo.trajectoryRollout(x0[b], step, zo.AffinePolicy(l[b], L[b]), zo.Trajectory(xPrev[b], uPrev[b]), alpha=0.5)
        assert np.all(np.isfinite(r.xTraj))
        assert _rel(t.xTraj[b], r.xTraj) <= 1e-9 and _rel(t.uTraj[b], r.uTraj) <= 1e-9


def test_quadcopter_model_known_answers_on_device(mods):
    """reference tests/test_quadcopter.py:62-86 through a 1-step rollout: x1 = x0 + dt*f(x0,u0)."""
    ilqr, models, pt = mods
    dt = 0.1
    hover = np.array([9.807, 0, 0, 0.0])
    states = np.zeros((3, 12))
    states[1, 0:3] = [0.1, 0.2, 0.3]
    states[2, 0:3] = [0.1, 0.2, 0.3]
    states[2, 8] = np.pi / 2
    pol = pt.AffinePolicy(np.zeros((3, 1, 4)), np.zeros((3, 1, 4, 12)))
    prev = pt.Trajectory(np.zeros((3, 2, 12)), np.tile(hover, (3, 1, 1)))
    t = ilqr.trajectoryRollout(states, models.QuadcopterEuler(dt), pol, prev)
    xdot = (t.xTraj[:, 1] - states) / dt
    assert xdot[0] == pytest.approx(np.zeros(12), abs=1e-12)                    # hover: no motion
    assert xdot[1, 9:] == pytest.approx(np.array([0.1, 0.2, 0.3]), rel=1e-12)   # no rotation
    assert xdot[2, 9:] == pytest.approx(np.array([-0.2, 0.1, 0.3]), rel=1e-12)  # psi = 90 deg


@pytest.mark.parametrize("kind", ["linear", "quadcopter"])
def test_forwardPass2_parity(mods, kind):
    ilqr, models, pt = mods
    rng = np.random.default_rng(11)
    batch, N = 10, 25
    if kind == "linear":
        n, m = 12, 4
        model = models.LinearModel(rng.standard_normal((n, n)) * (1.05 / np.sqrt(n)), rng.standard_normal((n, m)))
        x0, l, L, xPrev, uPrev = _random_policy_problem(rng, batch, N, n, m)
        l *= 4.0      # big steps: smaller alphas must win for some trajectories
        f = model
    else:
        n, m = 12, 4
        model = models.QuadcopterEuler(0.1)
        x0, l, L, xPrev, uPrev = _quad_problem(rng, batch, N)
        l *= 60.0     # big feed-forward steps: smaller alphas must win for some trajectories
        f = zo.quad_euler_step(0.1)
    Mq = rng.standard_normal((n, n)); Mr = rng.standard_normal((m, m))
    cost = models.QuadraticCost(Mq @ Mq.T / n + np.eye(n), Mr @ Mr.T / m + np.eye(m), 10 * np.eye(n))
    traj, J = ilqr.forwardPass2(x0, model, cost, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev))
    assert J.shape == (batch,)
    picked = set()
    for b in range(batch):
        rt, rJ = zo.forwardPass2(x0[b], f, cost.runningCost, cost.terminalCost, zo.AffinePolicy(l[b], L[b]),
                                 zo.Trajectory(xPrev[b], uPrev[b]))
        assert abs(J[b] - rJ) <= 1e-10 * abs(rJ)
        assert _rel(traj.xTraj[b], rt.xTraj) <= 1e-9 and _rel(traj.uTraj[b], rt.uTraj) <= 1e-9
        picked.add(round(float(np.max(np.abs(rt.uTraj - uPrev[b]))), 6))
    assert len(picked) > 1


def test_nan_wins_argmin(mods):
    """jnp.argmin / np.argmin treat NaN as the minimum: the first NaN cost wins the line search (ilqrUtils.py:147),
    even against -inf.  alpha = 1 drives the state to (-inf, +inf) -> x'Qx = inf - inf = NaN; smaller steps give -inf."""
    ilqr, models, pt = mods
    n = m = 2
    model = models.LinearModel(np.array([[1.0, -1.0], [0.0, 1.0]]), np.eye(2))
    cost = models.QuadraticCost(np.zeros((2, 2)), np.zeros((2, 2)), np.diag([1.0, -1.0]))
    N = 3
    l = np.full((1, N, m), 1.5e308)
    L = np.zeros((1, N, m, n))
    prev = pt.Trajectory(np.zeros((1, N + 1, n)), np.zeros((1, N, m)))
    x0 = np.zeros((1, n))
    traj, J = ilqr.forwardPass2(x0, model, cost, pt.AffinePolicy(l, L), prev)
    with np.errstate(all="ignore"):
        Js = [zo.trajectoryCost(cost.runningCost, cost.terminalCost,
                                zo.trajectoryRollout(x0[0], model, zo.AffinePolicy(l[0], L[0]),
                                                     zo.Trajectory(prev.xTraj[0], prev.uTraj[0]), alpha=a))
              for a in zo.LINESEARCH_ALPHAS]
        rt, rJ = zo.forwardPass2(x0[0], model, cost.runningCost, cost.terminalCost, zo.AffinePolicy(l[0], L[0]),
                                 zo.Trajectory(prev.xTraj[0], prev.uTraj[0]))
    assert np.isnan(Js[0]) and (-np.inf in Js)           # the construction really has NaN competing with -inf
    assert np.isnan(rJ) and np.isnan(J[0])
    assert np.array_equal(traj.uTraj[0], rt.uTraj, equal_nan=True)       # alpha = 1 (index 0)


def test_callables_take_the_generic_path_and_non_callables_are_rejected(mods):
    """a torch-traceable callable is rolled out by zopt_amd/generic.py (the reference's KAT, tests/test_ilqrUtils.py:7-22: x+ = x + u,
    policy alpha * k); anything that is neither a registered model nor callable is refused"""
    ilqr, models, pt = mods
    k = KATS["A6_trajectoryRollout"]
    N = k["N"]
    policy = pt.AffinePolicy(np.arange(N, dtype=np.float64)[:, None], np.zeros((N, 1, 1)))
    prev = pt.Trajectory(np.zeros((N + 1, 1)), np.zeros((N, 1)))
    t = ilqr.trajectoryRollout(np.array(k["x0"]), lambda x, u: x + u, policy, prev)
    assert np.all(t.xTraj == np.array(k["alpha1"]["xTraj"])[:, None]) and np.all(t.uTraj == np.array(k["alpha1"]["uTraj"])[:, None])
    with pytest.raises(TypeError):
        ilqr.trajectoryRollout(np.zeros(1), "no model", pt.AffinePolicy(np.zeros((1, 1)), np.zeros((1, 1, 1))),
                               pt.Trajectory(np.zeros((2, 1)), np.zeros((1, 1))))


def test_cost_function_call_matches_reference_definition():
    """CostFunction.__call__ (pytrees.py:40-55): J = terminalCost(x_N) + sum runningCost(x_k, u_k); index k -> one running cost."""
    from zopt_amd import models, pytrees
    rng = np.random.default_rng(21)
    Q, R, Qf = rng.standard_normal((12, 12)), rng.standard_normal((4, 4)), rng.standard_normal((12, 12))
    cost = models.QuadraticCost(Q, R, Qf)
    xT, uT = rng.standard_normal((3, 8, 12)), rng.standard_normal((3, 7, 4))
    J = cost(pytrees.Trajectory(xT, uT))
    ref = np.array([sum(xT[b, k] @ Q @ xT[b, k] + uT[b, k] @ R @ uT[b, k] for k in range(7)) + xT[b, -1] @ Qf @ xT[b, -1]
                    for b in range(3)])
    assert J.shape == (3,) and np.max(np.abs(J - ref)) <= 1e-11 * np.max(np.abs(ref))
    j2 = cost(pytrees.Trajectory(xT[1], uT[1]), k=2)
    assert abs(j2 - (xT[1, 2] @ Q @ xT[1, 2] + uT[1, 2] @ R @ uT[1, 2])) <= 1e-12
    assert isinstance(cost(pytrees.Trajectory(xT[0], uT[0])), float)


@pytest.mark.parametrize("kind", ["linear", "quadcopter"])
def test_diagonal_weight_kernels_agree(mods, kind):
    """Diagonal Q, R, Qf (the demos' weights) run a leaner rollout kernel when the cost handle says so
    (`zm_quadcost_t.diagonal`, set by zopt_amd.models.QuadraticCost from the matrices themselves); with the hint cleared the
    general kernel finds the diagonal structure itself.  Both against the oracle and bit for bit against each other."""
    ilqr, models, pt = mods
    rng = np.random.default_rng(5)
    batch, N, n, m = 9, 30, 12, 4
    if kind == "linear":
        model = models.LinearModel(rng.standard_normal((n, n)) * (1.05 / np.sqrt(n)), rng.standard_normal((n, m)))
        x0, l, L, xPrev, uPrev = _random_policy_problem(rng, batch, N, n, m)
        l *= 4.0
        f = model
    else:
        model = models.QuadcopterEuler(0.1)
        x0, l, L, xPrev, uPrev = _quad_problem(rng, batch, N)
        l *= 60.0
        f = zo.quad_euler_step(0.1)
    Q, R, Qf = np.diag(rng.uniform(0.5, 2.0, n)), np.diag(rng.uniform(0.5, 2.0, m)), np.diag(rng.uniform(5.0, 20.0, n))
    cost = models.QuadraticCost(Q, R, Qf)
    assert cost.c_struct().diagonal == 1
    assert models.QuadraticCost(Q + 1e-300 * np.eye(n, k=1), R, Qf).c_struct().diagonal == 0

    class Unhinted(models.QuadraticCost):
        def c_struct(self):
            s = super().c_struct()
            s.diagonal = 0
            assert s._owner is self        # the handle keeps the device copies of Q, R, Qf alive
            return s

    pol, prev = pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev)
    traj, J = ilqr.forwardPass2(x0, model, cost, pol, prev)
    traj0, J0 = ilqr.forwardPass2(x0, model, Unhinted(Q, R, Qf), pol, prev)
    assert np.array_equal(J, J0) and np.array_equal(traj.xTraj, traj0.xTraj) and np.array_equal(traj.uTraj, traj0.uTraj)
    for b in range(batch):
        rt, rJ = zo.forwardPass2(x0[b], f, cost.runningCost, cost.terminalCost, zo.AffinePolicy(l[b], L[b]),
                                 zo.Trajectory(xPrev[b], uPrev[b]))
        assert abs(J[b] - rJ) <= 1e-10 * abs(rJ)
        assert _rel(traj.xTraj[b], rt.xTraj) <= 1e-9 and _rel(traj.uTraj[b], rt.uTraj) <= 1e-9


@pytest.mark.parametrize("N", [1, 2, 3, 7, 9, 30])     # every remainder of the kernels' three- and four-step loop bodies
def test_reroll_as_its_own_launch_matches_the_second_pass(mods, N):
    """zm_rollout_linesearch_list_f64 with an alpha_idx buffer re-rolls the winners in rollout_quad_reroll_kernel (four lanes per
    rollout, 16 trajectories per wave); without one the second pass runs inside the line-search kernel.  Same bits, same winners,
    also under a list with a mask and for horizons that exercise the two-steps-per-iteration loop's tail."""
    import ctypes
    import torch
    from zopt_amd import _lib
    ilqr, models, pt = mods
    rng = np.random.default_rng(100 + N)
    batch = 37
    x0, l, L, xPrev, uPrev = _quad_problem(rng, batch, N)
    l *= 60.0
    cost = models.QuadraticCost(np.diag(rng.uniform(0.5, 2.0, 12)), np.diag(rng.uniform(0.5, 2.0, 4)), np.diag(rng.uniform(5.0, 20.0, 12)))
    md, cs = models.QuadcopterEuler(0.1).c_struct(), cost.c_struct()
    dev = [torch.as_tensor(np.ascontiguousarray(X), device="cuda") for X in (x0, l, L, xPrev, uPrev)]
    al = torch.as_tensor(0.5 ** np.arange(16), device="cuda")
    lst = torch.as_tensor(np.array([3, 0, 36, 17, 18, 19, 20, 5, 6, 7, 30, 31, 8, 9, 10, 11, 12, 1, 2], dtype=np.int32), device="cuda")
    act = torch.ones(batch, dtype=torch.int32, device="cuda")
    act[17] = 0
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for with_idx in (False, True):
        xT = torch.full((batch, N + 1, 12), -7.0, dtype=torch.float64, device="cuda")
        uT = torch.full((batch, N, 4), -7.0, dtype=torch.float64, device="cuda")
        J = torch.full((batch,), -7.0, dtype=torch.float64, device="cuda")
        idx = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib().zm_rollout_linesearch_list_f64(
            ctypes.addressof(md), ctypes.addressof(cs), *[t.data_ptr() for t in dev], al.data_ptr(), 16, lst.data_ptr(), lst.numel(),
            act.data_ptr(), xT.data_ptr(), uT.data_ptr(), J.data_ptr(), idx.data_ptr() if with_idx else None, batch, N, st), "rollout")
        torch.cuda.synchronize()
        out.append((xT.cpu().numpy(), uT.cpu().numpy(), J.cpu().numpy(), idx.cpu().numpy()))
    (x1, u1, J1, _), (x2, u2, J2, i2) = out
    assert np.array_equal(x1, x2) and np.array_equal(u1, u2) and np.array_equal(J1, J2)
    listed = sorted(set(lst.cpu().numpy().tolist()) - {17})
    assert np.all(x2[17] == -7.0) and np.all(i2[listed] >= 0) and len(set(i2[listed].tolist())) > 1   # several winners, not only alpha_0
    untouched = sorted(set(range(batch)) - set(lst.cpu().numpy().tolist()))
    assert np.all(x2[untouched] == -7.0) and np.all(J2[untouched] == -7.0)


def test_fast_rollout_kernels_random_horizons_and_batches(mods):
    """Seeded sweep over horizons (every remainder of the four-step loop body of rollout_ls_fast_kernel and of the three-step body of the
    four-lane kernels) and batch sizes (groups that do not fill a wave, more waves than one) for the (12, 4) line search: forwardPass2 and
    a single-step-size trajectoryRollout of the quadcopter and of a linear model, general and diagonal weights, against the oracle."""
    ilqr, models, pt = mods
    rng = np.random.default_rng(2025)
    for case in range(10):
        N = int(rng.integers(1, 23))
        batch = int(rng.choice([1, 2, 3, 5, 17, 66]))
        kind = "quadcopter" if case % 2 == 0 else "linear"
        if kind == "linear":
            model = models.LinearModel(rng.standard_normal((12, 12)) * (1.0 / np.sqrt(12)), rng.standard_normal((12, 4)))
            x0, l, L, xPrev, uPrev = _random_policy_problem(rng, batch, N, 12, 4)
            l *= 4.0
            f = model
        else:
            model = models.QuadcopterEuler(0.1)
            x0, l, L, xPrev, uPrev = _quad_problem(rng, batch, N)
            l *= 40.0
            f = zo.quad_euler_step(0.1)
        if case % 3 == 0:
            cost = models.QuadraticCost(np.diag(rng.uniform(0.5, 2.0, 12)), np.diag(rng.uniform(0.5, 2.0, 4)), np.diag(rng.uniform(5.0, 20.0, 12)))
        else:
            Mq, Mr = rng.standard_normal((12, 12)), rng.standard_normal((4, 4))
            cost = models.QuadraticCost(Mq @ Mq.T / 12 + np.eye(12), Mr @ Mr.T / 4 + np.eye(4), 10 * np.eye(12))
        traj, J = ilqr.forwardPass2(x0, model, cost, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev))
        one = ilqr.trajectoryRollout(x0, model, pt.AffinePolicy(l, L), pt.Trajectory(xPrev, uPrev), alpha=0.25)
        for b in sorted(set([0, batch // 2, batch - 1])):
            rt, rJ = zo.forwardPass2(x0[b], f, cost.runningCost, cost.terminalCost, zo.AffinePolicy(l[b], L[b]), zo.Trajectory(xPrev[b], uPrev[b]))
            assert abs(J[b] - rJ) <= 1e-10 * abs(rJ), (case, kind, N, batch, b)
            assert _rel(traj.xTraj[b], rt.xTraj) <= 1e-9 and _rel(traj.uTraj[b], rt.uTraj) <= 1e-9, (case, kind, N, batch, b)
            r1 = zo.trajectoryRollout(x0[b], f, zo.AffinePolicy(l[b], L[b]), zo.Trajectory(xPrev[b], uPrev[b]), 0.25)
            assert _rel(one.xTraj[b], r1.xTraj) <= 1e-9 and _rel(one.uTraj[b], r1.uTraj) <= 1e-9, (case, kind, N, batch, b)
