"""GPU tests for K9 mpc_box_qp (class lqrMpc, reference mpcUtils.py:12-81).

The reference's arithmetic here is OSQP's (not in the reference tree, not installed): numeric parity is UNPINNED.  The tests
therefore use (i) the reference's own test problem with its hand-derived optimum, (ii) solver-independent KKT certificates,
(iii) an independent SciPy solve of the condensed QP, (iv) iterate-level agreement with the NumPy restatement of the ADMM."""
import numpy as np
import pytest

from oracle import mpc_oracle as mo
from oracle import zopt_oracle as zo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mpc():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import mpcUtils
    return mpcUtils


def test_kat_reference_problem(mpc):
    """reference tests/test_mpcUtils.py:9-23 -> status "optimal"; no bound is active at the optimum, so the answer is the
    Riccati one: u0 = (-0.6,-0.6), x = [(1,1),(.4,.4),(.2,.2)], u = [(-.6,-.6),(-.2,-.2)], cost 3.2."""
    I, one = np.eye(2), np.ones(2)
    prob = mpc.lqrMpc(I, I, I, I, 2, -one, one, -one, one)
    u, traj, status = prob.solve(one)
    assert status == "optimal"
    assert u == pytest.approx([-0.6, -0.6], abs=1e-4)
    assert traj.xTraj == pytest.approx(np.array([[1, 1], [.4, .4], [.2, .2]]), abs=1e-4)
    assert traj.uTraj == pytest.approx(np.array([[-.6, -.6], [-.2, -.2]]), abs=1e-4)
    assert mo.cost(I, I, I, traj.xTraj, traj.uTraj) == pytest.approx(3.2, abs=1e-6)
    u2, traj2, status2 = prob.solve(one, solver="OSQP", eps_abs=1e-9, eps_rel=1e-9)
    assert status2 == "optimal" and u2 == pytest.approx([-0.6, -0.6], abs=1e-8)


def _random_problem(rng, n, m, N):
    G = rng.standard_normal((n, n))
    A = 0.95 * G / np.max(np.abs(np.linalg.eigvals(G)))      # stable: every start inside the box admits a feasible trajectory
    B = rng.standard_normal((n, m))
    Mq, Mr = rng.standard_normal((n, n)), rng.standard_normal((m, m))
    Q, R = Mq @ Mq.T / n + np.eye(n), Mr @ Mr.T / m + 0.5 * np.eye(m)
    Qf = 3 * Q
    return A, B, Q, R, Qf


@pytest.mark.parametrize("n,m,N", [(2, 1, 6), (2, 2, 5), (4, 2, 6), (4, 1, 8), (1, 1, 4), (16, 5, 5), (24, 8, 4), (13, 2, 6)])
def test_constrained_small_problems_against_independent_solve(mpc, n, m, N):
    """Active input / state bounds: KKT certificate for every instance, independent SciPy solve for three of them.  The last three
    shapes lie beyond the 16-index tile of the 16-lanes-per-instance kernel: they run the lane-per-instance kernel at (24, 8)
    (smaller ones embedded with inert padding), fixed penalty."""
    rng = np.random.default_rng(10 * n + m)
    A, B, Q, R, Qf = _random_problem(rng, n, m, N)
    x_ub = np.full(n, 4.0); u_ub = np.full(m, 0.15)
    prob = mpc.lqrMpc(A, B, Q, R, N, -x_ub, x_ub, -u_ub, u_ub, Qf=Qf)
    large = n > 12 or m > 4           # (the SciPy reference solve takes tens of seconds at these sizes: one instance of them gets it)
    nb, eps, max_ref = (4, 1e-6, 1 if n < 20 else 0) if large else (8, 1e-6, 3)     # (24, 8): KKT certificate only (SciPy: a minute)
    x0 = rng.uniform(-1.0, 1.0, (nb, n))
    u0, traj, status = prob.solve(x0, eps_abs=eps, eps_rel=eps, max_iter=30000)
    n_active = n_ref = 0
    for b in range(nb):
        if status[b] != "optimal":
            continue
        x, u = traj.xTraj[b], traj.uTraj[b]
        kkt = mo.kkt_residuals(A, B, Q, R, Qf, N, -x_ub, x_ub, -u_ub, u_ub, x0[b], x, u, act_tol=1e-4)
        assert kkt["dyn"] <= 1e-12 and kkt["bound"] <= 1e-4 and kkt["stat"] <= 1e-3
        active = bool(np.max(np.abs(u)) >= 0.15 - 1e-5 or np.max(np.abs(x[1:])) >= 4.0 - 1e-5)
        n_active += int(active)
        if active and n_ref < max_ref:
            n_ref += 1
            xr, ur, fr = mo.solve_reference(A, B, Q, R, Qf, N, -x_ub, x_ub, -u_ub, u_ub, x0[b])
            assert np.max(np.abs(u - ur)) <= 2e-3
            assert mo.cost(Q, R, Qf, x, u) <= fr + 1e-4 * max(1.0, fr)
    assert n_active >= 1 and np.all(status == "optimal")
    assert n_ref == min(max_ref, n_active)


def _quad_mpc(mpc, N=30):
    """BASELINE config 3 / demos/lqrMpc.py:13-29: 12-state quadcopter linearised at hover (explicit Jacobian of
    inertialDynamics, quirk Q6), forward Euler dt = 0.1, Q = I, R = I, the demo's bounds."""
    dt = 0.1
    uTrim = np.array([9.807, 0, 0, 0.0])
    _, Aw, Bw = zo.jacobians(zo.quad_inertialDynamics, np.zeros(12), uTrim)
    A, B = np.eye(12) + dt * Aw, dt * Bw
    x_ub = np.array([1, 1, 1, 0.3, 0.3, 0.1, 0.5, 0.5, np.inf, np.inf, np.inf, np.inf])
    u_ub = np.array([3.0, 3, 3, 3])
    Q, R = np.eye(12), np.eye(4)
    return mpc.lqrMpc(A, B, Q, R, N, -x_ub, x_ub, -u_ub, u_ub), (A, B, Q, R, Q, x_ub, u_ub)


def test_quadcopter_config3_batch(mpc):
    prob, (A, B, Q, R, Qf, x_ub, u_ub) = _quad_mpc(mpc)
    N = 30
    rng = np.random.default_rng(1)
    Bn = 1024
    x0 = np.clip(0.03 * rng.standard_normal((Bn, 12)), -x_ub + 1e-6, x_ub - 1e-6)   # small: near a bound with outward rates is infeasible
    x0[:, 9:12] = rng.uniform(-10, 10, (Bn, 3))
    u0, traj, status = prob.solve(x0, solver="OSQP", eps_abs=1e-2, eps_rel=1e-2)      # the demo's tolerances
    assert np.all(status == "optimal")
    x, u = traj.xTraj, traj.uTraj
    assert x.shape == (Bn, N + 1, 12) and u.shape == (Bn, N, 4) and u0.shape == (Bn, 4)
    assert np.max(np.abs(x[:, 1:] - (np.einsum('ij,bkj->bki', A, x[:, :-1]) + np.einsum('ij,bkj->bki', B, u)))) <= 1e-11
    tol = 1e-2 + 1e-2 * np.max(np.abs(x)) + 1e-9      # the primal tolerance the solve was asked for (eps_abs + eps_rel |w|)
    assert np.max(np.maximum(np.abs(x) - x_ub, 0)) <= tol and np.max(np.maximum(np.abs(u) - u_ub, 0)) <= tol
    # tight solve of a few instances: KKT certificate + iterate-level agreement with the NumPy restatement
    u0t, trajt, statt = prob.solve(x0[:4], eps_abs=1e-4, eps_rel=1e-4, max_iter=100000, adaptive_rho=False)
    assert np.all(statt == "optimal")
    for b in range(4):
        kkt = mo.kkt_residuals(A, B, Q, R, Qf, N, -x_ub, x_ub, -u_ub, u_ub, x0[b], trajt.xTraj[b], trajt.uTraj[b], act_tol=5e-3)
        assert kkt["dyn"] <= 1e-11 and kkt["bound"] <= 2e-3 and kkt["stat"] <= 2e-2
    # iterate-level agreement: the same number of ADMM iterations and the same iterate, with OSQP's default over-relaxation
    # (alpha = 1.6, the solve's default) and without (alpha = 1)
    its_by_alpha = {}
    for alpha in (1.6, 1.0):
        _, tra, sta = prob.solve(x0[:1], eps_abs=1e-4, eps_rel=1e-4, max_iter=100000, adaptive_rho=False, alpha=alpha, warm_start=False)
        xo, uo, so, ito = mo.admm(A, B, Q, R, Qf, N, -x_ub, x_ub, -u_ub, u_ub, x0[0], rho=prob.rho, eps_abs=1e-4, eps_rel=1e-4,
                                  max_iter=100000, alpha=alpha)
        assert so == "optimal" and sta[0] == "optimal" and ito == int(prob.last_iterations[0])
        assert np.max(np.abs(tra.uTraj[0] - uo)) <= 1e-9 and np.max(np.abs(tra.xTraj[0] - xo)) <= 1e-9
        its_by_alpha[alpha] = ito
    assert its_by_alpha[1.6] < its_by_alpha[1.0]          # what the relaxation is for
    with pytest.raises(ValueError):
        prob.solve(x0[:1], alpha=2.0)
    # the loose (demo tolerance) solutions cost about the same as the tight ones
    for b in range(4):
        ct, cl = mo.cost(Q, R, Qf, trajt.xTraj[b], trajt.uTraj[b]), mo.cost(Q, R, Qf, x[b], u[b])
        assert abs(cl - ct) <= 5e-2 * ct


def test_unbounded_box_equals_lqr(mpc):
    """No active constraint => the QP solution is the finite-horizon LQR rollout."""
    rng = np.random.default_rng(5)
    n, m, N = 4, 2, 10
    A, B, Q, R, Qf = _random_problem(rng, n, m, N)
    inf_n, inf_m = np.full(n, np.inf), np.full(m, np.inf)
    prob = mpc.lqrMpc(A, B, Q, R, N, -inf_n, inf_n, -inf_m, inf_m, Qf=Qf)
    x0 = rng.standard_normal((5, n))
    u0, traj, status = prob.solve(x0, eps_abs=1e-10, eps_rel=1e-10)
    assert np.all(status == "optimal")
    for b in range(5):
        xr, ur, fr = mo.solve_reference(A, B, Q, R, Qf, N, -inf_n, inf_n, -inf_m, inf_m, x0[b])
        assert np.max(np.abs(traj.uTraj[b] - ur)) <= 1e-7


def test_infeasible_instances(mpc):
    I = np.eye(2)
    one = np.ones(2)
    # (a) x0 outside its own bounds (the QP constrains x_0 too: mpcUtils.py:56,58)
    prob = mpc.lqrMpc(I, I, I, I, 3, -one, one, -one, one)
    u, traj, status = prob.solve(np.array([[2.0, 0.0], [0.5, 0.5]]))
    assert list(status) == ["infeasible", "optimal"]
    # (b) dynamics + bounds admit no trajectory: x1 = 2 x0 + u with |u| <= 0.1 leaves |x| <= 1
    prob = mpc.lqrMpc(2 * I, I, I, I, 3, -one, one, -0.1 * one, 0.1 * one)
    u, traj, status = prob.solve(np.array([[0.9, 0.9], [0.01, -0.01]]))
    assert status[0] == "infeasible" and status[1] == "optimal"


def _lp_feasible(A, B, Q, R, Qf, N, x_ub, u_ub, x0):
    """Independent feasibility oracle: HiGHS LP on the condensed constraints."""
    import scipy.optimize as spo
    Phi, Gam, H, g, c = mo.condense(A, B, Q, R, Qf, N, x0)
    rows, hi, lo = [], [], []
    for k in range(1, N + 1):
        for i in np.where(np.isfinite(x_ub))[0]:
            rows.append(Gam[k][i])
            off = (Phi[k] @ x0)[i]
            hi.append(x_ub[i] - off)
            lo.append(-x_ub[i] - off)
    Aub = np.vstack([np.array(rows), -np.array(rows)])
    bub = np.concatenate([hi, -np.array(lo)])
    m = len(u_ub)
    lp = spo.linprog(np.zeros(N * m), A_ub=Aub, b_ub=bub, bounds=[(-u_ub[j % m], u_ub[j % m]) for j in range(N * m)],
                     method="highs")
    return lp.status == 0


def test_receding_horizon_loop(mpc):
    """demos/lqrMpc.py:40-47: clip, solve, "assume perfect tracking" x <- xMpc[i][1]; 25 steps on a batch.  A clipped state
    with outward rates can make the next QP infeasible (the reference demo never checks its status).  Soundness: every
    instance reported "infeasible" must be LP-infeasible; hard ones that are not certified in time report "user_limit"."""
    prob, (A, B, Q, R, Qf, x_ub, u_ub) = _quad_mpc(mpc, N=25)
    rng = np.random.default_rng(3)
    x = np.zeros((32, 12))
    x[:, 9:12] = rng.uniform(-10, 10, (32, 3))
    d0 = np.linalg.norm(x[:, 9:12], axis=1)
    alive = np.ones(32, dtype=bool)
    checked = 0
    for _ in range(25):
        x = np.clip(x, -x_ub + 1e-6, x_ub - 1e-6)
        u, traj, status = prob.solve(x, solver="OSQP", eps_prim_inf=1e-3, eps_dual_inf=1e-3, eps_abs=1e-2, eps_rel=1e-2,
                                     max_iter=2000, warm_start=False)      # the demo's options (demos/lqrMpc.py:32)
        assert set(status[alive]) <= {"optimal", "optimal_inaccurate", "infeasible", "user_limit"}
        for b in np.where(alive & (status == "infeasible"))[0][:2]:
            if checked < 4:
                assert not _lp_feasible(A, B, Q, R, Qf, 25, x_ub, u_ub, x[b])
                checked += 1
        alive &= (status == "optimal")
        x = np.where(alive[:, None], traj.xTraj[:, 1], x)
    assert alive.sum() >= 16
    assert np.all(np.linalg.norm(x[alive, 9:12], axis=1) < d0[alive])      # every surviving instance moved towards the origin


def test_status_vocabulary_at_the_iteration_limit(mpc):
    """cvxpy's names for OSQP's outcomes at `max_iter` (mpcUtils.py:74,78): "optimal_inaccurate" (OSQP "solved inaccurate") when both
    residuals are within 10x their tolerances, "user_limit" otherwise; with enough iterations the same instances are "optimal".  The
    residuals the kernel reports (`last_residuals`) decide which."""
    prob, (A, B, Q, R, Qf, x_ub, u_ub) = _quad_mpc(mpc, N=25)
    rng = np.random.default_rng(4)
    x = np.zeros((64, 12))
    x[:, 9:12] = rng.uniform(-10, 10, (64, 3))
    _, _, full = prob.solve(x, eps_abs=1e-3, eps_rel=1e-3, max_iter=4000, warm_start=False)
    its = prob.last_iterations.copy()
    assert np.all(full == "optimal")
    seen = set()
    for cap in (5, 15, 30, 60):
        _, _, st = prob.solve(x, eps_abs=1e-3, eps_rel=1e-3, max_iter=cap, warm_start=False)
        assert set(st) <= {"optimal", "optimal_inaccurate", "user_limit"}
        assert np.all((st == "optimal") == (its <= cap))               # the cap only cuts the iteration short
        assert np.all(prob.last_iterations[st != "optimal"] == cap)
        seen |= set(st)
    assert {"optimal_inaccurate", "user_limit"} <= seen


def test_warm_start_reuses_previous_iterates(mpc):
    """cvxpy's default warm_start=True (mpcUtils.py:77 forwards **kwargs): a second solve of the same batch shape starts from
    the previous ADMM iterates (same x0, tighter tolerance: fewer iterations than from zero); warm_start="shift" advances
    them by one step for the receding-horizon loop.  Same answers as cold solves within the tolerance; an instance that
    was not "optimal" last time starts cold again."""
    prob, (A, B, Q, R, Qf, x_ub, u_ub) = _quad_mpc(mpc, N=25)
    rng = np.random.default_rng(4)
    x = np.zeros((16, 12))
    x[:, 9:12] = rng.uniform(-5, 5, (16, 3))
    loose = dict(eps_abs=1e-3, eps_rel=1e-3, max_iter=200000)
    tight = dict(eps_abs=1e-5, eps_rel=1e-5, max_iter=200000)
    # (a) refine the same problem
    _, _, s0 = prob.solve(x, warm_start=False, **loose)
    assert np.all(s0 == "optimal")
    ur, tr, sr = prob.solve(x, **tight)                   # warm (default)
    it_refine = prob.last_iterations.copy()
    uc, tc, sc = prob.solve(x, warm_start=False, **tight)
    it_cold = prob.last_iterations.copy()
    assert np.all(sr == "optimal") and np.all(sc == "optimal")
    assert np.max(np.abs(tr.uTraj - tc.uTraj)) <= 1e-2 and np.max(np.abs(tr.xTraj - tc.xTraj)) <= 1e-2
    assert it_refine.sum() < it_cold.sum()
    # cold start is reproducible bit for bit
    _, tc2, _ = prob.solve(x, warm_start=False, **tight)
    assert np.array_equal(tc2.uTraj, tc.uTraj) and np.array_equal(prob.last_iterations, it_cold)
    # (b) receding horizon at the demo's tolerance: x0 <- x_1 of the plan ("assume perfect tracking", demos/lqrMpc.py:47),
    #     iterates shifted by one step
    demo = dict(eps_abs=1e-2, eps_rel=1e-2, max_iter=200000)
    _, t0, s0 = prob.solve(x, warm_start=False, **demo)
    x1 = t0.xTraj[:, 1]
    us, ts, ss = prob.solve(x1, warm_start="shift", **demo)
    it_shift = prob.last_iterations.copy()
    uc1, tc1, sc1 = prob.solve(x1, warm_start=False, **demo)
    it_cold1 = prob.last_iterations.copy()
    assert np.all(s0 == "optimal") and np.all(ss == "optimal") and np.all(sc1 == "optimal")
    cw = np.array([mo.cost(Q, R, Qf, ts.xTraj[b], ts.uTraj[b]) for b in range(16)])
    cc = np.array([mo.cost(Q, R, Qf, tc1.xTraj[b], tc1.uTraj[b]) for b in range(16)])
    assert np.all(np.abs(cw - cc) <= 5e-2 * np.maximum(cc, 1.0))       # both are 1e-2-accurate solutions of the same QP
    assert it_shift.sum() < 0.5 * it_cold1.sum()
    _, tc1, _ = prob.solve(x1, warm_start=False, **tight)
    it_cold1 = prob.last_iterations.copy()
    # (c) an infeasible instance does not poison the next solve of its slot
    xb = x1.copy()
    xb[0, 0] = 5.0                                        # outside its bound
    _, _, sb = prob.solve(xb, warm_start=False, **tight)
    assert sb[0] == "infeasible"
    un, tn, sn = prob.solve(x1, **tight)                  # warm: slot 0 restarts cold
    assert np.all(sn == "optimal") and np.max(np.abs(tn.uTraj - tc1.uTraj)) <= 1e-2
    assert prob.last_iterations[0] == it_cold1[0]


def test_adaptive_rho_default_options(mpc):
    """cvxpy's defaults (OSQP: eps 1e-5, max_iter 10000, adaptive_rho on) -- the options the reference's own unit test runs
    with (tests/test_mpcUtils.py:22-23, which asserts status == "optimal").  With a fixed penalty this problem needs > 10^4
    iterations at that tolerance; with the adaptive levels every instance reports "optimal" well inside the default limit, and
    the answer satisfies the KKT conditions and matches the fixed-penalty solution."""
    prob, (A, B, Q, R, Qf, x_ub, u_ub) = _quad_mpc(mpc)
    N = 30
    rng = np.random.default_rng(1)
    x0 = np.clip(0.03 * rng.standard_normal((64, 12)), -x_ub + 1e-6, x_ub - 1e-6)
    x0[:, 9:12] = rng.uniform(-10, 10, (64, 3))
    u0, traj, status = prob.solve(x0)                                    # all defaults
    it_ad = prob.last_iterations.copy()
    assert np.all(status == "optimal") and it_ad.max() < 10000
    for b in range(4):
        kkt = mo.kkt_residuals(A, B, Q, R, Qf, N, -x_ub, x_ub, -u_ub, u_ub, x0[b], traj.xTraj[b], traj.uTraj[b], act_tol=1e-3)
        assert kkt["dyn"] <= 1e-11 and kkt["bound"] <= 5e-4 and kkt["stat"] <= 5e-3
    uf, tf, sf = prob.solve(x0[:8], adaptive_rho=False, max_iter=400000, warm_start=False)
    it_fx = prob.last_iterations.copy()
    assert np.all(sf == "optimal")
    assert np.max(np.abs(tf.uTraj - traj.uTraj[:8])) <= 5e-3 and np.max(np.abs(tf.xTraj - traj.xTraj[:8])) <= 5e-3
    assert it_ad[:8].sum() < 0.2 * it_fx.sum()


def test_bad_arguments(mpc):
    I, one = np.eye(2), np.ones(2)
    prob = mpc.lqrMpc(I, I, I, I, 2, -one, one, -one, one)
    with pytest.raises(ValueError):
        prob.solve(np.ones(3))
    with pytest.raises(ValueError):
        prob.solve(one, solver="CLARABEL")
    with pytest.raises(ValueError):
        mpc.lqrMpc(np.eye(25), np.ones((25, 2)), np.eye(25), np.eye(2), 2, -np.ones(25), np.ones(25), -np.ones(2), np.ones(2)).solve(np.ones(25))


@pytest.mark.parametrize("n,m", [(3, 2), (5, 3), (6, 1), (9, 4), (10, 2), (3, 3)])
def test_shapes_between_the_compiled_kernels(mpc, n, m):
    """Any n <= 12, m <= 4: shapes without a kernel of their own are embedded in the next compiled one (inert padding);
    answers are checked against the condensed-QP reference solve and the KKT conditions."""
    rng = np.random.default_rng(100 + 10 * n + m)
    N = 8
    A, B, Q, R, Qf = _random_problem(rng, n, m, N)
    x_ub, u_ub = np.full(n, 3.0), np.full(m, 0.4)
    prob = mpc.lqrMpc(A, B, Q, R, N, -x_ub, x_ub, -u_ub, u_ub, Qf=Qf)
    x0 = 0.5 * rng.standard_normal((5, n))
    u0, traj, status = prob.solve(x0, eps_abs=1e-7, eps_rel=1e-7, max_iter=100000)
    assert u0.shape == (5, m) and traj.xTraj.shape == (5, N + 1, n) and traj.uTraj.shape == (5, N, m)
    for b in range(5):
        if status[b] != "optimal":
            continue
        kkt = mo.kkt_residuals(A, B, Q, R, Qf, N, -x_ub, x_ub, -u_ub, u_ub, x0[b], traj.xTraj[b], traj.uTraj[b], act_tol=1e-5)
        assert kkt["dyn"] <= 1e-12 and kkt["bound"] <= 1e-5 and kkt["stat"] <= 1e-4
    assert np.sum(status == "optimal") >= 3
    b = int(np.where(status == "optimal")[0][0])
    xr, ur, fr = mo.solve_reference(A, B, Q, R, Qf, N, -x_ub, x_ub, -u_ub, u_ub, x0[b])
    assert np.max(np.abs(traj.uTraj[b] - ur)) <= 2e-4
