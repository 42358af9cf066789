"""GPU tests for K5 psd_project, K7 linearize/quadratize and the A8 iterativeLqr driver (reference ilqrUtils.py:217-327,
pytrees.py:72-153)."""
import numpy as np
import pytest

from oracle import zopt_oracle as zo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import _lib, ilqrUtils, models, pytrees
    return ilqrUtils, models, pytrees, _lib


# ---------------------------------------------------------------- K5
@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 12, 15, 16])
def test_ensurePositiveDefinite_parity(mods, k):
    ilqr = mods[0]
    rng = np.random.default_rng(k)
    M = rng.standard_normal((7, k, k))
    A = M + np.swapaxes(M, -1, -2)                      # symmetric indefinite: the clamp is active
    A[0] = M[0]                                          # nonsymmetric input: eigh symmetrises it
    A[1] = M[1] @ M[1].T + np.eye(k)                     # already PD: (numerically) unchanged
    A[2] = np.diag(np.linspace(-1.0, 2.0, k))            # diagonal
    out = ilqr.ensurePositiveDefinite(A)
    ref = zo.ensurePositiveDefinite(A)
    assert out.shape == A.shape
    assert np.max(np.abs(out - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))
    w = np.linalg.eigvalsh(0.5 * (out + np.swapaxes(out, -1, -2)))
    assert np.min(w) >= 1e-3 * (1 - 1e-9)
    assert out[1] == pytest.approx(A[1], rel=1e-12, abs=1e-13)


def test_conditionQuadraticCost_parity(mods):
    ilqr, _, pt, _ = mods
    rng = np.random.default_rng(1)
    b, N, n, m = 3, 5, 12, 4
    M = rng.standard_normal((b, N, n + m, n + m))
    H = M + np.swapaxes(M, -1, -2)
    cost = (rng.standard_normal((b, N)), rng.standard_normal((b, N, n)), rng.standard_normal((b, N, m)),
            np.ascontiguousarray(H[..., :n, :n]), np.ascontiguousarray(H[..., n:, :n]), np.ascontiguousarray(H[..., n:, n:]))
    out = ilqr.conditionQuadraticCost(pt.QuadraticCostFunction(*cost))
    ref = zo.conditionQuadraticCost(zo.QuadraticCostFunction(*cost))
    for name in ("c_xx", "c_ux", "c_uu"):
        assert _rel(getattr(out, name), getattr(ref, name)) <= 1e-12
    assert out.c is cost[0] and out.c_x is cost[1]
    Vf = ilqr.conditionValueFunction(pt.QuadraticValueFunction(0.0, np.zeros(n), H[0, 0, :n, :n]))
    assert _rel(Vf.v_xx, zo.ensurePositiveDefinite(H[0, 0, :n, :n])) <= 1e-12


# ---------------------------------------------------------------- K7
def _lin_call(mods, model, xT, uT):
    import ctypes
    import torch
    _, _, _, _lib = mods
    b, Np1, n = xT.shape
    N, m = Np1 - 1, uT.shape[-1]
    dx, du = torch.as_tensor(xT, device="cuda"), torch.as_tensor(uT, device="cuda")
    f = torch.empty((b, N, n), dtype=torch.float64, device="cuda")
    f_x = torch.empty((b, N, n, n), dtype=torch.float64, device="cuda")
    f_u = torch.empty((b, N, n, m), dtype=torch.float64, device="cuda")
    md = model.c_struct()
    rc = _lib.lib().zm_linearize_dynamics_f64(ctypes.addressof(md), dx.data_ptr(), du.data_ptr(), None, f.data_ptr(),
                                              f_x.data_ptr(), f_u.data_ptr(), b, N, None)
    assert rc == 0
    torch.cuda.synchronize()
    return f.cpu().numpy(), f_x.cpu().numpy(), f_u.cpu().numpy()


def test_linearize_quadcopter_matches_complex_step(mods):
    """Forward-mode dual numbers on the device model vs the oracle's complex-step Jacobians (pytrees.py:139-153)."""
    models = mods[1]
    rng = np.random.default_rng(5)
    b, N = 3, 7
    xT = 0.4 * rng.standard_normal((b, N + 1, 12))
    uT = np.array([9.807, 0, 0, 0]) + 0.5 * rng.standard_normal((b, N, 4))
    f, f_x, f_u = _lin_call(mods, models.QuadcopterEuler(0.1), xT, uT)
    step = zo.quad_euler_step(0.1)
    for i in range(b):
        ref = zo.affine_dynamics_from_trajectory(step, zo.Trajectory(xT[i], uT[i]))
        assert _rel(f[i], ref.f) <= 1e-13 and _rel(f_x[i], ref.f_x) <= 1e-12 and _rel(f_u[i], ref.f_u) <= 1e-12


def test_linearize_linear_model_is_exact(mods):
    models = mods[1]
    rng = np.random.default_rng(6)
    A, B = rng.standard_normal((5, 5)), rng.standard_normal((5, 2))
    xT, uT = rng.standard_normal((2, 4, 5)), rng.standard_normal((2, 3, 2))
    f, f_x, f_u = _lin_call(mods, models.LinearModel(A, B), xT, uT)
    assert np.array_equal(f_x, np.broadcast_to(A, f_x.shape)) and np.array_equal(f_u, np.broadcast_to(B, f_u.shape))
    assert f == pytest.approx(np.einsum('ij,bkj->bki', A, xT[:, :-1]) + np.einsum('ij,bkj->bki', B, uT), rel=1e-14)


def test_quadratize_cost_parity(mods):
    import ctypes
    import torch
    _, models, _, _lib = mods
    rng = np.random.default_rng(8)
    b, N, n, m = 4, 6, 12, 4
    Q, R, Qf = rng.standard_normal((n, n)), rng.standard_normal((m, m)), rng.standard_normal((n, n))   # nonsymmetric on purpose
    cost = models.QuadraticCost(Q, R, Qf)
    xT, uT = rng.standard_normal((b, N + 1, n)), rng.standard_normal((b, N, m))
    t = lambda *s: torch.empty(s, dtype=torch.float64, device="cuda")
    c, c_x, c_u, v, v_x = t(b, N), t(b, N, n), t(b, N, m), t(b), t(b, n)
    c_xx, c_ux, c_uu, v_xx = t(n, n), t(m, n), t(m, m), t(n, n)
    cs = cost.c_struct()
    dx, du = torch.as_tensor(xT, device="cuda"), torch.as_tensor(uT, device="cuda")
    rc = _lib.lib().zm_quadratize_cost_f64(ctypes.addressof(cs), n, m, dx.data_ptr(), du.data_ptr(), None, c.data_ptr(),
                                           c_x.data_ptr(), c_u.data_ptr(), v.data_ptr(), v_x.data_ptr(), c_xx.data_ptr(),
                                           c_ux.data_ptr(), c_uu.data_ptr(), v_xx.data_ptr(), b, N, None)
    assert rc == 0
    torch.cuda.synchronize()
    for i in range(b):
        ref = zo.quadratic_cost_from_trajectory(Q, R, zo.Trajectory(xT[i], uT[i]))
        vf = zo.terminal_value_function(Qf, xT[i, -1])
        assert _rel(c[i].cpu().numpy(), ref.c) <= 1e-13 and _rel(c_x[i].cpu().numpy(), ref.c_x) <= 1e-13
        assert _rel(c_u[i].cpu().numpy(), ref.c_u) <= 1e-13
        assert abs(v[i].item() - vf.v) <= 1e-13 * abs(vf.v) and _rel(v_x[i].cpu().numpy(), vf.v_x) <= 1e-13
    assert np.array_equal(c_xx.cpu().numpy(), Q + Q.T) and np.array_equal(c_uu.cpu().numpy(), R + R.T)
    assert np.array_equal(v_xx.cpu().numpy(), Qf + Qf.T) and not c_ux.cpu().numpy().any()


# ---------------------------------------------------------------- A8
def test_kat_iterativeLqr(mods):
    """reference tests/test_ilqrUtils.py:167-181: A=B=Q=R=I, N=3, x0=(2,1), uGuess=0 -> converged.  The problem is LQ, so
    the result must also be the Riccati-optimal trajectory."""
    ilqr, models, pt, _ = mods
    I = np.eye(2)
    cost = models.QuadraticCost(I, I, I)
    x0 = np.array([2.0, 1.0])
    traj, L, J, converged = ilqr.iterativeLqr(models.LinearModel(I, I), cost.runningCost, cost.terminalCost, x0,
                                              np.zeros((3, 2)))
    assert converged is True
    rt, rL, rJ, rc = zo.iterativeLqr(lambda x, u: x + u, I, I, I, x0, np.zeros((3, 2)))
    assert rc
    assert _rel(traj.xTraj, rt.xTraj) <= 1e-9 and _rel(traj.uTraj, rt.uTraj) <= 1e-9 and _rel(L, rL) <= 1e-9
    assert J == pytest.approx(rJ, rel=1e-10)
    assert isinstance(traj, pt.Trajectory) and traj.xTraj.shape == (4, 2) and L.shape == (3, 2, 2)


def test_iterativeLqr_quadcopter_demo_problem(mods):
    """demos/iterativeLqr.py:22-39: quadcopter, dt=0.1, N=100, Q=I12, R=I4, terminal 10 x'Qx, x0[9:12]=(10,10,10),
    uGuess=uTrim -- plus perturbed starts, against the CPU oracle loop (same iteration-by-iteration decisions)."""
    ilqr, models, pt, _ = mods
    N = 100
    Q, R = np.eye(12), np.eye(4)
    Qf = 10 * Q
    cost = models.QuadraticCost(Q, R, Qf)
    model = models.QuadcopterEuler(0.1)
    rng = np.random.default_rng(2)
    x0 = np.zeros((3, 12))
    x0[0, 9:12] = [10, 10, 10]
    x0[1:, 9:12] = rng.uniform(-10, 10, (2, 3))
    uGuess = np.tile(models.QuadcopterEuler.uTrim, (3, N, 1))
    traj, L, J, converged = ilqr.iterativeLqr(model, cost, cost, x0, uGuess)
    assert traj.xTraj.shape == (3, N + 1, 12) and L.shape == (3, N, 4, 12) and J.shape == (3,) and converged.dtype == bool
    step = zo.quad_euler_step(0.1)
    for i in range(3):
        rt, rL, rJ, rc = zo.iterativeLqr(step, Q, R, Qf, x0[i], uGuess[i])
        assert bool(converged[i]) == rc
        assert J[i] == pytest.approx(rJ, rel=1e-7)
        assert _rel(traj.xTraj[i], rt.xTraj) <= 1e-6 and _rel(traj.uTraj[i], rt.uTraj) <= 1e-6
        assert _rel(L[i], rL) <= 1e-6
    assert np.all(converged)


def test_iterativeLqr_maxiter_and_batch_independence(mods):
    ilqr, models, _, _ = mods
    N = 30
    cost = models.QuadraticCost(np.eye(12), 0.5 * np.eye(4), 10 * np.eye(12))
    model = models.QuadcopterEuler(0.1)
    rng = np.random.default_rng(3)
    x0 = np.zeros((6, 12))
    x0[:, 9:12] = rng.uniform(-5, 5, (6, 3))
    ug = np.tile(models.QuadcopterEuler.uTrim, (6, N, 1))
    t1, L1, J1, c1 = ilqr.iterativeLqr(model, cost, cost, x0, ug, maxIter=1)
    assert not c1.any()                                   # one iteration can never report convergence from the guess
    tf, Lf, Jf, cf = ilqr.iterativeLqr(model, cost, cost, x0, ug)
    assert np.all(Jf <= J1 + 1e-9) and cf.all()
    # solving a sub-batch alone gives the same answers: trajectories are independent
    ts, Ls, Js, cs = ilqr.iterativeLqr(model, cost, cost, x0[2:4], ug[2:4])
    assert np.array_equal(ts.xTraj, tf.xTraj[2:4]) and np.array_equal(Ls, Lf[2:4]) and np.array_equal(Js, Jf[2:4])


def test_pytree_expansion_constructors():
    """pytrees.py:72-81, 100-115, 139-153, 180-194: from_function / from_trajectory / fromTerminalCostFunction on
    registered models against the oracle's complex-step / autograd expansions."""
    from zopt_amd import models, pytrees
    rng = np.random.default_rng(11)
    model = models.QuadcopterEuler(0.1)
    N = 5
    xT = 0.3 * rng.standard_normal((3, N + 1, 12))
    uT = models.QuadcopterEuler.uTrim + 0.3 * rng.standard_normal((3, N, 4))
    dyn = pytrees.AffineDynamics.from_trajectory(model, pytrees.Trajectory(xT, uT))
    assert dyn.f.shape == (3, N, 12) and dyn.f_x.shape == (3, N, 12, 12) and dyn.f_u.shape == (3, N, 12, 4)
    for b in range(3):
        for k in range(N):
            f, fx, fu = zo.jacobians(zo.quad_euler_step(0.1), xT[b, k], uT[b, k])
            assert np.max(np.abs(dyn.f[b, k] - f)) <= 1e-13
            assert np.max(np.abs(dyn.f_x[b, k] - fx)) <= 1e-12 and np.max(np.abs(dyn.f_u[b, k] - fu)) <= 1e-12
    one = pytrees.AffineDynamics.from_function(model, xT[0, 0], uT[0, 0])
    assert one.f.shape == (12,) and one.f_x.shape == (12, 12) and one.f_u.shape == (12, 4)
    assert np.array_equal(one.f_x, dyn.f_x[0, 0])
    qd = pytrees.QuadraticDynamics.from_trajectory(model, pytrees.Trajectory(xT[0], uT[0]))
    ref = zo.quadratic_dynamics_from_trajectory(zo.quad_euler_step_torch(0.1), zo.Trajectory(xT[0], uT[0]))
    for a, b_ in zip(qd, ref):
        assert a.shape == np.asarray(b_).shape and np.max(np.abs(a - np.asarray(b_))) <= 1e-10
    Q, R, Qf = rng.standard_normal((12, 12)), rng.standard_normal((4, 4)), rng.standard_normal((12, 12))
    cost = models.QuadraticCost(Q, R, Qf)
    qc = pytrees.QuadraticCostFunction.from_trajectory(cost, pytrees.Trajectory(xT[0], uT[0]))
    for k in range(N):
        x, u = xT[0, k], uT[0, k]
        assert abs(qc.c[k] - (x @ Q @ x + u @ R @ u)) <= 1e-12
        assert np.max(np.abs(qc.c_x[k] - (Q + Q.T) @ x)) <= 1e-12 and np.max(np.abs(qc.c_u[k] - (R + R.T) @ u)) <= 1e-12
        assert np.array_equal(qc.c_xx[k], Q + Q.T) and np.array_equal(qc.c_uu[k], R + R.T) and not qc.c_ux[k].any()
    vf = pytrees.QuadraticValueFunction.fromTerminalCostFunction(cost, xT[0, -1])
    xf = xT[0, -1]
    assert vf.v.shape == () and abs(vf.v - xf @ Qf @ xf) <= 1e-12
    assert np.max(np.abs(vf.v_x - (Qf + Qf.T) @ xf)) <= 1e-12 and np.array_equal(vf.v_xx, Qf + Qf.T)
    # a callable goes down the generic path (torch.func on the GPU, tests/test_generic_gpu.py); a non-callable is refused
    ad = pytrees.AffineDynamics.from_function(lambda x, u: 2.0 * x, xT[0, 0], uT[0, 0])
    assert np.array_equal(ad.f_x, 2.0 * np.eye(12)) and not ad.f_u.any()
    with pytest.raises(TypeError):
        pytrees.AffineDynamics.from_function("not a model", xT[0, 0], uT[0, 0])


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4, 5, 6, 7])
def test_ensurePositiveDefinite_adversarial_spectra(mods, kind):
    """K5 computes V max(w, eps) V^T without an eigen-decomposition (matrix-sign iterations on the MFMA tile, ns16.h; NumPy model
    in tools/ns_psd_model.py): the spectra that stress it -- many decades of magnitude, eigenvalues hugging eps from both
    sides, rank one, dominant pairs, exactly zero rows/columns (the structure of the DDP Hessians of a model that is affine in
    some variables), everything already PD -- against eigh (reference ilqrUtils.py:217-225).  Tolerance 2e-11 relative to the
    largest entry of the result (the iteration resolves eigenvalues of a - eps I down to ~1e-12 of its norm)."""
    ilqr = mods[0]
    rng = np.random.default_rng(100 + kind)
    mats = []
    for t in range(24):
        k = int(rng.integers(2, 17))
        Q, _ = np.linalg.qr(rng.standard_normal((k, k)))
        if kind == 0:
            lam = rng.standard_normal(k) * 10 ** rng.uniform(-3, 3)
        elif kind == 1:
            lam = np.concatenate([rng.standard_normal(k // 2), 1e-3 + rng.standard_normal(k - k // 2) * 1e-9])
        elif kind == 2:
            lam = 10.0 ** rng.uniform(-14, 2, k) * rng.choice([-1, 1], k)
        elif kind == 3:
            lam = np.zeros(k)
            lam[0] = rng.standard_normal()
        elif kind == 4:
            lam = rng.standard_normal(k)
            lam[:2] = 1e3
        elif kind == 5:
            lam = 1e-3 + 10.0 ** rng.uniform(-16, -2, k) * rng.choice([-1, 1], k)
        elif kind == 6:      # exactly zero rows / columns around a dense block
            lam = rng.standard_normal(k) * 10 ** rng.uniform(0, 2)
        else:                # already positive definite, wide range
            lam = 10.0 ** rng.uniform(-2, 4, k)
        a = (Q * lam) @ Q.T
        a = 0.5 * (a + a.T)
        if kind == 6:
            dead = rng.choice(k, size=max(1, k // 3), replace=False)
            a[dead, :] = 0.0
            a[:, dead] = 0.0
        A = np.zeros((16, 16))
        A[:k, :k] = a
        mats.append((k, a))
    for k in sorted({k for k, _ in mats}):
        batch = np.stack([a for kk, a in mats if kk == k])
        out = ilqr.ensurePositiveDefinite(batch)
        ref = zo.ensurePositiveDefinite(batch)
        for o, r in zip(out, ref):
            assert np.max(np.abs(o - r)) <= 2e-11 * max(np.max(np.abs(r)), 1e-3)
            assert np.min(np.linalg.eigvalsh(0.5 * (o + o.T))) >= 1e-3 * (1 - 1e-6)


def test_ensurePositiveDefinite_nonfinite_and_zero(mods):
    ilqr = mods[0]
    A = np.zeros((3, 5, 5))
    A[1] = np.nan
    A[2, 0, 0] = np.inf
    out = ilqr.ensurePositiveDefinite(A)
    assert out[0] == pytest.approx(1e-3 * np.eye(5), abs=0)        # the zero matrix: eps I exactly
    assert np.all(np.isnan(out[1]))                                  # eigh of NaN is NaN
    assert not np.all(np.isfinite(out[2]))


def test_iterativeLqr_baseline_config4_full_size(mods):
    """BASELINE configs[3] at full size: 8192 quadcopter iLQR problems, T = 100, x0[9:12] ~ U(-10, 10)^3, uGuess = uTrim
    (tools/bench_ilqr.py's workload).  Size-independent properties on the whole batch, the oracle loop on a few trajectories,
    and batch-composition independence (a trajectory solved alone gives the same answer as inside the 8192)."""
    ilqr, models, pt, _ = mods
    batch, N = 8192, 100
    Q, R, Qf = np.eye(12), np.eye(4), 10 * np.eye(12)
    cost = models.QuadraticCost(Q, R, Qf)
    model = models.QuadcopterEuler(0.1)
    rng = np.random.default_rng(2)
    x0 = np.zeros((batch, 12))
    x0[:, 9:12] = rng.uniform(-10, 10, (batch, 3))
    ug = np.tile(models.QuadcopterEuler.uTrim, (batch, N, 1))
    traj, L, J, conv = ilqr.iterativeLqr(model, cost, cost, x0, ug)
    assert traj.xTraj.shape == (batch, N + 1, 12) and L.shape == (batch, N, 4, 12) and J.shape == (batch,)
    assert conv.mean() > 0.85                                   # most starts converge within 100 iterations
    # the initial guess hovers at x0: J0 = N (x0'Q x0 + uTrim'R uTrim) + x0'Qf x0; every converged solve improved on it
    uT = models.QuadcopterEuler.uTrim
    J0 = (N + 10) * np.sum(x0 ** 2, axis=1) + N * float(uT @ R @ uT)
    assert np.all(J[conv] < J0[conv])
    assert np.all(np.isfinite(traj.xTraj[conv])) and np.all(np.isfinite(L[conv]))
    assert np.all(traj.xTraj[:, 0] == x0)                        # rollouts start at x0 exactly
    # the returned trajectory is the rollout of its own controls and J is its cost
    step = zo.quad_euler_step(0.1)
    pick = [0, 1234, 4095, 8191, int(np.flatnonzero(~conv)[0])] if (~conv).any() else [0, 1234, 4095, 8191]
    for i in pick[:4]:
        x = x0[i].copy()
        Ji = 0.0
        for k in range(N):
            Ji += x @ Q @ x + traj.uTraj[i, k] @ R @ traj.uTraj[i, k]
            x = step(x, traj.uTraj[i, k])
            assert np.max(np.abs(x - traj.xTraj[i, k + 1])) <= 1e-9 * max(1.0, np.max(np.abs(x)))
        Ji += x @ Qf @ x
        assert J[i] == pytest.approx(Ji, rel=1e-10)
    # oracle loop (iteration-by-iteration the same decisions) on two converged trajectories
    for i in pick[:2]:
        rt, rL, rJ, rc = zo.iterativeLqr(step, Q, R, Qf, x0[i], ug[i])
        assert bool(conv[i]) == rc and J[i] == pytest.approx(rJ, rel=1e-7)
        assert _rel(traj.uTraj[i], rt.uTraj) <= 1e-6 and _rel(L[i], rL) <= 1e-6
    # batch-composition independence, including a start that does not converge
    sub = np.array(pick)
    ts, Ls, Js, cs = ilqr.iterativeLqr(model, cost, cost, x0[sub], ug[sub])
    assert np.array_equal(cs, conv[sub])
    assert np.array_equal(Js, J[sub], equal_nan=True)
    assert np.array_equal(ts.uTraj, traj.uTraj[sub], equal_nan=True) and np.array_equal(Ls, L[sub], equal_nan=True)


def _agreement(J_gpu, a_gpu, trace, rtol=1e-9):
    """Leading iterations in which the kernel's cost after the acceptance step and its winning step-size index agree with the
    oracle loop's (cost to `rtol`, NaN == NaN; index exactly)."""
    k = 0
    for (Jo, idx, _), Jg, ag in zip(trace, J_gpu, a_gpu):
        same_J = (np.isnan(Jo) and np.isnan(Jg)) or (Jo == Jg) or (np.isfinite(Jo) and np.isfinite(Jg) and
                                                                   abs(Jo - Jg) <= rtol * abs(Jo))
        if not (same_J and int(ag) == idx):
            break
        k += 1
    return k


def test_iterativeLqr_non_converging_starts_follow_the_oracle_loop(mods):
    """The ~9 % of BASELINE configs[3]'s 8192 starts that never converge (the reference has no regularisation and NaN wins
    forwardPass2's argmin, ilqrUtils.py:140-150, 301-324): one start whose cost ends non-finite and one that runs to maxIter with a
    finite cost are solved by `oracle.iterativeLqr` too.  Same `converged` (False), same finiteness of the final J, and -- through
    zm_ilqr_solve_trace_f64's per-iteration record -- the same cost (1e-9) and the same winning step-size index iteration by
    iteration for as long as the two chaotic iterations stay together (>= 10 required; the count is printed).  A start that went
    NaN on the GPU only, or oscillated on the GPU only, would be a divergence of the HIP path: this is the test that sees it."""
    import warnings
    ilqr, models, pt, _ = mods
    batch, N = 8192, 100
    Q, R, Qf = np.eye(12), np.eye(4), 10 * np.eye(12)
    cost = models.QuadraticCost(Q, R, Qf)
    model = models.QuadcopterEuler(0.1)
    rng = np.random.default_rng(2)
    x0 = np.zeros((batch, 12))
    x0[:, 9:12] = rng.uniform(-10, 10, (batch, 3))
    ug = np.tile(models.QuadcopterEuler.uTrim, (batch, N, 1))
    ilqr._TRACE = []
    try:
        traj, L, J, conv = ilqr.iterativeLqr(model, cost, cost, x0, ug)
        rec = dict(ilqr._TRACE)
    finally:
        ilqr._TRACE = None
    Jtr, atr = rec["J_trace"], rec["alpha_trace"]
    assert Jtr.shape == (rec["iterations"], batch) and atr.shape == Jtr.shape
    assert np.array_equal(Jtr[-1], J, equal_nan=True)                  # the record's last row is the returned cost
    nonfinite = np.flatnonzero(~np.isfinite(J))
    capped = np.flatnonzero(np.isfinite(J) & ~conv)
    assert nonfinite.size > 0 and capped.size > 0, "the workload is expected to hold both kinds of non-converging starts"
    assert not conv[nonfinite].any()
    step = zo.quad_euler_step(0.1)
    report = {}
    for kind, i in (("J non-finite", int(nonfinite[0])), ("maxIter, J finite", int(capped[0]))):
        tr = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                            # overflow / invalid: the reference's NaN path, on purpose
            rt, rL, rJ, rc, its = zo.iterativeLqr(step, Q, R, Qf, x0[i], ug[i], return_iters=True, trace=tr)
        agree = _agreement(Jtr[:, i], atr[:, i], tr)
        report[kind] = (i, agree, its, float(J[i]), float(rJ))
        assert rc is False and not conv[i], (kind, i)
        assert its == N                                                # the oracle loop also runs to maxIter
        assert np.isfinite(rJ) == np.isfinite(J[i]), (kind, i, rJ, J[i])
        assert agree >= 10, (kind, i, agree)
        if np.isfinite(rJ) and agree == its:                           # together to the end: the answers must then be the same
            assert J[i] == pytest.approx(rJ, rel=1e-7) and _rel(traj.uTraj[i], rt.uTraj) <= 1e-5
    print("iLQR non-converging starts vs oracle loop (index, iterations in agreement, oracle iterations, J kernel, J oracle):", report)


def test_expansions_over_an_id_list_match_the_full_calls(mods):
    """zm_linearize_dynamics_list_f64 / zm_quadratize_cost_list_f64 / zm_quadratic_dynamics_list_f64: the listed trajectories get
    bit for bit what the plain entry points write, every other trajectory is left untouched, listed-but-inactive ones are skipped,
    an empty list is a no-op."""
    import ctypes
    import torch
    _, models, _, _lib = mods
    lib = _lib.lib()
    rng = np.random.default_rng(12)
    B, N, n, m = 11, 7, 12, 4
    xT = torch.as_tensor(0.4 * rng.standard_normal((B, N + 1, n)), device="cuda")
    uT = torch.as_tensor(np.array([9.807, 0, 0, 0]) + 0.5 * rng.standard_normal((B, N, m)), device="cuda")
    md = models.QuadcopterEuler(0.1).c_struct()
    cs = models.QuadraticCost(np.eye(n) + 0.1 * np.ones((n, n)), np.eye(m), 10 * np.eye(n)).c_struct()
    pm, pc = ctypes.addressof(md), ctypes.addressof(cs)
    ids = torch.tensor([9, 2, 5, 0], dtype=torch.int32, device="cuda")
    act = torch.ones(B, dtype=torch.int32, device="cuda")
    act[5] = 0
    mk = lambda *s: torch.full(s, 7.0, dtype=torch.float64, device="cuda")
    shapes = {"f_x": (B, N, n, n), "f_u": (B, N, n, m), "c_x": (B, N, n), "c_u": (B, N, m), "v_x": (B, n), "c": (B, N),
              "f_xx": (B, N, n, n, n), "f_ux": (B, N, n, m, n), "f_uu": (B, N, n, m, m)}
    full = {k: mk(*s) for k, s in shapes.items()}
    part = {k: mk(*s) for k, s in shapes.items()}
    p = lambda t: t.data_ptr()
    assert lib.zm_linearize_dynamics_f64(pm, p(xT), p(uT), None, None, p(full["f_x"]), p(full["f_u"]), B, N, None) == 0
    assert lib.zm_quadratize_cost_f64(pc, n, m, p(xT), p(uT), None, p(full["c"]), p(full["c_x"]), p(full["c_u"]), None, p(full["v_x"]),
                                      None, None, None, None, B, N, None) == 0
    assert lib.zm_quadratic_dynamics_f64(pm, p(xT), p(uT), None, p(full["f_xx"]), p(full["f_ux"]), p(full["f_uu"]), B, N, None) == 0
    for cnt in (0, int(ids.numel())):
        assert lib.zm_linearize_dynamics_list_f64(pm, p(xT), p(uT), p(ids), cnt, p(act), None, p(part["f_x"]), p(part["f_u"]),
                                                  B, N, None) == 0
        assert lib.zm_quadratize_cost_list_f64(pc, n, m, p(xT), p(uT), p(ids), cnt, p(act), p(part["c"]), p(part["c_x"]),
                                               p(part["c_u"]), None, p(part["v_x"]), None, None, None, None, B, N, None) == 0
        assert lib.zm_quadratic_dynamics_list_f64(pm, p(xT), p(uT), p(ids), cnt, p(act), p(part["f_xx"]), p(part["f_ux"]),
                                                  p(part["f_uu"]), B, N, None) == 0
        torch.cuda.synchronize()
        done = [9, 2, 0] if cnt else []
        for k in shapes:
            for b in range(B):
                if b in done:
                    assert torch.equal(part[k][b], full[k][b]), (k, b)
                else:
                    assert bool((part[k][b] == 7.0).all()), (k, b)
    assert lib.zm_linearize_dynamics_list_f64(pm, p(xT), p(uT), p(ids), B + 1, None, None, p(part["f_x"]), p(part["f_u"]),
                                              B, N, None) == _lib.ZM_EINVAL


def test_quadratize_cost_diagonal_shortcut_is_exact(mods):
    """With `zm_quadcost_t.diagonal = 1` the gradients c_x = (Q + Q^T) x, c_u, v_x skip the vanishing off-diagonal terms: bit for
    bit the general formula (finite states), and equal to the oracle."""
    import ctypes
    import torch
    _, models, _, _lib = mods
    rng = np.random.default_rng(21)
    b, N, n, m = 3, 5, 12, 4
    Q, R, Qf = np.diag(rng.uniform(0.5, 2, n)), np.diag(rng.uniform(0.5, 2, m)), np.diag(rng.uniform(5, 20, n))
    xT, uT = rng.standard_normal((b, N + 1, n)), rng.standard_normal((b, N, m))
    dx, du = torch.as_tensor(xT, device="cuda"), torch.as_tensor(uT, device="cuda")
    outs = []
    cost = models.QuadraticCost(Q, R, Qf)       # keeps the device copies of Q, R, Qf alive: c_struct() only holds their addresses
    for hint in (1, 0):
        cs = cost.c_struct()
        assert cs.diagonal == 1
        cs.diagonal = hint
        t = lambda *s: torch.empty(s, dtype=torch.float64, device="cuda")
        c, c_x, c_u, v, v_x = t(b, N), t(b, N, n), t(b, N, m), t(b), t(b, n)
        rc = _lib.lib().zm_quadratize_cost_f64(ctypes.addressof(cs), n, m, dx.data_ptr(), du.data_ptr(), None, c.data_ptr(),
                                               c_x.data_ptr(), c_u.data_ptr(), v.data_ptr(), v_x.data_ptr(), None, None, None,
                                               None, b, N, None)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append([a.cpu().numpy() for a in (c, c_x, c_u, v, v_x)])
    for a, bb in zip(*outs):
        assert np.array_equal(a, bb)
    ref = zo.quadratic_cost_from_trajectory(Q, R, zo.Trajectory(xT[0], uT[0]))
    assert _rel(outs[0][1][0], ref.c_x) <= 1e-13 and _rel(outs[0][2][0], ref.c_u) <= 1e-13
