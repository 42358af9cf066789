"""GPU parity tests for K4 ddp_backward, the second-order lineariser and the DDP driver
(reference ilqrUtils.py:184-214, 237-251, 330-397; pytrees.py:180-194)."""
import json
import os

import numpy as np
import pytest

from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import _lib, ilqrUtils, models, pytrees
    return ilqrUtils, models, pytrees, _lib


def test_kat_riccati_step_ddp(mods):
    """reference tests/test_ilqrUtils.py:110-135 as a 1-step DDP backward pass with zero second derivatives:
    v_xx = 1.5 I, L = -0.5 I at rel 1e-3 (the PD clamp adds 1e-3 I to Q_xx, Q_uu)."""
    ilqr, _, pt, _ = mods
    I2 = np.eye(2)
    z = np.zeros((1, 2, 2, 2))
    dyn = pt.QuadraticDynamics(np.zeros((1, 2)), I2[None], I2[None], z, z, z)
    cost = pt.QuadraticCostFunction(np.zeros(1), np.zeros((1, 2)), np.zeros((1, 2)), I2[None], np.zeros((1, 2, 2)), I2[None])
    pol = ilqr.backwardPass_ddp(dyn, cost, pt.QuadraticValueFunction(0.0, np.zeros(2), I2))
    assert pol.l[0] == pytest.approx(np.array(KATS["A4_riccatiStep_ddp"]["l"]))
    assert pol.L[0] == pytest.approx(np.array(KATS["A4_riccatiStep_ddp"]["L"]), rel=1e-3)
    # exact value with the clamp: Q_uu = (1 + 1 + 1e-3) I, Q_ux = I  ->  L = -1/2.001 I
    assert pol.L[0] == pytest.approx(-np.eye(2) / 2.001, rel=1e-12)


@pytest.mark.parametrize("n,m,T,batch", [(12, 4, 20, 4), (2, 2, 3, 3), (4, 1, 7, 2), (8, 4, 5, 2), (5, 3, 4, 2), (12, 4, 1, 2)])
def test_backwardPass_ddp_parity(mods, n, m, T, batch):
    ilqr = mods[0]
    dyn, cost, Vf = problems.random_ilqr_model(batch, T, n, m, seed=7 * n + m + T)
    rng = np.random.default_rng(n + m)
    sym = lambda X: 0.5 * (X + np.swapaxes(X, -1, -2))
    f_xx = 0.2 * sym(rng.standard_normal((batch, T, n, n, n)))
    f_ux = 0.2 * rng.standard_normal((batch, T, n, m, n))
    f_uu = 0.2 * sym(rng.standard_normal((batch, T, n, m, m)))
    pol = ilqr.backwardPass_ddp(dyn + (f_xx, f_ux, f_uu), cost, Vf)
    ref = zo.backwardPass_ddp(zo.QuadraticDynamics(*dyn, f_xx, f_ux, f_uu), zo.QuadraticCostFunction(*cost),
                              zo.QuadraticValueFunction(*Vf))
    assert _rel(pol.L, ref.L) <= 1e-9 and _rel(pol.l, ref.l) <= 1e-9


def test_quadratic_dynamics_quadcopter(mods):
    """Hyper-dual second derivatives of the device model vs torch autograd of the oracle's restatement (pytrees.py:180-194)."""
    import ctypes
    import torch
    _, models, _, _lib = mods
    rng = np.random.default_rng(9)
    b, N = 2, 5
    xT = 0.4 * rng.standard_normal((b, N + 1, 12))
    uT = np.array([9.807, 0, 0, 0]) + 0.5 * rng.standard_normal((b, N, 4))
    dx, du = torch.as_tensor(xT, device="cuda"), torch.as_tensor(uT, device="cuda")
    t = lambda *s: torch.full(s, float("nan"), dtype=torch.float64, device="cuda")
    f_xx, f_ux, f_uu = t(b, N, 12, 12, 12), t(b, N, 12, 4, 12), t(b, N, 12, 4, 4)
    md = models.QuadcopterEuler(0.1).c_struct()
    rc = _lib.lib().zm_quadratic_dynamics_f64(ctypes.addressof(md), dx.data_ptr(), du.data_ptr(), None, f_xx.data_ptr(),
                                              f_ux.data_ptr(), f_uu.data_ptr(), b, N, None)
    assert rc == 0
    torch.cuda.synchronize()
    ft = zo.quad_euler_step_torch(0.1)
    for i in range(b):
        ref = zo.quadratic_dynamics_from_trajectory(ft, zo.Trajectory(xT[i], uT[i]))
        assert np.max(np.abs(f_xx[i].cpu().numpy() - ref.f_xx)) <= 1e-12
        assert np.max(np.abs(f_ux[i].cpu().numpy() - ref.f_ux)) <= 1e-12
        assert np.max(np.abs(f_uu[i].cpu().numpy() - ref.f_uu)) <= 1e-12
    assert np.max(np.abs(f_xx.cpu().numpy())) > 1e-3      # the model really has curvature


def test_kat_ddp_driver(mods):
    """reference tests/test_ilqrUtils.py:184-196: A=B=Q=R=I, N=3, x0=(2,1) -> converged (LQ: second derivatives vanish,
    only the 1e-3 clamp of the zero vf_zz differs from iLQR)."""
    ilqr, models, _, _ = mods
    I = np.eye(2)
    cost = models.QuadraticCost(I, I, I)
    x0 = np.array([2.0, 1.0])
    traj, L, J, converged = ilqr.differentialDynamicProgramming(models.LinearModel(I, I), cost.runningCost, cost.terminalCost,
                                                               x0, np.zeros((3, 2)))
    assert converged is True
    import torch
    ft = lambda x, u: x + u
    rt, rL, rJ, rc = zo.differentialDynamicProgramming(lambda x, u: x + u, ft, I, I, I, x0, np.zeros((3, 2)))
    assert rc and _rel(traj.uTraj, rt.uTraj) <= 1e-9 and _rel(L, rL) <= 1e-9 and J == pytest.approx(rJ, rel=1e-10)


def test_ddp_quadcopter_demo_problem(mods):
    """demos/differentialDynamicProgramming.py:22-39: quadcopter, N=100 (here 40 to bound the oracle's CPU time), Q=I,
    R=0.2 I, terminal 10 x'Qx, x0[9:12]=(0,5,0), uGuess=uTrim, against the CPU oracle loop."""
    ilqr, models, _, _ = mods
    N = 40
    Q, R = np.eye(12), 0.2 * np.eye(4)
    Qf = 10 * Q
    cost = models.QuadraticCost(Q, R, Qf)
    x0 = np.zeros((2, 12))
    x0[0, 9:12] = [0, 5, 0]
    x0[1, 9:12] = [1, -2, 3]
    ug = np.tile(models.QuadcopterEuler.uTrim, (2, N, 1))
    traj, L, J, converged = ilqr.differentialDynamicProgramming(models.QuadcopterEuler(0.1), cost, cost, x0, ug)
    fn, ft = zo.quad_euler_step(0.1), zo.quad_euler_step_torch(0.1)
    for i in range(2):
        rt, rL, rJ, rc = zo.differentialDynamicProgramming(fn, ft, Q, R, Qf, x0[i], ug[i])
        assert bool(converged[i]) == rc
        assert J[i] == pytest.approx(rJ, rel=1e-7)
        assert _rel(traj.xTraj[i], rt.xTraj) <= 1e-6 and _rel(traj.uTraj[i], rt.uTraj) <= 1e-6 and _rel(L[i], rL) <= 1e-5


def test_ddp_baseline_config4_full_size(mods):
    """BASELINE configs[3], DDP leg, at its stated size: 8192 quadcopter problems, T = 100, R = 0.2 I
    (demos/differentialDynamicProgramming.py:22-39; tools/secondary_bench.config3_ilqr(ddp=True) is the same workload).
    Size-independent properties on the whole batch, the oracle loop (ilqrUtils.py:360-397 restated) on picked trajectories,
    and batch-composition independence.  Twin of tests/test_ilqr_solve_gpu.py::test_iterativeLqr_baseline_config4_full_size."""
    ilqr, models, _, _ = mods
    batch, N = 8192, 100
    Q, R, Qf = np.eye(12), 0.2 * np.eye(4), 10 * np.eye(12)
    cost = models.QuadraticCost(Q, R, Qf)
    model = models.QuadcopterEuler(0.1)
    rng = np.random.default_rng(2)
    x0 = np.zeros((batch, 12))
    x0[:, 9:12] = rng.uniform(-10, 10, (batch, 3))
    ug = np.tile(models.QuadcopterEuler.uTrim, (batch, N, 1))
    traj, L, J, conv = ilqr.differentialDynamicProgramming(model, cost, cost, x0, ug)
    assert traj.xTraj.shape == (batch, N + 1, 12) and traj.uTraj.shape == (batch, N, 4)
    assert L.shape == (batch, N, 4, 12) and J.shape == (batch,) and conv.shape == (batch,) and conv.dtype == np.bool_
    assert conv.mean() > 0.95                                    # nearly every start converges within 100 iterations
    # the initial guess hovers at x0: J0 = N (x0'Q x0 + uTrim'R uTrim) + x0'Qf x0; the argmin over 16 step sizes (down to
    # 0.5^15) never accepts a worse trajectory than the previous one, so J is non-increasing from J0 on every finite solve
    uT = models.QuadcopterEuler.uTrim
    J0 = (N + 10) * np.sum(x0 ** 2, axis=1) + N * float(uT @ R @ uT)
    fin = np.isfinite(J)
    assert fin.mean() > 0.99 and np.all(J[fin] <= J0[fin] * (1 + 1e-12))
    assert np.all(J[conv] < J0[conv]) and np.all(fin[conv])       # `converged` is never set on a NaN cost (NaN compares false)
    assert np.all(np.isfinite(traj.xTraj[conv])) and np.all(np.isfinite(L[conv]))
    assert np.all(traj.xTraj[:, 0] == x0)                         # rollouts start at x0 exactly
    # dynamics consistency: the returned trajectory is the rollout of its own controls, and J is its cost
    step = zo.quad_euler_step(0.1)
    nonconv = np.flatnonzero(~conv & fin)
    pick = [0, 1234, 4095, 8191] + ([int(nonconv[0])] if nonconv.size else [])
    for i in pick:
        x = x0[i].copy()
        Ji = 0.0
        for k in range(N):
            Ji += x @ Q @ x + traj.uTraj[i, k] @ R @ traj.uTraj[i, k]
            x = step(x, traj.uTraj[i, k])
            assert np.max(np.abs(x - traj.xTraj[i, k + 1])) <= 1e-9 * max(1.0, np.max(np.abs(x)))
        Ji += x @ Qf @ x
        assert J[i] == pytest.approx(Ji, rel=1e-10)
    # oracle loop on two picked trajectories: same `converged`, same cost, same controls and gains.  Tolerances 1e-5 (trajectories) /
    # 1e-4 (gains) are those of a WHOLE SOLVE, not of a sweep: the kernel's matrix-sign PD projection and the oracle's `eigh` agree to
    # 1e-12 per matrix and the sweeps to 1e-9 (tests above, tests/test_reference_fixtures_gpu.py), but a DDP solve feeds each
    # iteration's rounding differences through the next linearisation, projection and 16-way argmin for up to 100 iterations, and the
    # gains L = -Q_uu^-1 Q_ux carry the conditioning of Q_uu on top (measured: 1e-9 .. 1e-7 on these problems).
    ft = zo.quad_euler_step_torch(0.1)
    for i in (0, 4095):
        rt, rL, rJ, rc = zo.differentialDynamicProgramming(step, ft, Q, R, Qf, x0[i], ug[i])
        assert bool(conv[i]) == rc and J[i] == pytest.approx(rJ, rel=1e-7)
        if rc:
            assert _rel(traj.uTraj[i], rt.uTraj) <= 1e-5 and _rel(traj.xTraj[i], rt.xTraj) <= 1e-5 and _rel(L[i], rL) <= 1e-4
    # batch-composition independence (incl. a start that does not converge, when there is one)
    sub = np.array(pick)
    ts, Ls, Js, cs = ilqr.differentialDynamicProgramming(model, cost, cost, x0[sub], ug[sub])
    assert np.array_equal(cs, conv[sub])
    assert np.array_equal(Js, J[sub], equal_nan=True)
    assert np.array_equal(ts.uTraj, traj.uTraj[sub], equal_nan=True) and np.array_equal(Ls, L[sub], equal_nan=True)


def test_conditionQuadraticDynamics_matches_oracle():
    """ilqrUtils.py:237-251: contraction with v_x, PD projection of the stacked block, blocks sliced back."""
    from zopt_amd import ilqrUtils, pytrees
    rng = np.random.default_rng(33)
    for n, m, lead in ((12, 4, (5,)), (3, 2, (2, 4)), (7, 1, ())):
        f_xx = rng.standard_normal(lead + (n, n, n))
        f_xx = 0.5 * (f_xx + np.swapaxes(f_xx, -1, -2))
        f_uu = rng.standard_normal(lead + (n, m, m))
        f_uu = 0.5 * (f_uu + np.swapaxes(f_uu, -1, -2))
        f_ux = rng.standard_normal(lead + (n, m, n))
        v_x = rng.standard_normal(lead + (n,))
        z = np.zeros(lead + (n,))
        dyn = pytrees.QuadraticDynamics(z, np.zeros(lead + (n, n)), np.zeros(lead + (n, m)), f_xx, f_ux, f_uu)
        gxx, gux, guu = ilqrUtils.conditionQuadraticDynamics(dyn, v_x)
        rxx, rux, ruu = zo.conditionQuadraticDynamics(zo.QuadraticDynamics(*dyn), v_x)
        assert gxx.shape == lead + (n, n) and gux.shape == lead + (m, n) and guu.shape == lead + (m, m)
        sc = max(np.max(np.abs(rxx)), 1.0)
        assert np.max(np.abs(gxx - rxx)) <= 1e-10 * sc and np.max(np.abs(gux - rux)) <= 1e-10 * sc
        assert np.max(np.abs(guu - ruu)) <= 1e-10 * sc


def test_control_affine_models_skip_the_zero_blocks(mods):
    """`zm_model_nonlinear_mask` declares the variables a registered model is not affine in (quadcopter: the 9 velocity / rate / angle
    states; linear model: none).  The autograd oracle confirms every other second derivative is zero; the second-order expansion and
    the DDP sweep accept NULL for f_ux / f_uu of such a model and give bit-identical results to zero tensors."""
    import ctypes
    import torch
    ilqr, models, _, _lib = mods
    lib = _lib.lib()
    mask = ctypes.c_uint32(123)
    md = models.QuadcopterEuler(0.1).c_struct()
    assert lib.zm_model_nonlinear_mask(ctypes.addressof(md), ctypes.byref(mask)) == 0 and mask.value == 0x1FF
    mdl = models.LinearModel(np.eye(3), np.ones((3, 2))).c_struct()
    assert lib.zm_model_nonlinear_mask(ctypes.addressof(mdl), ctypes.byref(mask)) == 0 and mask.value == 0
    assert lib.zm_model_nonlinear_mask(None, ctypes.byref(mask)) == _lib.ZM_EINVAL
    # oracle: all second derivatives outside the declared 9 x 9 block vanish
    rng = np.random.default_rng(4)
    b, N = 2, 4
    xT = 0.5 * rng.standard_normal((b, N + 1, 12))
    uT = np.array([9.807, 0, 0, 0]) + rng.standard_normal((b, N, 4))
    ref = zo.quadratic_dynamics_from_trajectory(zo.quad_euler_step_torch(0.1), zo.Trajectory(xT[0], uT[0]))
    assert np.all(ref.f_ux == 0) and np.all(ref.f_uu == 0) and np.all(ref.f_xx[:, :, 9:, :] == 0) and np.all(ref.f_xx[:, :, :, 9:] == 0)
    # expansion with and without the zero blocks
    dx, du = torch.as_tensor(xT, device="cuda"), torch.as_tensor(uT, device="cuda")
    t = lambda *s: torch.full(s, float("nan"), dtype=torch.float64, device="cuda")
    f_xx, f_ux, f_uu, g_xx = t(b, N, 12, 12, 12), t(b, N, 12, 4, 12), t(b, N, 12, 4, 4), t(b, N, 12, 12, 12)
    pm = ctypes.addressof(md)
    assert lib.zm_quadratic_dynamics_f64(pm, dx.data_ptr(), du.data_ptr(), None, f_xx.data_ptr(), f_ux.data_ptr(), f_uu.data_ptr(),
                                         b, N, None) == 0
    assert lib.zm_quadratic_dynamics_f64(pm, dx.data_ptr(), du.data_ptr(), None, g_xx.data_ptr(), None, None, b, N, None) == 0
    assert lib.zm_quadratic_dynamics_f64(pm, dx.data_ptr(), du.data_ptr(), None, g_xx.data_ptr(), f_ux.data_ptr(), None,
                                         b, N, None) == _lib.ZM_EINVAL          # f_ux and f_uu go together
    torch.cuda.synchronize()
    assert torch.equal(f_xx, g_xx) and not bool(f_ux.any()) and not bool(f_uu.any())
    # sweep with zero tensors vs NULL
    dyn, cost, Vf = problems.random_ilqr_model(b, N, 12, 4, seed=3)
    dev = [torch.as_tensor(np.ascontiguousarray(X), device="cuda") for X in (dyn[1], dyn[2], cost[1], cost[2], cost[3], cost[4], cost[5],
                                                                             Vf[1], Vf[2])]
    outs = []
    for ux, uu in ((f_ux, f_uu), (None, None)):
        l, L = t(b, N, 4), t(b, N, 4, 12)
        rc = lib.zm_ddp_backward_f64(dev[0].data_ptr(), dev[1].data_ptr(), f_xx.data_ptr(), ux.data_ptr() if ux is not None else None,
                                     uu.data_ptr() if uu is not None else None, *[d.data_ptr() for d in dev[2:]], None, 0,
                                     l.data_ptr(), L.data_ptr(), b, N, 12, 4, None)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append((l.cpu().numpy(), L.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    refp = zo.backwardPass_ddp(zo.QuadraticDynamics(dyn[0][0], dyn[1][0], dyn[2][0], f_xx[0].cpu().numpy(), f_ux[0].cpu().numpy(),
                                                    f_uu[0].cpu().numpy()),
                               zo.QuadraticCostFunction(*[c[0] for c in cost]), zo.QuadraticValueFunction(*[v[0] for v in Vf]))
    assert _rel(outs[1][1][0], refp.L) <= 1e-9 and _rel(outs[1][0][0], refp.l) <= 1e-9


@pytest.mark.parametrize("wind", [(0.0, 0.0, 0.0), (3.0, 1.0, -0.5)])
def test_declared_hessian_pairs_cover_every_second_derivative(mods, wind):
    """The quadcopter declares the 28 variable pairs that can have a second derivative (model_hessian_pairs, models.h); the kernel
    evaluates only those, two trajectory points per wave.  Independent check, also with a constant wind (no autograd restatement
    exists for it): central differences of the first-derivative kernel in all 16 variables reproduce every entry -- the declared
    ones and the structural zeros."""
    import ctypes
    import torch
    _, models, _, _lib = mods
    lib = _lib.lib()
    rng = np.random.default_rng(17)
    n, m, N = 12, 4, 3
    md = models.QuadcopterEuler(0.1, wind_ned=wind).c_struct()
    pm = ctypes.addressof(md)
    x = 0.6 * rng.standard_normal((1, N + 1, n))
    u = np.array([9.807, 0, 0, 0]) + rng.standard_normal((1, N, m))
    p = lambda t: t.data_ptr()

    def jac(xa, ua):
        dx, du = torch.as_tensor(xa, device="cuda"), torch.as_tensor(ua, device="cuda")
        fx = torch.empty((1, N, n, n), dtype=torch.float64, device="cuda")
        fu = torch.empty((1, N, n, m), dtype=torch.float64, device="cuda")
        assert lib.zm_linearize_dynamics_f64(pm, p(dx), p(du), None, None, p(fx), p(fu), 1, N, None) == 0
        torch.cuda.synchronize()
        return np.concatenate([fx.cpu().numpy(), fu.cpu().numpy()], axis=-1)[0]          # (N, n, n + m)

    dx, du = torch.as_tensor(x, device="cuda"), torch.as_tensor(u, device="cuda")
    t = lambda *s: torch.full(s, float("nan"), dtype=torch.float64, device="cuda")
    f_xx, f_ux, f_uu = t(1, N, n, n, n), t(1, N, n, m, n), t(1, N, n, m, m)
    assert lib.zm_quadratic_dynamics_f64(pm, p(dx), p(du), None, p(f_xx), p(f_ux), p(f_uu), 1, N, None) == 0
    torch.cuda.synchronize()
    H = np.zeros((N, n, n + m, n + m))                         # H[k, i, a, b] = d2 f_i / dz_a dz_b from the kernel
    H[:, :, :n, :n] = f_xx[0].cpu().numpy()
    H[:, :, n:, :n] = f_ux[0].cpu().numpy()
    H[:, :, :n, n:] = np.swapaxes(f_ux[0].cpu().numpy(), -1, -2)
    H[:, :, n:, n:] = f_uu[0].cpu().numpy()
    h = 1e-5
    for j in range(n + m):
        xp, xm_, up, um = x.copy(), x.copy(), u.copy(), u.copy()
        if j < n:
            xp[0, :N, j] += h
            xm_[0, :N, j] -= h
        else:
            up[0, :, j - n] += h
            um[0, :, j - n] -= h
        fd = (jac(xp, up) - jac(xm_, um)) / (2 * h)           # (N, n, n + m): d/dz_j of the Jacobian
        assert np.max(np.abs(fd - H[:, :, :, j])) <= 2e-8 * max(1.0, np.max(np.abs(H)))
    assert np.max(np.abs(H)) > 1e-3
    if any(wind):     # the wind really changes the curvature (aerodynamic force quadratic in the air-relative velocity)
        md0 = models.QuadcopterEuler(0.1).c_struct()
        g_xx = t(1, N, n, n, n)
        assert lib.zm_quadratic_dynamics_f64(ctypes.addressof(md0), p(dx), p(du), None, p(g_xx), None, None, 1, N, None) == 0
        torch.cuda.synchronize()
        assert float((g_xx - f_xx).abs().max()) > 1e-4


def test_packed_second_derivatives_match_the_full_tensors(mods):
    """zm_model_hessian_pairs / zm_quadratic_dynamics_pairs_list_f64 / zm_ddp_backward_pairs_list_f64 (the form the fused DDP
    driver uses): the packed entries (closed-form second derivatives, zopt_amd/csrc/quad_derivs_gen.h) are the nonzero entries of
    f_xx (hyper-dual evaluation of the model; QuadraticDynamics.from_trajectory, pytrees.py:180-194) to rounding, every other entry
    of f_xx is exactly zero, and the sweep over the packed form returns bit for bit the policy of zm_ddp_backward_f64 over the full
    tensors holding the same numbers (ilqrUtils.py:184-214, 237-251)."""
    import ctypes
    import torch
    ilqr, models, pt, _lib = mods
    lib = _lib.lib()
    rng = np.random.default_rng(41)
    b, T, n, m = 5, 9, 12, 4
    model = models.QuadcopterEuler(0.1, wind_ned=(1.0, -2.0, 0.5))
    md = model.c_struct()
    pmd = ctypes.addressof(md)
    npairs = ctypes.c_int32(0)
    pairs = (ctypes.c_int32 * 64)()
    _lib.check(lib.zm_model_hessian_pairs(pmd, ctypes.addressof(pairs), ctypes.addressof(npairs)), "pairs")
    P = npairs.value
    ab = np.array(pairs[:2 * P]).reshape(P, 2)
    assert P == 28 and np.all(ab[:, 0] <= ab[:, 1]) and np.all(ab < 9) and len({tuple(x) for x in ab}) == P
    xT = torch.as_tensor(0.4 * rng.standard_normal((b, T + 1, n)), device="cuda")
    uT = torch.as_tensor(models.QuadcopterEuler.uTrim + rng.standard_normal((b, T, m)), device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    H = torch.full((b, T, P, n), float("nan"), dtype=torch.float64, device="cuda")
    _lib.check(lib.zm_quadratic_dynamics_pairs_list_f64(pmd, xT.data_ptr(), uT.data_ptr(), None, 0, None, H.data_ptr(), b, T, st), "packed")
    f_xx = torch.empty((b, T, n, n, n), dtype=torch.float64, device="cuda")
    _lib.check(lib.zm_quadratic_dynamics_f64(pmd, xT.data_ptr(), uT.data_ptr(), None, f_xx.data_ptr(), None, None, b, T, st), "full")
    Hn, F = H.cpu().numpy(), f_xx.cpu().numpy()
    assert np.all(np.isfinite(Hn))
    rebuilt = np.zeros_like(F)
    for p, (a, bb) in enumerate(ab):
        rebuilt[:, :, :, a, bb] = Hn[:, :, p, :]
        rebuilt[:, :, :, bb, a] = Hn[:, :, p, :]
    declared = rebuilt != 0
    assert np.all(F[~declared & (np.abs(F) > 0)] == 0) and np.count_nonzero(F[~declared]) == 0   # everything else exactly zero
    assert np.max(np.abs(rebuilt - F)) <= 1e-13 * np.max(np.abs(F))     # closed forms vs hyper-dual numbers: rounding only
    f_xx = torch.as_tensor(rebuilt, device="cuda")                      # the full tensors with exactly the packed numbers
    # the sweep: packed operand vs full tensors (f_ux = f_uu = NULL: the quadcopter is affine in its controls)
    (f, f_x, f_u), (c, c_x, c_u, c_xx, c_ux, c_uu), (v, v_x, v_xx) = problems.random_ilqr_model(b, T, n, m, seed=43)
    dev = [torch.as_tensor(np.ascontiguousarray(X), device="cuda") for X in (f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, v_x, v_xx)]
    l1, L1 = torch.empty((b, T, m), dtype=torch.float64, device="cuda"), torch.empty((b, T, m, n), dtype=torch.float64, device="cuda")
    l2, L2 = torch.empty_like(l1), torch.empty_like(L1)
    _lib.check(lib.zm_ddp_backward_f64(dev[0].data_ptr(), dev[1].data_ptr(), f_xx.data_ptr(), None, None, *[t.data_ptr() for t in dev[2:]],
                                       None, 0, l1.data_ptr(), L1.data_ptr(), b, T, n, m, st), "full sweep")
    _lib.check(lib.zm_ddp_backward_pairs_list_f64(pmd, dev[0].data_ptr(), dev[1].data_ptr(), H.data_ptr(), *[t.data_ptr() for t in dev[2:]],
                                                  None, 0, None, 0, l2.data_ptr(), L2.data_ptr(), b, T, st), "packed sweep")
    assert torch.equal(l1, l2) and torch.equal(L1, L2)
    # a model without declared pairs is refused, not mis-read
    lin = models.LinearModel(np.eye(2), np.eye(2)).c_struct()
    assert lib.zm_quadratic_dynamics_pairs_list_f64(ctypes.addressof(lin), xT.data_ptr(), uT.data_ptr(), None, 0, None, H.data_ptr(), 1, 1,
                                                    st) == _lib.ZM_EUNSUPPORTED


def test_ddp_follows_the_oracle_loop_iteration_by_iteration(mods):
    """Twin of tests/test_ilqr_solve_gpu.py::test_iterativeLqr_non_converging_starts_follow_the_oracle_loop for the DDP driver
    (ilqrUtils.py:360-397): zm_ilqr_solve_trace_f64's record of every iteration's accepted cost and winning step-size index against
    `oracle.differentialDynamicProgramming(trace=...)` on the first 1024 of BASELINE configs[3]'s starts -- one start that converges
    and, when the shard holds one, one that runs to maxIter.  The kernel projects with the matrix-sign iteration, the oracle with
    `eigh` (1e-12 apart per matrix), so the two loops are the same computation up to rounding: same `converged`, the same iteration
    count for the converging start, cost (1e-9) and step-size index equal for >= 10 leading iterations (count printed)."""
    import warnings
    from tests.test_ilqr_solve_gpu import _agreement
    ilqr, models, _, _ = mods
    batch, N = 1024, 100
    Q, R, Qf = np.eye(12), 0.2 * np.eye(4), 10 * np.eye(12)
    cost = models.QuadraticCost(Q, R, Qf)
    rng = np.random.default_rng(2)
    x0 = np.zeros((8192, 12))
    x0[:, 9:12] = rng.uniform(-10, 10, (8192, 3))
    x0 = x0[:batch]
    ug = np.tile(models.QuadcopterEuler.uTrim, (batch, N, 1))
    ilqr._TRACE = []
    try:
        traj, L, J, conv = ilqr.differentialDynamicProgramming(models.QuadcopterEuler(0.1), cost, cost, x0, ug)
        rec = dict(ilqr._TRACE)
    finally:
        ilqr._TRACE = None
    Jtr, atr = rec["J_trace"], rec["alpha_trace"]
    assert Jtr.shape == (rec["iterations"], batch) and np.array_equal(Jtr[-1], J, equal_nan=True)
    fn, ft = zo.quad_euler_step(0.1), zo.quad_euler_step_torch(0.1)
    capped = np.flatnonzero(~conv)
    picks = [("converges", int(np.flatnonzero(conv)[0]))] + ([("maxIter", int(capped[0]))] if capped.size else [])
    report = {}
    for kind, i in picks:
        tr = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rt, rL, rJ, rc, its = zo.differentialDynamicProgramming(fn, ft, Q, R, Qf, x0[i], ug[i], return_iters=True, trace=tr)
        agree = _agreement(Jtr[:, i], atr[:, i], tr)
        report[kind] = (i, agree, its, float(J[i]), float(rJ))
        assert rc == bool(conv[i]), (kind, i)
        assert agree >= min(10, its), (kind, i, agree, its)
        if rc:
            assert agree == its and J[i] == pytest.approx(rJ, rel=1e-9) and _rel(traj.uTraj[i], rt.uTraj) <= 1e-5
    print("DDP vs oracle loop (index, iterations in agreement, oracle iterations, J kernel, J oracle):", report)
