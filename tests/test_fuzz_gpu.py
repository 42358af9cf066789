"""Seeded randomised parity sweep over shapes: many small (n, m, T, batch) combinations per entry point against the oracle, to
catch lane-map / padding corner cases the hand-picked parametrisations miss (ragged n and m, single steps, batches that do not
fill a wave, batches larger than the SIMD count)."""
import numpy as np
import pytest

from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import ilqrUtils, lqrUtils, pytrees
    return lqrUtils, ilqrUtils, pytrees


def test_lqr_small_shapes_fp64(mods):
    lqr, _, _ = mods
    rng = np.random.default_rng(2024)
    for case in range(60):
        n, m = int(rng.integers(1, 13)), int(rng.integers(1, 5))
        T, b = int(rng.integers(1, 8)), int(rng.integers(1, 71))
        A, B, Q, R = problems.random_time_varying(b, T, n, m, seed=1000 + case)
        if case % 3 == 0:            # nonsymmetric weights every third case
            Q = Q + 0.1 * rng.standard_normal(Q.shape)
            R = R + 0.1 * rng.standard_normal(R.shape)
        L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
        assert _rel(L, zo.discreteFiniteHorizonLqr(A, B, Q, R, T)) <= 1e-10, (case, n, m, T, b)


def test_lqr_medium_shapes_fp32_and_fp64(mods):
    lqr, _, _ = mods
    rng = np.random.default_rng(2025)
    for case in range(30):
        n, m = int(rng.integers(13, 65)), int(rng.integers(1, 17))
        T, b = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        A, B, Q, R = problems.random_time_varying(b, T, n, m, seed=2000 + case)
        Lr = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
        assert _rel(lqr.discreteFiniteHorizonLqr(A, B, Q, R, T), Lr) <= 1e-10, ("f64", case, n, m, T, b)
        A32, B32, Q32, R32 = (x.astype(np.float32) for x in (A, B, Q, R))
        L64 = zo.discreteFiniteHorizonLqr(*(x.astype(np.float64) for x in (A32, B32, Q32, R32)), T)
        L32 = zo.discreteFiniteHorizonLqr(A32, B32, Q32, R32, T)
        Lg = lqr.discreteFiniteHorizonLqr(A32, B32, Q32, R32, T)
        assert np.max(np.abs(Lg - L64)) <= max(2e-4 * np.max(np.abs(L64)), 4 * np.max(np.abs(L32 - L64))), ("f32", case, n, m, T, b)


def test_affine_and_ilqr_sweeps(mods):
    lqr, ilqr, pt = mods
    rng = np.random.default_rng(2026)
    for case in range(40):
        n, m = int(rng.integers(1, 13)), int(rng.integers(1, 5))
        T, b = int(rng.integers(1, 7)), int(rng.integers(1, 40))
        (f, f_x, f_u), (c, c_x, c_u, c_xx, c_ux, c_uu), (v, v_x, v_xx) = problems.random_ilqr_model(b, T, n, m, seed=3000 + case)
        pol = ilqr.backwardPass_ilqr(pt.AffineDynamics(f, f_x, f_u), pt.QuadraticCostFunction(c, c_x, c_u, c_xx, c_ux, c_uu),
                                     pt.QuadraticValueFunction(v, v_x, v_xx))
        for i in range(min(b, 3)):
            ref = zo.backwardPass_ilqr(zo.AffineDynamics(f[i], f_x[i], f_u[i]),
                                       zo.QuadraticCostFunction(c[i], c_x[i], c_u[i], c_xx[i], c_ux[i], c_uu[i]),
                                       zo.QuadraticValueFunction(v[i], v_x[i], v_xx[i]))
            assert _rel(pol.L[i], ref.L) <= 1e-9 and _rel(pol.l[i], ref.l) <= 1e-9, ("ilqr", case, n, m, T, b)
        # bilinear / affine LQR on the same random data: d = f, H = c_ux, q = c_x, r = c_u
        q0 = rng.standard_normal((b, T))
        Lg, lg = lqr.bilinearAffineLqr(f_x, f_u, f, c_xx, c_uu, c_ux, c_x, c_u, q0, T)
        for i in range(min(b, 3)):
            Lr, lr = zo.bilinearAffineLqr(f_x[i], f_u[i], f[i], c_xx[i], c_uu[i], c_ux[i], c_x[i], c_u[i], q0[i], T)
            assert _rel(Lg[i], Lr) <= 1e-9 and _rel(lg[i], lr) <= 1e-9, ("affine", case, n, m, T, b)


def test_dare_and_psd_sweeps(mods):
    lqr, ilqr, _ = mods
    rng = np.random.default_rng(2027)
    for case in range(20):
        n, m, b = int(rng.integers(1, 13)), int(rng.integers(1, 5)), int(rng.integers(1, 9))
        A, B, Q, R = problems.random_lti_systems(b, n, m, seed=4000 + case, rho=float(rng.uniform(0.5, 1.2)))
        L = lqr.discreteInfiniteHorizonLqr(A, B, Q, R)
        for i in range(b):
            Lr, _ = zo.discreteInfiniteHorizonLqr(A[i], B[i], Q[i], R[i])
            assert _rel(L[i], Lr) <= 1e-9, ("dare", case, n, m, i)
    for case in range(20):
        k, b = int(rng.integers(1, 17)), int(rng.integers(1, 30))
        M = rng.standard_normal((b, k, k))
        Amat = M + np.swapaxes(M, -1, -2)
        if case % 2:
            Amat[:, :, k // 2] = 0; Amat[:, k // 2, :] = 0            # exact zero row / column: rank-deficient input
        P = ilqr.ensurePositiveDefinite(Amat)
        w, vv = np.linalg.eigh(Amat)
        ref = (vv * np.maximum(w, 1e-3)[:, None, :]) @ np.swapaxes(vv, -1, -2)
        assert np.max(np.abs(P - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref))), ("psd", case, k, b)


def test_quadcopter_expansions_random_points_horizons_and_wind(mods):
    """K7 for the quadcopter (closed forms generated by sympy, quad_derivs_gen.h; 16 lanes per point, LDS-transposed stores) over random
    horizons, batch sizes that do not fill a workgroup, large attitude angles and wind: Jacobians against complex-step differentiation
    of the oracle model, packed second derivatives against the hyper-dual evaluation of the full-tensor entry point."""
    import ctypes
    import torch
    from zopt_amd import _lib, models
    _, _, pt = mods
    rng = np.random.default_rng(77)
    lib = _lib.lib()
    for case in range(12):
        b, N = int(rng.integers(1, 40)), int(rng.integers(1, 23))
        wind = (0.0, 0.0, 0.0) if case % 3 else tuple(rng.uniform(-4, 4, 3))
        dt = 0.0 if case == 5 else float(rng.uniform(0.02, 0.2))
        model = models.QuadcopterEuler(dt, wind_ned=wind)
        xT = rng.standard_normal((b, N + 1, 12)) * np.array([3, 3, 3, 1, 1, 1, 0.9, 0.9, 3.0, 5, 5, 5])
        uT = np.array([9.807, 0, 0, 0]) + rng.standard_normal((b, N, 4))
        ad = pt.AffineDynamics.from_trajectory(model, pt.Trajectory(xT, uT))
        h = 1e-30
        for (bi, k) in [(0, 0), (b - 1, N - 1), (int(rng.integers(0, b)), int(rng.integers(0, N)))]:
            J = np.zeros((12, 16))
            for j in range(16):
                z = np.concatenate([xT[bi, k], uT[bi, k]]).astype(complex)
                z[j] += 1j * h
                xd = zo.quad_inertialDynamics(z[:12], z[12:], np.array(wind))
                J[:, j] = np.imag(z[:12] + dt * xd if dt else xd) / h
            assert np.max(np.abs(ad.f_x[bi, k] - J[:, :12])) <= 1e-11 * max(1.0, np.abs(J).max()), (case, bi, k)
            assert np.max(np.abs(ad.f_u[bi, k] - J[:, 12:])) <= 1e-11 * max(1.0, np.abs(J).max()), (case, bi, k)
        # packed second derivatives (closed forms) against the full tensors (hyper-dual numbers)
        md = model.c_struct()
        dx, du = torch.as_tensor(xT, device="cuda"), torch.as_tensor(uT, device="cuda")
        H = torch.full((b, N, 28, 12), float("nan"), dtype=torch.float64, device="cuda")
        f_xx = torch.empty((b, N, 12, 12, 12), dtype=torch.float64, device="cuda")
        pairs, npairs = (ctypes.c_int32 * 64)(), ctypes.c_int32(0)
        _lib.check(lib.zm_model_hessian_pairs(ctypes.addressof(md), ctypes.addressof(pairs), ctypes.addressof(npairs)), "pairs")
        _lib.check(lib.zm_quadratic_dynamics_pairs_list_f64(ctypes.addressof(md), dx.data_ptr(), du.data_ptr(), None, 0, None, H.data_ptr(), b, N, None), "packed")
        _lib.check(lib.zm_quadratic_dynamics_f64(ctypes.addressof(md), dx.data_ptr(), du.data_ptr(), None, f_xx.data_ptr(), None, None, b, N, None), "full")
        torch.cuda.synchronize()
        Hn, F = H.cpu().numpy(), f_xx.cpu().numpy()
        ab = np.array(pairs[:56]).reshape(28, 2)
        scale = max(1.0, np.abs(F).max())
        for p, (a_, b_) in enumerate(ab):
            assert np.max(np.abs(Hn[:, :, p, :] - F[:, :, :, a_, b_])) <= 1e-12 * scale, (case, p)


def test_large_state_sweeps_rollouts_and_projection(mods):
    """The round-3 large-state kernels over random shapes: sweep_tiled_f64 (backwardPass_ilqr, bilinearAffineLqr; 12 < n <= 48 or
    4 < m <= 16), rollout_wide (forwardPass2 of linear models up to n <= 64, m <= 16), psd_tiled (ensurePositiveDefinite up to 64)."""
    lqr, ilqr, pt = mods
    from zopt_amd import models
    rng = np.random.default_rng(2027)
    for case in range(24):
        big_n = case % 2 == 0
        n = int(rng.integers(13, 49)) if big_n else int(rng.integers(1, 13))
        m = int(rng.integers(1, 17)) if big_n else int(rng.integers(5, 17))
        T, b = int(rng.integers(1, 6)), int(rng.integers(1, 5))
        dyn, cost, Vf = problems.random_ilqr_model(b, T, n, m, seed=5000 + case)
        if case % 3 == 0:
            c, c_x, c_u, c_xx, c_ux, c_uu = cost
            cost = (c, c_x, c_u, c_xx + 0.1 * rng.standard_normal(c_xx.shape), c_ux, c_uu + 0.1 * rng.standard_normal(c_uu.shape))
        pol = ilqr.backwardPass_ilqr(dyn, cost, Vf)
        for i in range(b):
            ref = zo.backwardPass_ilqr(zo.AffineDynamics(*(x[i] for x in dyn)), zo.QuadraticCostFunction(*(x[i] for x in cost)),
                                       zo.QuadraticValueFunction(*(x[i] for x in Vf)))
            assert _rel(pol.L[i], ref.L) <= 1e-10 and _rel(pol.l[i], ref.l) <= 1e-10, ("ilqr", case, n, m, T, b)
        A, B, Q, R = problems.random_time_varying(b, T, n, m, seed=6000 + case)
        d, H = 0.5 * rng.standard_normal((b, T, n)), 0.2 * rng.standard_normal((b, T, m, n))
        q, r, q0 = rng.standard_normal((b, T, n)), rng.standard_normal((b, T, m)), rng.standard_normal((b, T))
        L, l = lqr.bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, T)
        Lr, lr = zo.bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, T)
        assert _rel(L, Lr) <= 1e-10 and _rel(l, lr) <= 1e-10, ("affine", case, n, m, T, b)
    for case in range(12):
        n, m = int(rng.integers(13, 65)), int(rng.integers(1, 17))
        N, b = int(rng.integers(1, 9)), int(rng.integers(1, 4))
        model = models.LinearModel(rng.standard_normal((n, n)) * (1.0 / np.sqrt(n)), rng.standard_normal((n, m)))
        Mq = rng.standard_normal((n, n))
        cost = models.QuadraticCost(Mq @ Mq.T / n + np.eye(n), np.eye(m) + 0.1 * rng.standard_normal((m, m)), 10 * np.eye(n))
        x0, l = rng.standard_normal((b, n)), 2.0 * rng.standard_normal((b, N, m)) * rng.uniform(0.05, 3.0, (b, 1, 1))
        L = 0.2 * rng.standard_normal((b, N, m, n)) / np.sqrt(n)
        xp, up = rng.standard_normal((b, N + 1, n)), 0.3 * rng.standard_normal((b, N, m))
        traj, J = ilqr.forwardPass2(x0, model, cost, pt.AffinePolicy(l, L), pt.Trajectory(xp, up))
        for i in range(b):
            rt, rJ = zo.forwardPass2(x0[i], model, cost.runningCost, cost.terminalCost, zo.AffinePolicy(l[i], L[i]), zo.Trajectory(xp[i], up[i]))
            assert abs(J[i] - rJ) <= 1e-10 * abs(rJ) and _rel(traj.uTraj[i], rt.uTraj) <= 1e-9, ("rollout", case, n, m, N, b)
    for case in range(12):
        k = int(rng.integers(17, 65))
        M = rng.standard_normal((3, k, k))
        A = M + np.swapaxes(M, -1, -2)
        A[1] *= 1e-3                                            # a spectrum around the clamp
        out, ref = ilqr.ensurePositiveDefinite(A), zo.ensurePositiveDefinite(A)
        assert np.max(np.abs(out - ref)) <= 2e-11 * max(1.0, np.max(np.abs(ref))), ("psd", case, k)
