"""Child process of tests/test_ab_switches_gpu.py: one small parity pass over every kernel family an A/B switch can re-route,
with whatever ZOPT_AMD_* variables the parent put into the environment (the switches are read once per process, in static
initialisers).  Prints "AB-PARITY-OK" and exits 0 when every result agrees with the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import zopt_oracle as zo  # noqa: E402
from tests import problems  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


def main():
    from zopt_amd import ilqrUtils, lqrUtils, models, mpcUtils, pytrees as pt
    # K1 (DMA ring: G4 / ring depth), both fast-path shapes, a horizon that is not a multiple of the ring depth
    for n, m, T in ((12, 4, 11), (8, 4, 7)):
        A, B, Q, R = problems.random_time_varying(9, T, n, m, seed=n + T)
        assert rel(lqrUtils.discreteFiniteHorizonLqr(A, B, Q, R, T), zo.discreteFiniteHorizonLqr(A, B, Q, R, T)) <= 1e-10
    # fp32 arrays at the fast-path shape: K1 on fp32 storage (fp64 arithmetic) or, with ZOPT_AMD_LQR_F32=tile, the fp32 tile kernel
    A, B, Q, R = problems.random_time_varying(6, 9, 12, 4, seed=21, dtype=np.float32)
    L32 = lqrUtils.discreteFiniteHorizonLqr(A, B, Q, R, 9)
    assert L32.dtype == np.float32 and rel(L32, zo.discreteFiniteHorizonLqr(*(X.astype(np.float64) for X in (A, B, Q, R)), 9)) <= 2e-5
    # large-state fp64 (tile / LDS coverage kernel)
    A, B, Q, R = problems.random_time_varying(2, 5, 24, 8, seed=5)
    assert rel(lqrUtils.discreteFiniteHorizonLqr(A, B, Q, R, 5), zo.discreteFiniteHorizonLqr(A, B, Q, R, 5)) <= 1e-10
    # K2 / K3 (LDS-DMA ring or register prefetch)
    rng = np.random.default_rng(3)
    n, m, T, b = 12, 4, 13, 5
    (f, f_x, f_u), (c, c_x, c_u, c_xx, c_ux, c_uu), (v, v_x, v_xx) = problems.random_ilqr_model(b, T, n, m, seed=4)
    pol = ilqrUtils.backwardPass_ilqr(pt.AffineDynamics(f, f_x, f_u), pt.QuadraticCostFunction(c, c_x, c_u, c_xx, c_ux, c_uu),
                                      pt.QuadraticValueFunction(v, v_x, v_xx))
    ref = zo.backwardPass_ilqr(zo.AffineDynamics(f, f_x, f_u), zo.QuadraticCostFunction(c, c_x, c_u, c_xx, c_ux, c_uu),
                               zo.QuadraticValueFunction(v, v_x, v_xx))
    assert rel(pol.l, ref.l) <= 1e-10 and rel(pol.L, ref.L) <= 1e-10
    A, B, Q, R = problems.random_time_varying(b, T, 8, 4, seed=6)
    d, H = 0.3 * rng.standard_normal((b, T, 8)), 0.3 * rng.standard_normal((b, T, 4, 8))
    q, r, q0 = rng.standard_normal((b, T, 8)), rng.standard_normal((b, T, 4)), rng.standard_normal((b, T))
    L, l = lqrUtils.bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, T)
    Lr, lr = zo.bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, T)
    assert rel(L, Lr) <= 1e-10 and rel(l, lr) <= 1e-10
    # K6 rollout + line search (fast / generic), and the iLQR driver (host synchronisation interval) on the quadcopter
    N = 20
    Qc, Rc, Qf = np.eye(12), np.eye(4), 10 * np.eye(12)
    cost = models.QuadraticCost(Qc, Rc, Qf)
    model = models.QuadcopterEuler(0.1)
    x0 = np.zeros((3, 12))
    x0[:, 9:12] = [[1, -2, 3], [0, 5, 0], [-4, 1, 2]]
    ug = np.tile(models.QuadcopterEuler.uTrim, (3, N, 1))
    traj, Lg, J, conv = ilqrUtils.iterativeLqr(model, cost, cost, x0, ug)
    step = zo.quad_euler_step(0.1)
    for i in range(3):
        rt, rL, rJ, rc = zo.iterativeLqr(step, Qc, Rc, Qf, x0[i], ug[i])
        assert bool(conv[i]) == rc and abs(J[i] - rJ) <= 1e-7 * abs(rJ) and rel(traj.uTraj[i], rt.uTraj) <= 1e-6
    polq = pt.AffinePolicy(0.05 * rng.standard_normal((3, N, 4)), 0.02 * rng.standard_normal((3, N, 4, 12)))
    prev = pt.Trajectory(0.1 * rng.standard_normal((3, N + 1, 12)), ug + 0.05 * rng.standard_normal((3, N, 4)))
    t2, J2 = ilqrUtils.forwardPass2(x0, model, cost, polq, prev)
    rcost, tcost = zo.quadratic_costs(Qc, Rc, Qf)
    for i in range(3):
        rt, rJ = zo.forwardPass2(x0[i], step, rcost, tcost, zo.AffinePolicy(polq.l[i], polq.L[i]), zo.Trajectory(prev.xTraj[i], prev.uTraj[i]))
        assert rel(t2.xTraj[i], rt.xTraj) <= 1e-10 and abs(J2[i] - rJ) <= 1e-11 * abs(rJ)
    # K9 MPC (16 lanes per instance / lane per instance): the reference's test problem, Riccati optimum u0 = (-0.6, -0.6)
    I, one = np.eye(2), np.ones(2)
    u, tr, status = mpcUtils.lqrMpc(I, I, I, I, 2, -one, one, -one, one).solve(one, eps_abs=1e-9, eps_rel=1e-9)
    assert status == "optimal" and np.max(np.abs(u - [-0.6, -0.6])) <= 1e-7
    print("AB-PARITY-OK")


if __name__ == "__main__":
    main()
