"""GPU tests of the closed-loop simulation mirror (simulator.py:124-138 with the control laws of
lqrUtils.py:266-269 and demos/iterativeLqr.py:16-17, wind of demos/iterativeLqr.py:48) against plain NumPy loops over
the oracle's quadcopter restatement (1e-10: same formulas, different FMA contraction / sin-cos implementation)."""
import numpy as np
import pytest

from oracle import zopt_oracle as zo

pytestmark = pytest.mark.gpu
WIND = np.array([3.0, 1.0, 0.0])


def _np_loop(x0, law, N, dt, wind):
    xs, us = [np.asarray(x0, dtype=np.float64)], []
    for k in range(N):
        u = law(k, xs[-1])
        us.append(u)
        xs.append(xs[-1] + dt * zo.quad_inertialDynamics(xs[-1], u, wind_ned=wind))
    return np.array(xs), np.array(us)


def test_linearisation_with_wind():
    from zopt_amd import models, pytrees
    rng = np.random.default_rng(0)
    x = 0.3 * rng.standard_normal((4, 12))
    u = models.QuadcopterEuler.uTrim + 0.3 * rng.standard_normal((4, 4))
    dyn = pytrees.AffineDynamics.from_function(models.QuadcopterEuler(0.1, wind_ned=WIND), x, u)
    step = lambda x_, u_: x_ + 0.1 * zo.quad_inertialDynamics(x_, u_, wind_ned=WIND)
    for b in range(4):
        f, fx, fu = zo.jacobians(step, x[b], u[b])
        assert np.max(np.abs(dyn.f[b] - f)) <= 1e-13
        assert np.max(np.abs(dyn.f_x[b] - fx)) <= 1e-12 and np.max(np.abs(dyn.f_u[b] - fu)) <= 1e-12
    still = pytrees.AffineDynamics.from_function(models.QuadcopterEuler(0.1), x, u)
    assert np.max(np.abs(still.f - dyn.f)) > 1e-3          # the wind does act


def test_tracking_controller_closed_loop_with_wind():
    """demos/iterativeLqr.py:41-56: iLQR solution tracked in a windy simulation."""
    from zopt_amd import ilqrUtils, models, simulator
    dt, N = 0.1, 40
    x0 = np.zeros((3, 12))
    x0[:, 9:12] = [[2, 1, -1], [-1, 2, 0.5], [0.5, 0.5, 0.5]]
    ug = np.tile(models.QuadcopterEuler.uTrim, (3, N, 1))
    cost = models.QuadraticCost(np.eye(12), np.eye(4), 10 * np.eye(12))
    traj, LArr, J, conv = ilqrUtils.iterativeLqr(models.QuadcopterEuler(dt), cost, cost, x0, ug)
    sim = simulator.simulateTrackingController(models.QuadcopterEuler(dt, wind_ned=WIND), x0, LArr, traj)
    assert sim.xTraj.shape == (3, N + 1, 12) and sim.uTraj.shape == (3, N, 4)
    for b in range(3):
        xs, us = _np_loop(x0[b], lambda k, x: LArr[b, k] @ (x - traj.xTraj[b, k]) + traj.uTraj[b, k], N, dt, WIND)
        assert np.max(np.abs(sim.xTraj[b] - xs)) <= 1e-10 * max(1.0, np.max(np.abs(xs)))
        assert np.max(np.abs(sim.uTraj[b] - us)) <= 1e-10 * max(1.0, np.max(np.abs(us)))
    # without wind the simulation reproduces the planned trajectory itself
    calm = simulator.simulateTrackingController(models.QuadcopterEuler(dt), x0, LArr, traj)
    assert np.max(np.abs(calm.xTraj - traj.xTraj)) <= 1e-9
    assert np.max(np.abs(sim.xTraj - traj.xTraj)) > 1e-2


def test_hover_regulation_pipeline_on_device():
    """Producer -> path -> consumer without leaving the GPU stack: hover linearisation (AffineDynamics.from_function,
    demos/lqrMpc.py:26-28), DARE gain (discreteInfiniteHorizonLqr), finite-horizon gains (discreteFiniteHorizonLqr), and the
    nonlinear closed loop `u = -K (x - x0) + u0` (lqrUtils.py:266-269)."""
    from zopt_amd import lqrUtils, models, pytrees, simulator
    dt, N = 0.1, 150
    model = models.QuadcopterEuler(dt)
    xT, uT = np.zeros(12), models.QuadcopterEuler.uTrim
    lin = pytrees.AffineDynamics.from_function(model, xT, uT)
    A, B = lin.f_x, lin.f_u
    Q, R = np.eye(12), np.eye(4)
    K = lqrUtils.discreteInfiniteHorizonLqr(A, B, Q, R)
    Kr, _ = zo.discreteInfiniteHorizonLqr(A, B, Q, R)
    assert np.max(np.abs(K - Kr)) <= 1e-9 * np.max(np.abs(Kr))
    rng = np.random.default_rng(5)
    x0 = np.zeros((8, 12))
    x0[:, 9:12] = rng.uniform(-1, 1, (8, 3))
    x0[:, 6:8] = rng.uniform(-0.1, 0.1, (8, 2))
    sim = simulator.simulateProportionalFeedback(model, x0, np.broadcast_to(K, (8, 4, 12)), xT, uT, N=N)
    assert sim.xTraj.shape == (8, N + 1, 12)
    xs, us = _np_loop(x0[2], lambda k, x: -K @ (x - xT) + uT, N, dt, np.zeros(3))
    assert np.max(np.abs(sim.xTraj[2] - xs)) <= 1e-10 and np.max(np.abs(sim.uTraj[2] - us)) <= 1e-10
    assert np.max(np.abs(sim.xTraj[:, -1])) <= 5e-2 * np.max(np.abs(x0))      # regulated to hover
    # time-indexed gains of the finite-horizon sweep drive the same loop
    T = 60
    tile = lambda X: np.ascontiguousarray(np.broadcast_to(X, (8, T) + X.shape))
    KT = lqrUtils.discreteFiniteHorizonLqr(tile(A), tile(B), tile(Q), tile(R), T)
    simT = simulator.simulateProportionalFeedback(model, x0, KT, xT, uT)
    xs, us = _np_loop(x0[5], lambda k, x: -KT[5, k] @ (x - xT) + uT, T, dt, np.zeros(3))
    assert simT.xTraj.shape == (8, T + 1, 12) and np.max(np.abs(simT.xTraj[5] - xs)) <= 1e-10
