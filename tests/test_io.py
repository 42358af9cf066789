"""CPU tests of the .npz interchange schema (zopt_amd/io.py)."""
import numpy as np
import pytest

from tests import problems
from zopt_amd import io as zio


def test_lqr_round_trip(tmp_path):
    A, B, Q, R = problems.random_time_varying(2, 5, 4, 2, seed=1)
    L = np.zeros((2, 5, 2, 4))
    p = zio.save(tmp_path / "lqr.npz", "lqr", dict(A=A, B=B, Q=Q, R=R), dict(L=L))
    kind, prob, res = zio.load(p)
    assert kind == "lqr" and sorted(prob) == ["A", "B", "Q", "R"] and np.array_equal(prob["A"], A) and np.array_equal(res["L"], L)


def test_ilqr_and_mpc_round_trip_and_time_axis(tmp_path):
    prob = dict(x0=np.zeros((3, 12)), uGuess=np.zeros((3, 10, 4)), Q=np.eye(12), R=np.eye(4), Qf=10 * np.eye(12), dt=0.1,
                model="quadcopter")
    res = dict(xTraj=np.zeros((3, 11, 12)), uTraj=np.zeros((3, 10, 4)), L=np.zeros((3, 10, 4, 12)), J=np.ones(3),
               converged=np.array([True, False, True]))
    kind, p2, r2 = zio.load(zio.save(tmp_path / "ilqr.npz", "ilqr", prob, res))
    assert kind == "ilqr" and str(p2["model"]) == "quadcopter" and r2["converged"].dtype == bool
    assert np.allclose(r2["tArr"], np.arange(11) * 0.1)
    mprob = dict(A=np.eye(2), B=np.eye(2), Q=np.eye(2), R=np.eye(2), N=3, x_lb=-np.ones(2), x_ub=np.ones(2), u_lb=-np.ones(2),
                 u_ub=np.full(2, np.inf), x0=np.zeros((4, 2)))
    mres = dict(xTraj=np.zeros((4, 4, 2)), uTraj=np.zeros((4, 3, 2)), status=np.array(["optimal"] * 3 + ["infeasible"], dtype=object))
    kind, p3, r3 = zio.load(zio.save(tmp_path / "mpc.npz", "mpc", mprob, mres))
    assert kind == "mpc" and int(p3["N"]) == 3 and np.isinf(p3["u_ub"]).all() and list(r3["status"]) == ["optimal"] * 3 + ["infeasible"]


def test_rejects_incomplete_or_foreign_files(tmp_path):
    with pytest.raises(ValueError):
        zio.save(tmp_path / "x.npz", "lqr", dict(A=np.eye(2)))
    with pytest.raises(ValueError):
        zio.save(tmp_path / "x.npz", "lqr", dict(A=1, B=1, Q=1, R=1), dict(K=np.eye(2)))
    np.savez(tmp_path / "foreign.npz", a=np.eye(2))
    with pytest.raises(ValueError):
        zio.load(tmp_path / "foreign.npz")


def test_interpMapped_matches_componentwise_numpy_interp():
    """zopt.jaxUtils.interpMapped (jaxUtils.py:7-24): jnp.interp per row of fp, clipped ends by default."""
    import numpy as np
    from zopt_amd import jaxUtils
    xp = np.linspace(0.0, 2.0, 5)
    fp = np.stack([xp ** 2, -xp, np.ones_like(xp)])
    out = jaxUtils.interpMapped(0.75, xp, fp)
    assert out.shape == (3,)
    assert np.allclose(out, [0.5 * (0.25 + 1.0), -0.75, 1.0])
    assert np.allclose(jaxUtils.interpMapped(-1.0, xp, fp), fp[:, 0]) and np.allclose(jaxUtils.interpMapped(9.0, xp, fp), fp[:, -1])
    assert jaxUtils.interpMapped(np.array([0.1, 0.2]), xp, fp).shape == (3, 2)
    f = lambda a: a
    assert jaxUtils.maybeJit(f, True) is f and jaxUtils.maybeJitCls(f) is f


def test_plot_adapters_shapes_and_errors():
    """Arrays for plottingTools.plotTimeTrajectory (plottingTools.py:5-40) and mpcUtils.plotMpcTrajectory (mpcUtils.py:84-122)."""
    res = dict(xTraj=np.arange(3 * 11 * 12, dtype=np.float64).reshape(3, 11, 12), uTraj=np.zeros((3, 10, 4)))
    t, x = zio.time_trajectory_inputs(res, 0.1, "xTraj", index=2)
    assert t.shape == (11,) and x.shape == (11, 12) and np.allclose(t, np.arange(11) * 0.1) and np.array_equal(x, res["xTraj"][2])
    t, u = zio.time_trajectory_inputs(res, 0.1, "uTraj", index=(0,))
    assert t.shape == (10,) and u.shape == (10, 4)
    with pytest.raises(ValueError):
        zio.time_trajectory_inputs(res, 0.1)                   # batched without an index
    steps = [np.full((5, 31, 12), float(i)) for i in range(7)]   # 7 closed-loop steps, 5 instances, N_mpc = 31
    traj = zio.mpc_trajectory_array(steps, index=3)
    assert traj.shape == (7, 31, 12) and np.array_equal(traj[:, 0, 0], np.arange(7.0))
    with pytest.raises(ValueError):
        zio.mpc_trajectory_array(steps)
    with pytest.raises(ValueError):
        zio.mpc_trajectory_array([])
