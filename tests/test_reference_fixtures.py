"""The oracle against the REFERENCE-SOURCE fixtures (tests/golden/ref_n*_m*_T*.npz).

The fixtures hold inputs and the outputs of the reference's own `zopt/lqrUtils.py`, `zopt/ilqrUtils.py`, `zopt/pytrees.py`
and `zopt/quadcopter.py` source, executed in the build container under NumPy semantics by
tests/golden/make_reference_fixtures.py ("reference source, NumPy semantics, fp64": the reference's code and operation
order, evaluated by NumPy/LAPACK -- not XLA outputs).  Unlike the identity-matrix KATs of reference_kats.json every matrix
here is nonsymmetric and time-varying, so a transposition error in the oracle (B'VA vs A'VB, f_ux index order, V d vs V'd,
c_ux vs c_ux') fails these tests.  The GPU twin is tests/test_reference_fixtures_gpu.py.
"""
import glob
import os

import numpy as np
import pytest

from oracle import zopt_oracle as zo

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_n*_m*_T*.npz")))
TOL = 1e-11      # same NumPy/LAPACK arithmetic up to the association of a few products


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


def _load(path):
    z = np.load(path, allow_pickle=False)
    return {k: (z[k].astype(np.float64) if z[k].dtype == np.float32 else z[k]) for k in z.files}


def test_fixture_files_present_and_labelled():
    assert len(FILES) == 4
    for f in FILES:
        z = np.load(f, allow_pickle=False)
        assert "reference source" in str(z["label"]) and "NumPy semantics" in str(z["label"])
        # the inputs really are nonsymmetric / time-varying
        Q, A = z["A1_Q"].astype(np.float64), z["A1_A"].astype(np.float64)
        assert np.max(np.abs(Q - np.swapaxes(Q, -1, -2))) > 0.1 and np.max(np.abs(A[:, 0] - A[:, 1])) > 0.1


@pytest.fixture(scope="module", params=FILES, ids=[os.path.basename(f) for f in FILES])
def fx(request):
    return _load(request.param)


def test_A1_discreteFiniteHorizonLqr(fx):
    L = zo.discreteFiniteHorizonLqr(fx["A1_A"], fx["A1_B"], fx["A1_Q"], fx["A1_R"], int(fx["T"]))
    assert rel(L, fx["A1_L"]) <= TOL


def test_A2_bilinearAffineLqr(fx):
    L, l = zo.bilinearAffineLqr(*(fx["A2_" + k] for k in ("A", "B", "d", "Q", "R", "H", "q", "r", "q0")), int(fx["T"]))
    assert rel(L, fx["A2_L"]) <= TOL and rel(l, fx["A2_l"]) <= TOL


def _a3(fx, T=None):
    s = slice(None) if T is None else slice(0, T)
    b = None if T is None else fx["A4_f_xx"].shape[0]
    dyn = [fx["A3_" + k][:b, s] for k in ("f", "f_x", "f_u")]
    cost = [fx["A3_" + k][:b, s] for k in ("c", "c_x", "c_u", "c_xx", "c_ux", "c_uu")]
    Vf = [fx["A3_" + k][:b] for k in ("v", "v_x", "v_xx")]
    return dyn, cost, Vf


def test_A3_backwardPass_ilqr(fx):
    dyn, cost, Vf = _a3(fx)
    pol = zo.backwardPass_ilqr(zo.AffineDynamics(*dyn), zo.QuadraticCostFunction(*cost), zo.QuadraticValueFunction(*Vf))
    assert rel(pol.l, fx["A3_l"]) <= TOL and rel(pol.L, fx["A3_L"]) <= TOL
    V, p = zo.riccatiStep_ilqr([x[0, -1] for x in dyn], [x[0, -1] for x in cost], [x[0] for x in Vf])
    for got, key in ((V.v, "v"), (V.v_x, "v_x"), (V.v_xx, "v_xx"), (p.l, "l"), (p.L, "L")):
        assert rel(got, fx["A3_step_" + key]) <= TOL, key


def test_A4_backwardPass_ddp(fx):
    Td = int(fx["A4_T"])
    dyn, cost, Vf = _a3(fx, Td)
    qd = zo.QuadraticDynamics(*dyn, fx["A4_f_xx"], fx["A4_f_ux"], fx["A4_f_uu"])
    pol = zo.backwardPass_ddp(qd, zo.QuadraticCostFunction(*cost), zo.QuadraticValueFunction(*Vf))
    assert rel(pol.l, fx["A4_l"]) <= 1e-9 and rel(pol.L, fx["A4_L"]) <= 1e-9      # an eigendecomposition per step in the chain
    vf_xx, vf_ux, vf_uu = zo.conditionQuadraticDynamics(
        zo.QuadraticDynamics(None, None, None, fx["A4_f_xx"][0], fx["A4_f_ux"][0], fx["A4_f_uu"][0]), fx["A3_v_x"][0])
    assert rel(vf_xx, fx["A4_cond_vf_xx"]) <= TOL and rel(vf_ux, fx["A4_cond_vf_ux"]) <= TOL and rel(vf_uu, fx["A4_cond_vf_uu"]) <= TOL


def test_A5_positive_definite_projections(fx):
    assert rel(zo.ensurePositiveDefinite(fx["A5_a"]), fx["A5_psd"]) <= TOL
    T, n, m = int(fx["T"]), int(fx["n"]), int(fx["m"])
    cc = zo.conditionQuadraticCost(zo.QuadraticCostFunction(np.zeros(T), np.zeros((T, n)), np.zeros((T, m)), fx["A5_c_xx"],
                                                            fx["A5_c_ux"], fx["A5_c_uu"]))
    assert rel(cc.c_xx, fx["A5_cond_c_xx"]) <= TOL and rel(cc.c_ux, fx["A5_cond_c_ux"]) <= TOL and rel(cc.c_uu, fx["A5_cond_c_uu"]) <= TOL
    vv = zo.conditionValueFunction(zo.QuadraticValueFunction(0.0, np.zeros(n), fx["A5_v_xx"]))
    assert rel(vv.v_xx, fx["A5_cond_v_xx"]) <= TOL


def test_A6_A7_linear_dynamics(fx):
    A, B = fx["A6_lin_A"], fx["A6_lin_B"]
    dyn = lambda x, u: A @ x + B @ u                                # noqa: E731
    rc, tc = zo.quadratic_costs(fx["A67_Q"], fx["A67_R"], fx["A67_Qf"])
    for i in range(fx["A67_x0"].shape[0]):
        pol = zo.AffinePolicy(fx["A67_l"][i], fx["A67_L"][i])
        prev = zo.Trajectory(fx["A67_xPrev"][i], fx["A67_uPrev"][i])
        for tag, alpha in (("a1", 1.0), ("a025", 0.25)):
            t = zo.trajectoryRollout(fx["A67_x0"][i], dyn, pol, prev, alpha=alpha)
            assert rel(t.xTraj, fx[f"A6_lin_{tag}_xTraj"][i]) <= TOL and rel(t.uTraj, fx[f"A6_lin_{tag}_uTraj"][i]) <= TOL
        t, J = zo.forwardPass2(fx["A67_x0"][i], dyn, rc, tc, pol, prev)
        assert rel(t.xTraj, fx["A7_lin_xTraj"][i]) <= TOL and rel(t.uTraj, fx["A7_lin_uTraj"][i]) <= TOL
        assert J == pytest.approx(fx["A7_lin_J"][i], rel=1e-12)


def test_A10_quadcopter_model_and_rollouts():
    fx = _load(os.path.join(GOLDEN, "ref_n12_m4_T50.npz"))       # the quadcopter (n = 12, m = 4) exists at this shape only
    xs, us = fx["A10_x"], fx["A10_u"]
    for i in range(xs.shape[0]):
        assert rel(zo.quad_inertialDynamics(xs[i], us[i]), fx["A10_xdot"][i]) <= 1e-13
        assert rel(zo.quad_inertialDynamics(xs[i], us[i], fx["A10_wind_ned"]), fx["A10_xdot_wind"][i]) <= 1e-13
        assert rel(zo.quad_rigidBodyDynamics(xs[i, :8], us[i]), fx["A10_rb_xdot"][i]) <= 1e-13
    step = zo.quad_euler_step(0.1)
    rc, tc = zo.quadratic_costs(fx["A67_Q"], fx["A67_R"], fx["A67_Qf"])
    for i in range(fx["A67q_x0"].shape[0]):
        pol = zo.AffinePolicy(fx["A67q_l"][i], fx["A67q_L"][i])
        prev = zo.Trajectory(fx["A67q_xPrev"][i], fx["A67q_uPrev"][i])
        t = zo.trajectoryRollout(fx["A67q_x0"][i], step, pol, prev, alpha=0.5)
        assert rel(t.xTraj, fx["A6_quad_a05_xTraj"][i]) <= TOL and rel(t.uTraj, fx["A6_quad_a05_uTraj"][i]) <= TOL
        t, J = zo.forwardPass2(fx["A67q_x0"][i], step, rc, tc, pol, prev)
        assert rel(t.xTraj, fx["A7_quad_xTraj"][i]) <= TOL and rel(t.uTraj, fx["A7_quad_uTraj"][i]) <= TOL
        assert J == pytest.approx(fx["A7_quad_J"][i], rel=1e-12)
