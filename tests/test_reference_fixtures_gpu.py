"""HIP kernels against the REFERENCE-SOURCE fixtures (tests/golden/ref_n*_m*_T*.npz): outputs of the reference's own
`zopt/lqrUtils.py`, `zopt/ilqrUtils.py`, `zopt/pytrees.py`, `zopt/quadcopter.py` source on nonsymmetric, time-varying inputs
("reference source, NumPy semantics, fp64"; generator: tests/golden/make_reference_fixtures.py).  An independent pin: the
oracle is not involved in these comparisons at all.  zopt_amd.* -> ctypes -> C ABI -> HIP kernels.

Tolerances (fp64): 1e-10 relative on the sweeps and rollouts (measured 1e-14 .. 1e-12), 1e-9 where a PD projection
(matrix-sign iteration here, `eigh` in the reference) sits inside the chain."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_n*_m*_T*.npz")))
TOL = 1e-10


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


def _load(path):
    z = np.load(path, allow_pickle=False)
    return {k: (z[k].astype(np.float64) if z[k].dtype == np.float32 else z[k]) for k in z.files}


@pytest.fixture(scope="module", params=FILES, ids=[os.path.basename(f) for f in FILES])
def fx(request):
    import torch
    assert torch.cuda.is_available()
    return _load(request.param)


def test_fixtures_exist():
    assert len(FILES) == 4


def test_A1_discreteFiniteHorizonLqr(fx):
    from zopt_amd import lqrUtils
    L = lqrUtils.discreteFiniteHorizonLqr(fx["A1_A"], fx["A1_B"], fx["A1_Q"], fx["A1_R"], int(fx["T"]))
    assert rel(L, fx["A1_L"]) <= TOL
    L0 = lqrUtils.discreteFiniteHorizonLqr(fx["A1_A"][0], fx["A1_B"][0], fx["A1_Q"][0], fx["A1_R"][0], int(fx["T"]))   # reference shapes
    assert L0.shape == fx["A1_L"][0].shape and rel(L0, fx["A1_L"][0]) <= TOL


def test_A2_bilinearAffineLqr(fx):
    from zopt_amd import lqrUtils
    L, l = lqrUtils.bilinearAffineLqr(*(fx["A2_" + k] for k in ("A", "B", "d", "Q", "R", "H", "q", "r", "q0")), int(fx["T"]))
    assert rel(L, fx["A2_L"]) <= TOL and rel(l, fx["A2_l"]) <= TOL


def _a3(fx, T=None):
    s = slice(None) if T is None else slice(0, T)
    b = None if T is None else fx["A4_f_xx"].shape[0]
    dyn = [np.ascontiguousarray(fx["A3_" + k][:b, s]) for k in ("f", "f_x", "f_u")]
    cost = [np.ascontiguousarray(fx["A3_" + k][:b, s]) for k in ("c", "c_x", "c_u", "c_xx", "c_ux", "c_uu")]
    Vf = [np.ascontiguousarray(fx["A3_" + k][:b]) for k in ("v", "v_x", "v_xx")]
    return dyn, cost, Vf


def test_A3_backwardPass_ilqr(fx):
    from zopt_amd import ilqrUtils, pytrees as pt
    dyn, cost, Vf = _a3(fx)
    pol = ilqrUtils.backwardPass_ilqr(pt.AffineDynamics(*dyn), pt.QuadraticCostFunction(*cost), pt.QuadraticValueFunction(*Vf))
    assert rel(pol.l, fx["A3_l"]) <= TOL and rel(pol.L, fx["A3_L"]) <= TOL
    V, p = ilqrUtils.riccatiStep_ilqr(pt.AffineDynamics(*[x[0, -1] for x in dyn]), pt.QuadraticCostFunction(*[x[0, -1] for x in cost]),
                                      pt.QuadraticValueFunction(*[x[0] for x in Vf]))
    for got, key in ((V.v, "v"), (V.v_x, "v_x"), (V.v_xx, "v_xx"), (p.l, "l"), (p.L, "L")):
        assert rel(got, fx["A3_step_" + key]) <= TOL, key


def test_A4_backwardPass_ddp(fx):
    from zopt_amd import ilqrUtils, pytrees as pt
    Td = int(fx["A4_T"])
    dyn, cost, Vf = _a3(fx, Td)
    qd = pt.QuadraticDynamics(*dyn, fx["A4_f_xx"], fx["A4_f_ux"], fx["A4_f_uu"])
    pol = ilqrUtils.backwardPass_ddp(qd, pt.QuadraticCostFunction(*cost), pt.QuadraticValueFunction(*Vf))
    assert rel(pol.l, fx["A4_l"]) <= 1e-9 and rel(pol.L, fx["A4_L"]) <= 1e-9
    one = pt.QuadraticDynamics(dyn[0][0], dyn[1][0], dyn[2][0], fx["A4_f_xx"][0], fx["A4_f_ux"][0], fx["A4_f_uu"][0])
    vx = np.repeat(fx["A3_v_x"][0][None], Td, axis=0)
    vf_xx, vf_ux, vf_uu = ilqrUtils.conditionQuadraticDynamics(one, vx)
    assert rel(vf_xx, fx["A4_cond_vf_xx"]) <= 1e-9 and rel(vf_ux, fx["A4_cond_vf_ux"]) <= 1e-9 and rel(vf_uu, fx["A4_cond_vf_uu"]) <= 1e-9


def test_A5_positive_definite_projections(fx):
    from zopt_amd import ilqrUtils, pytrees as pt
    assert rel(ilqrUtils.ensurePositiveDefinite(fx["A5_a"]), fx["A5_psd"]) <= 1e-10
    T, n, m = int(fx["T"]), int(fx["n"]), int(fx["m"])
    cc = ilqrUtils.conditionQuadraticCost(pt.QuadraticCostFunction(np.zeros(T), np.zeros((T, n)), np.zeros((T, m)), fx["A5_c_xx"],
                                                                   fx["A5_c_ux"], fx["A5_c_uu"]))
    assert rel(cc.c_xx, fx["A5_cond_c_xx"]) <= 1e-10 and rel(cc.c_ux, fx["A5_cond_c_ux"]) <= 1e-10 and rel(cc.c_uu, fx["A5_cond_c_uu"]) <= 1e-10
    vv = ilqrUtils.conditionValueFunction(pt.QuadraticValueFunction(0.0, np.zeros(n), fx["A5_v_xx"]))
    assert rel(vv.v_xx, fx["A5_cond_v_xx"]) <= 1e-10


def test_A6_A7_linear_dynamics(fx):
    from zopt_amd import ilqrUtils, models, pytrees as pt
    model = models.LinearModel(fx["A6_lin_A"], fx["A6_lin_B"])
    cost = models.QuadraticCost(fx["A67_Q"], fx["A67_R"], fx["A67_Qf"])      # nonsymmetric weights
    pol = pt.AffinePolicy(fx["A67_l"], fx["A67_L"])
    prev = pt.Trajectory(fx["A67_xPrev"], fx["A67_uPrev"])
    for tag, alpha in (("a1", 1.0), ("a025", 0.25)):
        t = ilqrUtils.trajectoryRollout(fx["A67_x0"], model, pol, prev, alpha=alpha)
        assert rel(t.xTraj, fx[f"A6_lin_{tag}_xTraj"]) <= TOL and rel(t.uTraj, fx[f"A6_lin_{tag}_uTraj"]) <= TOL
    t, J = ilqrUtils.forwardPass2(fx["A67_x0"], model, cost, pol, prev)
    assert rel(t.xTraj, fx["A7_lin_xTraj"]) <= TOL and rel(t.uTraj, fx["A7_lin_uTraj"]) <= TOL
    assert np.allclose(J, fx["A7_lin_J"], rtol=1e-11, atol=0)


def test_A10_quadcopter_model_and_rollouts():
    fx = _load(os.path.join(GOLDEN, "ref_n12_m4_T50.npz"))       # the quadcopter (n = 12, m = 4) exists at this shape only
    from zopt_amd import ilqrUtils, models, pytrees as pt
    ac = models.Quadcopter()
    assert rel(ac.inertialDynamics(fx["A10_x"], fx["A10_u"]), fx["A10_xdot"]) <= 1e-12
    assert rel(ac.inertialDynamics(fx["A10_x"], fx["A10_u"], tuple(fx["A10_wind_ned"])), fx["A10_xdot_wind"]) <= 1e-12
    assert rel(ac.rigidBodyDynamics(fx["A10_x"][:, :8], fx["A10_u"]), fx["A10_rb_xdot"]) <= 1e-12
    model = models.QuadcopterEuler(0.1)
    cost = models.QuadraticCost(fx["A67_Q"], fx["A67_R"], fx["A67_Qf"])
    pol = pt.AffinePolicy(fx["A67q_l"], fx["A67q_L"])
    prev = pt.Trajectory(fx["A67q_xPrev"], fx["A67q_uPrev"])
    t = ilqrUtils.trajectoryRollout(fx["A67q_x0"], model, pol, prev, alpha=0.5)
    assert rel(t.xTraj, fx["A6_quad_a05_xTraj"]) <= TOL and rel(t.uTraj, fx["A6_quad_a05_uTraj"]) <= TOL
    t, J = ilqrUtils.forwardPass2(fx["A67q_x0"], model, cost, pol, prev)
    assert rel(t.xTraj, fx["A7_quad_xTraj"]) <= TOL and rel(t.uTraj, fx["A7_quad_uTraj"]) <= TOL
    assert np.allclose(J, fx["A7_quad_J"], rtol=1e-11, atol=0)
