"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/zopt_amd.h declares, rejects bad arguments without touching a GPU, and the Python surface
mirrors the reference's names/signatures."""
import inspect
import os
import re

import pytest

from zopt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "zopt_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(zm_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = _declared_symbols()
    assert names, "no symbols parsed from include/zopt_amd.h"
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/zopt_amd.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes signature in zopt_amd/_lib.py"
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_support_matrix():
    lib = _lib.lib()
    assert lib.zm_version() >= 100
    assert lib.zm_lqr_backward_supported(12, 4, 8) == 1
    assert lib.zm_lqr_backward_supported(4, 1, 8) == 1
    assert lib.zm_lqr_backward_supported(8, 4, 8) == 1
    assert lib.zm_lqr_backward_supported(64, 16, 4) == 1   # config 5: tiled fp32 MFMA kernel
    assert lib.zm_lqr_backward_supported(64, 16, 8) == 1   # fp64 LDS coverage kernel
    assert lib.zm_lqr_backward_supported(65, 16, 8) == 0 and lib.zm_lqr_backward_supported(12, 17, 4) == 0
    assert lib.zm_lqr_backward_supported(0, 1, 8) == 0


def test_bad_arguments_return_codes_without_gpu():
    lib = _lib.lib()
    rc = lib.zm_lqr_backward_f64(None, None, None, None, None, 1, 1, 2, 2, None)
    assert rc == _lib.ZM_EINVAL
    assert b"null" in lib.zm_last_error()
    dummy = 0x1000
    rc = lib.zm_lqr_backward_f64(dummy, dummy, dummy, dummy, dummy, 1, 5, 65, 16, None)
    assert rc == _lib.ZM_EUNSUPPORTED
    rc = lib.zm_lqr_backward_f64(dummy, dummy, dummy, dummy, dummy, -1, 5, 2, 2, None)
    assert rc == _lib.ZM_EINVAL
    assert lib.zm_care_f64(dummy, dummy, dummy, dummy, dummy, None, None, 1, 17, 2, 1e-14, 60, None) == _lib.ZM_EUNSUPPORTED
    assert lib.zm_care_f64(dummy, dummy, dummy, dummy, None, None, None, 1, 4, 2, 1e-14, 60, None) == _lib.ZM_EINVAL
    assert lib.zm_riccati_ode_f64(dummy, dummy, dummy, dummy, dummy, dummy, None, 1, 4, 2, 1, 50, -1.0, 1e-8, 1e-8, 100,
                                  None) == _lib.ZM_EINVAL
    assert lib.zm_riccati_ode_f64(dummy, dummy, dummy, dummy, dummy, dummy, None, 1, 17, 2, 1, 50, 1.0, 1e-8, 1e-8, 100,
                                  None) == _lib.ZM_EUNSUPPORTED
    with pytest.raises(ValueError):
        _lib.check(rc, "x")


def test_python_surface_mirrors_reference_signatures():
    """Same names and positional parameters as the reference modules, checked without a GPU:
    lqrUtils.py:144, :176, :207, :266; ilqrUtils.py:33, :116, :176, :209, :217, :222, :254, :261, :331; mpcUtils.py:14, :61;
    pytrees.py field orders and constructors."""
    from zopt_amd import ilqrUtils, lqrUtils, mpcUtils, pytrees
    params = lambda f: list(inspect.signature(f).parameters)
    assert params(lqrUtils.discreteFiniteHorizonLqr) == ["A", "B", "Q", "R", "N"]
    assert params(lqrUtils.discreteInfiniteHorizonLqr)[:4] == ["A", "B", "Q", "R"]
    assert params(lqrUtils.bilinearAffineLqr) == ["A", "B", "d", "Q", "R", "H", "q", "r", "q0", "N"]
    assert params(lqrUtils.infiniteHorizonLqr)[:4] == ["A", "B", "Q", "R"]                              # lqrUtils.py:13
    assert params(lqrUtils.infiniteHorizonIntegralLqr) == ["A", "B", "Q", "R", "Qi", "Ci"]              # lqrUtils.py:101
    assert params(lqrUtils.finiteHorizonLqr)[:7] == ["A", "B", "Q", "R_inv", "Qf", "T", "N"]            # lqrUtils.py:55
    assert inspect.signature(lqrUtils.finiteHorizonLqr).parameters["N"].default == 50
    assert params(lqrUtils.proportionalFeedbackController) == ["x", "x0", "u0", "K"]
    assert params(ilqrUtils.trajectoryRollout) == ["x0", "dynFun", "policy", "trajPrev", "alpha"]
    assert params(ilqrUtils.forwardPass2) == ["x0", "dynFun", "costFun", "policy", "trajPrev"]
    assert params(ilqrUtils.riccatiStep_ilqr) == ["dynamics", "cost", "value"]
    assert params(ilqrUtils.riccatiStep_ddp) == ["dynamics", "cost", "value"]
    assert params(ilqrUtils.backwardPass_ilqr) == ["dynamics", "cost", "Vf"]
    assert params(ilqrUtils.backwardPass_ddp) == ["dynamics", "cost", "Vf"]
    assert params(ilqrUtils.ensurePositiveDefinite) == ["a", "eps"]
    assert params(ilqrUtils.conditionQuadraticCost) == ["quadratic_cost"]
    assert params(ilqrUtils.conditionValueFunction) == ["Vf"]
    assert params(ilqrUtils.conditionQuadraticDynamics) == ["quadratic_dynamics", "v_x"]
    sig = ["dynamics", "runningCost", "terminalCost", "x0", "uGuess", "maxIter", "tol"]
    assert params(ilqrUtils.iterativeLqr) == sig and params(ilqrUtils.differentialDynamicProgramming) == sig
    assert inspect.signature(ilqrUtils.iterativeLqr).parameters["maxIter"].default == 100
    assert inspect.signature(ilqrUtils.iterativeLqr).parameters["tol"].default == 1e-3
    assert params(mpcUtils.lqrMpc.__init__) == ["self", "A", "B", "Q", "R", "N", "x_lb", "x_ub", "u_lb", "u_ub", "Qf"]
    assert params(mpcUtils.lqrMpc.solve) == ["self", "x0", "kwargs"]
    assert pytrees.Trajectory._fields == ("xTraj", "uTraj")
    assert pytrees.QuadraticValueFunction._fields == ("v", "v_x", "v_xx")
    assert pytrees.QuadraticCostFunction._fields == ("c", "c_x", "c_u", "c_xx", "c_ux", "c_uu")
    assert pytrees.AffineDynamics._fields == ("f", "f_x", "f_u")
    assert pytrees.QuadraticDynamics._fields == ("f", "f_x", "f_u", "f_xx", "f_ux", "f_uu")
    assert pytrees.AffinePolicy._fields == ("l", "L")
    for cls in (pytrees.AffineDynamics, pytrees.QuadraticDynamics, pytrees.QuadraticCostFunction):
        assert callable(cls.from_function) and callable(cls.from_trajectory)
    assert callable(pytrees.QuadraticValueFunction.fromTerminalCostFunction)
    from zopt_amd import quadcopter
    for name in ("trim", "linearize", "rigidBodyDynamics", "inertialDynamics"):                       # quadcopter.py:70-201
        assert callable(getattr(quadcopter.Quadcopter, name))


def test_product_never_imports_oracle():
    """The product path must not route through the oracle / any CPU fallback."""
    pkg = os.path.join(ROOT, "zopt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f"{f} imports the oracle"
                assert "zopt_oracle" not in txt and "c_oracle" not in txt, f"{f} references the oracle"


def test_no_gpu_means_loud_failure():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from zopt_amd import lqrUtils
    I = np.repeat(np.eye(2)[None], 2, axis=0)
    with pytest.raises(_lib.ZoptAmdError):
        lqrUtils.discreteFiniteHorizonLqr(I, I, I, I, 2)


def test_lab_switches_exist_only_in_the_lab_build():
    """The product library reads four environment variables, each selecting a fallback kernel; the kernel lab's A/B switches are
    compiled in only with -DZM_LAB (zm_common.h: lab_env) -- their names must not even occur in the product binary."""
    from zopt_amd import _lib
    prod = open(_lib.LIB_PATH, "rb").read()
    lab = open(_lib.LAB_LIB_PATH, "rb").read()
    for name in (b"ZOPT_AMD_LQR_PATH", b"ZOPT_AMD_ILQR_PATH", b"ZOPT_AMD_ROLLOUT_PATH", b"ZOPT_AMD_MPC_PATH"):
        assert name in prod and name in lab
    for name in (b"ZOPT_AMD_ILQR_TAIL", b"ZOPT_AMD_JAC", b"ZOPT_AMD_HES", b"ZOPT_AMD_ILQR_SWAP", b"ZOPT_AMD_EXPAND", b"ZOPT_AMD_SWEEP_WG4",
                 b"ZOPT_AMD_LQR_G4", b"ZOPT_AMD_LQR_D", b"ZOPT_AMD_LQR_F32", b"ZOPT_AMD_ROLLOUT_QUAD", b"ZOPT_AMD_QUAD_ALL_MAX",
                 b"ZOPT_AMD_TILED_GENERIC"):
        assert name not in prod, name
        assert name in lab, name
    import re
    assert sorted(set(re.findall(rb"ZOPT_AMD_[A-Z0-9_]+", prod))) == [b"ZOPT_AMD_ILQR_PATH", b"ZOPT_AMD_LQR_PATH", b"ZOPT_AMD_MPC_PATH",
                                                                       b"ZOPT_AMD_ROLLOUT_PATH"]


def test_ilqr_workspace_omits_the_all_store_scratch_where_the_line_search_cannot_use_it():
    """zm_ilqr_solve_workspace_f64 (host-only sizing): the all-store scratch of the tail's line search -- 16 step sizes x 16 doubles
    per step for up to 2048 trajectories, 423 MB at T = 100 -- is reserved only for models whose line search can run in that form
    (still-air quadcopter); a windy quadcopter's workspace is smaller by exactly that block."""
    import ctypes
    from zopt_amd import _lib, models
    lib = _lib.lib()
    still, windy = models.QuadcopterEuler(0.1).c_struct(), models.QuadcopterEuler(0.1, wind_ned=(1.0, 0.0, 0.0)).c_struct()
    for batch, T in ((4096, 100), (100, 7)):
        a = lib.zm_ilqr_solve_workspace_f64(ctypes.addressof(still), batch, T, 0)
        b = lib.zm_ilqr_solve_workspace_f64(ctypes.addressof(windy), batch, T, 0)
        assert a > 0 and b > 0
        assert a - b == min(batch, 2048) * (T + 1) * 256


def test_lqrMpc_constructor_refuses_what_cvxpy_refuses():
    """host logic, no GPU: inconsistent shapes and weights that are not positive semidefinite (the reference's cvxpy problem is then
    not DCP and `solve` raises; mpcUtils.py:49-59) are refused when the problem is built; a singular PSD weight is accepted"""
    import numpy as np
    import pytest
    from zopt_amd import mpcUtils
    A, B = np.eye(2), np.ones((2, 1))
    lb, ub = -np.ones(2), np.ones(2)
    mpcUtils.lqrMpc(A, B, np.diag([1.0, 0.0]), np.zeros((1, 1)), 3, lb, ub, -np.ones(1), np.ones(1))          # PSD, singular: fine
    with pytest.raises(ValueError, match="not positive semidefinite"):
        mpcUtils.lqrMpc(A, B, np.diag([1.0, -0.1]), np.eye(1), 3, lb, ub, -np.ones(1), np.ones(1))
    with pytest.raises(ValueError, match="Qf is not positive semidefinite"):
        mpcUtils.lqrMpc(A, B, np.eye(2), np.eye(1), 3, lb, ub, -np.ones(1), np.ones(1), Qf=np.array([[0.0, 1.0], [1.0, 0.0]]))
    with pytest.raises(ValueError, match="shapes"):
        mpcUtils.lqrMpc(A, B, np.eye(3), np.eye(1), 3, lb, ub, -np.ones(1), np.ones(1))
