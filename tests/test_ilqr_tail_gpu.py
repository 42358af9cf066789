"""The solvers' line search has two forms (zopt_amd/csrc/ilqr_solve.hip): the two-pass one (winner re-rolled) and, once at most
ZOPT_AMD_ILQR_TAIL trajectories are left, the all-store one (every step size's rollout kept in scratch, the accept step copies the
winner).  Both run the same arithmetic, so a solve must come out BIT-identical whichever threshold is set -- never, always, or
switching in the middle of the solve.  The threshold is read once per process: one child process per value."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the A/B switches exist only in the -DZM_LAB build of the library (zm_common.h: lab_env); the child processes load that one
LAB_ENVIRON = dict(os.environ, ZOPT_AMD_LIB=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zopt_amd", "csrc",
                                                          "libzopt_amd_lab.so"))


CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from zopt_amd import ilqrUtils, models
ddp = sys.argv[2] == "ddp"
rng = np.random.default_rng(11)
B, N = 96, int(sys.argv[3]) if len(sys.argv) > 3 else 30
x0 = np.zeros((B, 12)); x0[:, 9:12] = rng.uniform(-10, 10, (B, 3))
ug = np.tile(models.QuadcopterEuler.uTrim, (B, N, 1))
cost = models.QuadraticCost(np.eye(12), (0.2 if ddp else 1.0) * np.eye(4), 10 * np.eye(12))
solve = ilqrUtils.differentialDynamicProgramming if ddp else ilqrUtils.iterativeLqr
wind = (3.0, 1.0, -0.5) if (len(sys.argv) > 4 and sys.argv[4] == "wind") else (0.0, 0.0, 0.0)
traj, L, J, conv = solve(models.QuadcopterEuler(0.1, wind_ned=wind), cost, cost, x0, ug)
np.savez(sys.argv[1], x=np.asarray(traj.xTraj), u=np.asarray(traj.uTraj), L=np.asarray(L), J=np.asarray(J), c=np.asarray(conv))
print("CHILD-OK", int(np.sum(conv)))
""" % ROOT


@pytest.mark.parametrize("solver", ["ilqr", "ddp"])
def test_solve_is_bit_identical_for_every_tail_threshold(solver, tmp_path):
    """... and whether or not the four-lanes-per-rollout kernels (rollout_quad.hip: the all-store search of few trajectories, the
    re-roll of the winners as its own launch) take part (ZOPT_AMD_ROLLOUT_QUAD=0: everything in rollout_ls_fast_kernel)."""
    res = {}
    # never / from the first iteration / switches when 40 of the 96 are left; the same without the quad kernels
    # (the last case also with the full instead of the packed Jacobians between the expansion and the sweep: ZOPT_AMD_JAC=full; the
    # cases without the quad kernels also with the dense instead of the sparse second derivatives: ZOPT_AMD_HES=dense)
    # (and one case with the 16-lanes-per-point expansion kernels instead of the one-lane-per-point ones: ZOPT_AMD_EXPAND=group; two with
    # the acceptance step copying the new trajectories instead of the solve alternating between two buffers: ZOPT_AMD_ILQR_SWAP=0)
    cases = [("0", "1", "packed"), ("1000000", "1", "packed"), ("40", "1", "packed"), ("0", "0", "packed"), ("1000000", "0", "packed"),
             ("40", "0", "full"), ("40", "1", "packed-group"), ("40", "1", "packed-copy"), ("0", "0", "packed-copy")]
    for thr, quad, jac in cases:
        out = tmp_path / f"{solver}_{thr}_{quad}_{jac}.npz"
        p = subprocess.run([sys.executable, "-c", CHILD, str(out), solver],
                           env=dict(LAB_ENVIRON, ZOPT_AMD_ILQR_TAIL=thr, ZOPT_AMD_ROLLOUT_QUAD=quad, ZOPT_AMD_JAC=jac.split("-")[0],
                                    ZOPT_AMD_EXPAND="group" if jac.endswith("group") else "points",
                                    ZOPT_AMD_ILQR_SWAP="0" if jac.endswith("copy") else "1",
                                    ZOPT_AMD_HES="dense" if quad == "0" else "sparse"), capture_output=True,
                           text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0 and "CHILD-OK" in p.stdout, (p.stdout[-300:], p.stderr[-1500:])
        res[(thr, quad, jac)] = dict(np.load(out))
    ref = res[("0", "0", "packed")]
    assert ref["c"].sum() > 48                           # the problem set mostly converges (a real solve, not a no-op)
    for case in cases[:3] + cases[4:]:
        for k in ("x", "u", "L", "J", "c"):
            assert np.array_equal(ref[k], res[case][k], equal_nan=True), (solver, case, k)


@pytest.mark.parametrize("N", [1, 2, 7])
def test_short_and_odd_horizons_agree_across_line_search_forms(N, tmp_path):
    """The four-lane kernels walk the horizon two steps per loop iteration (ping-pong operand registers) with a tail step for odd T:
    horizons 1, 2 and 7, all-store from the start and never, with and without the quad kernels -- the same bits."""
    res = {}
    cases = [("0", "0"), ("1000000", "1"), ("0", "1"), ("1000000", "0")]
    for thr, quad in cases:
        out = tmp_path / f"h{N}_{thr}_{quad}.npz"
        p = subprocess.run([sys.executable, "-c", CHILD, str(out), "ilqr", str(N)],
                           env=dict(LAB_ENVIRON, ZOPT_AMD_ILQR_TAIL=thr, ZOPT_AMD_ROLLOUT_QUAD=quad), capture_output=True, text=True,
                           timeout=600, cwd=ROOT)
        assert p.returncode == 0 and "CHILD-OK" in p.stdout, (p.stdout[-300:], p.stderr[-1500:])
        res[(thr, quad)] = dict(np.load(out))
    for case in cases[1:]:
        for k in ("x", "u", "L", "J", "c"):
            assert np.array_equal(res[cases[0]][k], res[case][k], equal_nan=True), (N, case, k)


@pytest.mark.parametrize("wind", ["still", "wind"])
def test_horizon_of_several_chunks_agrees_across_expansion_forms(wind, tmp_path):
    """The one-lane-per-point expansion kernels cut a trajectory into chunks of at most 32 points (T = 131: four chunks of 27 and one of 23); the solve
    equals the one with the 16-lanes-per-point kernels (ZOPT_AMD_EXPAND=group) bit for bit, second derivatives included (DDP), in
    still air and with wind (the wind forms use three-wave workgroups)."""
    res = {}
    for form in ("points", "group"):
        out = tmp_path / f"long_{wind}_{form}.npz"
        p = subprocess.run([sys.executable, "-c", CHILD, str(out), "ddp", "131", wind], env=dict(LAB_ENVIRON, ZOPT_AMD_EXPAND=form),
                           capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0 and "CHILD-OK" in p.stdout, (p.stdout[-300:], p.stderr[-1500:])
        res[form] = dict(np.load(out))
    assert res["points"]["c"].sum() > 0
    for k in ("x", "u", "L", "J", "c"):
        assert np.array_equal(res["points"][k], res["group"][k], equal_nan=True), (wind, k)


@pytest.mark.parametrize("solver", ["ilqr", "ddp"])
def test_windy_model_packed_and_full_operands_agree_and_match_the_oracle(solver, tmp_path):
    """A constant NED wind selects the wind forms of the generated derivatives (59 packed Jacobian entries, 85 sparse second derivatives)
    and the generic rollout kernel: the solve with packed / sparse operands equals the solve with the full matrices and dense rows bit
    for bit, and the iLQR result matches the oracle loop on the windy model."""
    import importlib
    res = {}
    for jac, hes in (("packed", "sparse"), ("full", "dense")):
        out = tmp_path / f"w_{solver}_{jac}.npz"
        p = subprocess.run([sys.executable, "-c", CHILD, str(out), solver, "12", "wind"],
                           env=dict(LAB_ENVIRON, ZOPT_AMD_JAC=jac, ZOPT_AMD_HES=hes), capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0 and "CHILD-OK" in p.stdout, (p.stdout[-300:], p.stderr[-1500:])
        res[jac] = dict(np.load(out))
    for k in ("x", "u", "L", "J", "c"):
        assert np.array_equal(res["packed"][k], res["full"][k], equal_nan=True), (solver, k)
    if solver == "ilqr":
        sys.path.insert(0, ROOT)
        zo = importlib.import_module("oracle.zopt_oracle")
        wind = np.array([3.0, 1.0, -0.5])
        step = lambda x, u: x + 0.1 * zo.quad_inertialDynamics(x, u, wind_ned=wind)
        rng = np.random.default_rng(11)
        x0 = np.zeros((96, 12)); x0[:, 9:12] = rng.uniform(-10, 10, (96, 3))
        ug = np.tile(np.array([9.807, 0.0, 0.0, 0.0]), (12, 1))
        for i in (0, 5):
            rt, rL, rJ, rc = zo.iterativeLqr(step, np.eye(12), np.eye(4), 10 * np.eye(12), x0[i], ug)
            assert bool(res["packed"]["c"][i]) == rc and abs(res["packed"]["J"][i] - rJ) <= 1e-7 * abs(rJ)
            assert np.max(np.abs(res["packed"]["u"][i] - rt.uTraj)) <= 1e-6 * max(1.0, np.abs(rt.uTraj).max())
