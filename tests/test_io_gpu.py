"""F4 round trip from real GPU solves: iterativeLqr and a short receding-horizon lqrMpc run -> zopt_amd.io.save ->
load on the CPU side -> the arrays the reference's plot functions take (plottingTools.plotTimeTrajectory: tArr (N,), xArr (N, nx),
plottingTools.py:5-40; mpcUtils.plotMpcTrajectory: traj (N_t, N_mpc, n), mpcUtils.py:84-122).  The plots themselves are the
reference's presentation code (out of scope); what is checked is that GPU results arrive in exactly their input contract."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ilqr_and_mpc_results_round_trip_into_plot_inputs(tmp_path):
    import torch
    assert torch.cuda.is_available()
    from tools import secondary_bench as sb
    from zopt_amd import ilqrUtils, io as zio, models
    # iLQR: 4 quadcopter problems, T = 30, torch tensors on the GPU in -> results stay on the GPU until io.save
    N, dt = 30, 0.1
    model, cost, x0, ug = sb.config3_problem(batch=4, T=N)
    traj, L, J, conv = ilqrUtils.iterativeLqr(model, cost, cost, torch.as_tensor(x0, device="cuda"), torch.as_tensor(ug, device="cuda"))
    assert traj.xTraj.is_cuda
    path = zio.save(tmp_path / "ilqr.npz", "ilqr", dict(x0=x0, uGuess=ug, Q=cost.Q, R=cost.R, Qf=cost.Qf, dt=dt, model="quadcopter"),
                    dict(xTraj=traj.xTraj, uTraj=traj.uTraj, L=L, J=J, converged=conv))
    kind, prob, res = zio.load(path)
    assert kind == "ilqr" and res["xTraj"].shape == (4, N + 1, 12) and res["converged"].dtype == bool
    tArr, xArr = zio.time_trajectory_inputs(res, float(prob["dt"]), "xTraj", index=1)
    assert tArr.shape == (N + 1,) and xArr.shape == (N + 1, 12) and np.array_equal(tArr, res["tArr"])
    assert np.array_equal(xArr, traj.xTraj[1].cpu().numpy()) and np.array_equal(xArr[0], x0[1])
    tU, uArr = zio.time_trajectory_inputs(traj, dt, "uTraj", index=1)        # straight from the GPU Trajectory, no file
    assert tU.shape == (N,) and uArr.shape == (N, 4)
    # MPC: 6 closed-loop steps of the demo loop (clip -> solve -> x <- xTraj[1]; demos/lqrMpc.py:40-47) on 8 instances
    prob_mpc, x_ub = sb.mpc_problem(N=30)
    x = torch.as_tensor(sb.mpc_x0(8, x_ub), device="cuda")
    lo, hi = torch.as_tensor(-x_ub + 1e-6, device="cuda"), torch.as_tensor(x_ub - 1e-6, device="cuda")
    steps = []
    for _ in range(6):
        x = torch.minimum(torch.maximum(x, lo), hi)
        u0, tr, status = prob_mpc.solve(x, solver="OSQP", eps_abs=1e-2, eps_rel=1e-2, eps_prim_inf=1e-3)
        steps.append(tr.xTraj)
        x = tr.xTraj[:, 1]
    arr = zio.mpc_trajectory_array(steps, index=5)
    assert arr.shape == (6, 31, 12) and arr.dtype == np.float64 and np.all(np.isfinite(arr))
    # consecutive predictions chain as the demo assumes ("perfect tracking"): step i+1 starts at step i's second state (clipped)
    assert np.allclose(arr[1:, 0], np.clip(arr[:-1, 1], -x_ub + 1e-6, x_ub - 1e-6), rtol=0, atol=1e-12)
    p2 = zio.save(tmp_path / "mpc.npz", "mpc", dict(A=prob_mpc.A, B=prob_mpc.B, Q=prob_mpc.Q, R=prob_mpc.R, N=30, x_lb=-x_ub, x_ub=x_ub,
                                                    u_lb=prob_mpc.u_lb, u_ub=prob_mpc.u_ub, x0=steps[-1][:, 0]),
                  dict(xTraj=steps[-1], uTraj=tr.uTraj, status=status))
    kind, _, r3 = zio.load(p2)
    assert kind == "mpc" and r3["xTraj"].shape == (8, 31, 12) and set(r3["status"]) <= {"optimal", "infeasible", "user_limit"}
