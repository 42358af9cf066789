#!/usr/bin/env python3
"""Secondary benchmark (not the driver's bench.py line): BASELINE configs[2] -- lqrMpc on the quadcopter linearised at
hover (demos/lqrMpc.py:11-32: dt = 0.1, Q = R = I, the demo's bounds), N = 30, `batch` independent instances with
x0[9:12] ~ U(-10,10)^3 and small velocities / angles.  Reports wall time per batched solve, ADMM iterations and status mix."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--N", type=int, default=30)
    ap.add_argument("--eps", type=float, nargs="+", default=[1e-2, 1e-4])
    ap.add_argument("--max-iter", type=int, default=100000)
    ap.add_argument("--rh-steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    from zopt_amd import models, mpcUtils, pytrees
    dt = 0.1
    lin = pytrees.AffineDynamics.from_function(models.QuadcopterEuler(dt), np.zeros(12), models.QuadcopterEuler.uTrim)
    A, B = np.asarray(lin.f_x), np.asarray(lin.f_u)            # I + dt*Aw, dt*Bw   (demos/lqrMpc.py:26-28)
    x_ub = np.array([1, 1, 1, 0.3, 0.3, 0.1, 0.5, 0.5, np.inf, np.inf, np.inf, np.inf])
    u_ub = np.array([3.0, 3, 3, 3])
    prob = mpcUtils.lqrMpc(A, B, np.eye(12), np.eye(4), args.N, -x_ub, x_ub, -u_ub, u_ub)
    rng = np.random.default_rng(1)
    x0 = np.clip(0.03 * rng.standard_normal((args.batch, 12)), -x_ub + 1e-6, x_ub - 1e-6)
    x0[:, 9:12] = rng.uniform(-10, 10, (args.batch, 3))
    tx0 = torch.as_tensor(x0, device="cuda")
    prob.solve(tx0[:64], eps_abs=1e-2, eps_rel=1e-2)            # warm-up (tables, allocator)
    torch.cuda.synchronize()
    for eps in args.eps:
        t0 = time.perf_counter()
        u0, traj, status = prob.solve(tx0, solver="OSQP", eps_abs=eps, eps_rel=eps, max_iter=args.max_iter, warm_start=False)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        its = prob.last_iterations
        print(json.dumps({"workload": f"lqrMpc quadcopter n=12 m=4 N={args.N}, {args.batch} instances, eps={eps:g}",
                          "solve_ms": t * 1e3, "optimal_frac": float(np.mean(status == "optimal")),
                          "iters_mean": float(its.mean()), "iters_max": int(its.max()),
                          "instance_horizon_steps_per_s": args.batch * args.N / t,
                          "admm_sweep_steps_per_s": float(its.sum()) * args.N / t}))


    # receding-horizon run (50 MPC steps, demos/lqrMpc.py:40-47: clip, solve, x <- xTraj[1]), warm-started
    lo = torch.as_tensor(-x_ub + 1e-6, device="cuda")
    hi = torch.as_tensor(x_ub - 1e-6, device="cuda")
    for warm in ((False, "shift") if args.rh_steps > 0 else ()):
        x = tx0.clone()
        alive = torch.ones(args.batch, dtype=torch.bool, device="cuda")
        iters = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.rh_steps):
            x = torch.minimum(torch.maximum(x, lo), hi)
            u0, traj, status = prob.solve(x, solver="OSQP", eps_abs=1e-2, eps_rel=1e-2, eps_prim_inf=1e-3, max_iter=4000,
                                          warm_start=warm)
            ok = torch.as_tensor(status == "optimal", device="cuda")
            alive &= ok
            iters += float(prob.last_iterations.sum())
            # instances whose QP became infeasible / unsolved are parked at hover so that they stop costing iterations
            x = torch.where(alive[:, None], traj.xTraj[:, 1], torch.zeros_like(x))
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(json.dumps({"workload": f"receding horizon: {args.rh_steps} MPC steps x {args.batch} instances, N={args.N}, "
                                      f"eps=1e-2, warm_start={warm}",
                          "total_ms": t * 1e3, "ms_per_mpc_step": t * 1e3 / args.rh_steps,
                          "alive_frac": float(alive.double().mean().item()),
                          "admm_iters_per_solve": iters / (args.rh_steps * args.batch),
                          "instance_solves_per_s": args.rh_steps * args.batch / t}))


if __name__ == "__main__":
    main()
