#!/usr/bin/env python3
"""Secondary benchmark (not the driver's bench.py line): BASELINE configs[3] -- iterativeLqr on the quadcopter,
T=100, dt=0.1, demo weights (demos/iterativeLqr.py:22-39), x0[9:12] ~ U(-10,10)^3, uGuess = uTrim, seed 2.
Reports wall time per solve, iLQR iterations, horizon-steps/s (batch*T*iterations / time) and a per-kernel split."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ddp", action="store_true", help="differentialDynamicProgramming (R = 0.2 I as in its demo) instead of iterativeLqr")
    ap.add_argument("--max-iter", type=int, default=100, help="maxIter of the solve (a few iterations: every launch at the full batch, for PMC runs)")
    ap.add_argument("--no-warmup", action="store_true", help="skip the 64-trajectory warm-up solve (PMC means over one kernel's dispatches)")
    args = ap.parse_args()
    import torch
    from zopt_amd import ilqrUtils, models
    rng = np.random.default_rng(2)
    x0 = np.zeros((args.batch, 12))
    x0[:, 9:12] = rng.uniform(-10, 10, (args.batch, 3))
    ug = np.tile(models.QuadcopterEuler.uTrim, (args.batch, args.T, 1))
    cost = models.QuadraticCost(np.eye(12), (0.2 if args.ddp else 1.0) * np.eye(4), 10 * np.eye(12))
    solve = ilqrUtils.differentialDynamicProgramming if args.ddp else ilqrUtils.iterativeLqr
    model = models.QuadcopterEuler(0.1)
    tx0, tug = torch.as_tensor(x0, device="cuda"), torch.as_tensor(ug, device="cuda")
    if not args.no_warmup:
        solve(model, cost, cost, tx0[:64], tug[:64])       # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        traj, L, J, conv = solve(model, cost, cost, tx0, tug, maxIter=args.max_iter)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = min(times)
    print(json.dumps({"workload": f"{'differentialDynamicProgramming' if args.ddp else 'iterativeLqr'} quadcopter n=12 m=4 T={args.T} batch={args.batch} fp64",
                      "solve_ms": t * 1e3, "converged_frac": float(conv.double().mean().item()),
                      "J_mean_finite": float(J[torch.isfinite(J)].mean().item()), "J_nonfinite_frac": float((~torch.isfinite(J)).double().mean().item()),
                      "trajectories_per_s": args.batch / t}))


if __name__ == "__main__":
    main()
