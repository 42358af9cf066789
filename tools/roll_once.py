import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import secondary_bench as sb
from zopt_amd import ilqrUtils, pytrees as pt
model, cost, x0, ug = sb.config3_problem()
tx0, tug = torch.as_tensor(x0, device="cuda"), torch.as_tensor(ug, device="cuda")
rng = np.random.default_rng(0)
B, N = x0.shape[0], ug.shape[1]
pol = pt.AffinePolicy(torch.as_tensor(0.05 * rng.standard_normal((B, N, 4)), device="cuda"), torch.as_tensor(0.02 * rng.standard_normal((B, N, 4, 12)), device="cuda"))
xp = torch.zeros((B, N + 1, 12), dtype=torch.float64, device="cuda"); xp[:, :, :] = tx0[:, None, :]
prev = pt.Trajectory(xp, tug)
for _ in range(3):
    t, J = ilqrUtils.forwardPass2(tx0, model, cost, pol, prev)
torch.cuda.synchronize()
