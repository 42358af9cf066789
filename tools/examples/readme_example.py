import numpy as np, torch, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from zopt_amd.ilqrUtils import iterativeLqr
from zopt_amd import models
Q, R = np.eye(12), np.eye(4)
cost = models.QuadraticCost(Q, R, 10 * Q)
x0 = np.zeros((4, 12)); x0[:, 9:12] = 1.0
uGuess = np.tile(models.QuadcopterEuler.uTrim, (4, 30, 1))
traj, L, J, converged = iterativeLqr(models.QuadcopterEuler(0.1), cost, cost, x0, uGuess)
print(J, converged)
Qt = torch.eye(2, dtype=torch.float64, device="cuda")
dyn = lambda x, u: x + 0.1 * torch.stack([x[1], -torch.sin(x[0]) + u[0]])
traj, L, J, converged = iterativeLqr(dyn, lambda x, u: x @ Qt @ x + u @ u, lambda x: 10 * x @ Qt @ x, np.array([3.0, 0.0]), np.zeros((50, 1)))
print(J, converged, traj.xTraj.shape)
