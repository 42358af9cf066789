import numpy as np, sys, os
sys.path.insert(0, '/root/repo')
from oracle import zopt_oracle as zo
from tests import problems
from zopt_amd import lqrUtils
for (n, m, T) in [(16, 16, 1), (16, 16, 2), (32, 16, 1), (48, 16, 1)]:
    A, B, Q, R = problems.random_time_varying(1, T, n, m, seed=1, dtype=np.float32)
    Lg = lqrUtils.discreteFiniteHorizonLqr(A, B, Q, R, T)
    L64 = zo.discreteFiniteHorizonLqr(*(x.astype(np.float64) for x in (A, B, Q, R)), T)
    err = np.abs(Lg - L64)[0]
    print(n, m, T, "max err", err.max(), "per-column max (last step):", np.round(err[-1].max(axis=0), 4), "per-row:", np.round(err[-1].max(axis=1), 4))
