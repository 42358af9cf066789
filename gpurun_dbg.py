import numpy as np, sys
sys.path.insert(0, '/root/repo')
from zopt_amd import ilqrUtils
rng = np.random.default_rng(0)
for k in (1, 2, 3, 4, 8, 16):
    M = rng.standard_normal((2, k, k)); A = M + np.swapaxes(M, -1, -2)
    P = ilqrUtils.ensurePositiveDefinite(A)
    w, v = np.linalg.eigh(A); R = (v * np.maximum(w, 1e-3)[:, None, :]) @ np.swapaxes(v, -1, -2)
    print(k, np.max(np.abs(P - R)))
    if k <= 3: print(P[0], R[0])
