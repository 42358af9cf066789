#!/usr/bin/env python3
"""One-off robustness sweep of zopt_amd.lqrUtils.infiniteHorizonLqr against SciPy's solve_continuous_are over random designs
(stable / unstable A, 1 <= m <= n <= 16, several scales).  Prints failure counts and the error distribution."""
import os
import sys

import numpy as np
import scipy.linalg as spl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zopt_amd import lqrUtils  # noqa: E402

rng = np.random.default_rng(0)
errs, res_ratio, fails, scipy_fails = [], [], 0, 0
for n in range(1, 17):
    for m in sorted({1, max(1, n // 4), max(1, n // 2), n}):
        for scale in (0.1, 1.0, 5.0):
            batch = 8
            A = scale * rng.standard_normal((batch, n, n))
            B = rng.standard_normal((batch, n, m))
            M = rng.standard_normal((batch, n, n))
            Q = M @ np.swapaxes(M, -1, -2) / n + 0.1 * np.eye(n)
            M = rng.standard_normal((batch, m, m))
            R = M @ np.swapaxes(M, -1, -2) / m + 0.5 * np.eye(m)
            try:
                K, P, it = lqrUtils.infiniteHorizonLqr(A, B, Q, R, return_value=True)
            except np.linalg.LinAlgError:
                fails += 1
                continue
            for b in range(batch):
                try:
                    Pr = spl.solve_continuous_are(A[b], B[b], Q[b], R[b])
                except Exception:
                    scipy_fails += 1
                    continue
                G = B[b] @ np.linalg.solve(R[b], B[b].T)
                res = lambda X: np.max(np.abs(A[b].T @ X + X @ A[b] - X @ G @ X + Q[b])) / np.max(np.abs(X))
                errs.append(np.max(np.abs(P[b] - Pr)) / np.max(np.abs(Pr)))
                res_ratio.append(res(P[b]) / max(res(Pr), 1e-16))
errs, res_ratio = np.array(errs), np.array(res_ratio)
print("designs", len(errs), "batches rejected by the GPU solver", fails, "scipy failures", scipy_fails)
print("relative |P - P_scipy|: median %.1e  99%% %.1e  max %.1e" % (np.median(errs), np.quantile(errs, 0.99), errs.max()))
print("residual ratio (ours / scipy): median %.2f  99%% %.1f  max %.1f" % (np.median(res_ratio), np.quantile(res_ratio, 0.99), res_ratio.max()))
print("share with error > 1e-9:", float((errs > 1e-9).mean()))
