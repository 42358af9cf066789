"""Seeded synthetic problem generators shared by tests, bench.py and the golden script.

SURVEY.md section 8(d) "Synthetic inputs per config".  NumPy only (no reference, no oracle).
"""
from __future__ import annotations

import numpy as np


def random_lti_systems(batch: int, n: int, m: int, seed: int = 0, rho: float = 0.95, dtype=np.float64):
    """Per system: G~N(0,1), A = rho*G/spectral_radius(G); B~N(0,1); Q = MM^T/n + I; R = NN^T/m + I."""
    rng = np.random.default_rng(seed)
    A = np.empty((batch, n, n))
    B = np.empty((batch, n, m))
    Q = np.empty((batch, n, n))
    R = np.empty((batch, m, m))
    for i in range(batch):
        G = rng.standard_normal((n, n))
        A[i] = rho * G / np.max(np.abs(np.linalg.eigvals(G)))
        B[i] = rng.standard_normal((n, m))
        M = rng.standard_normal((n, n))
        N = rng.standard_normal((m, m))
        Q[i] = M @ M.T / n + np.eye(n)
        R[i] = N @ N.T / m + np.eye(m)
    return A.astype(dtype), B.astype(dtype), Q.astype(dtype), R.astype(dtype)


def tile_over_horizon(A, B, Q, R, T: int):
    """Materialise (b,T,.,.) time-varying tensors from per-system LTI matrices (the layout the
    reference API takes: demos/discreteFiniteHorizonLqr.py:30-34 tiles LTI matrices over T)."""
    def rep(X):
        return np.ascontiguousarray(np.repeat(X[:, None], T, axis=1))
    return rep(A), rep(B), rep(Q), rep(R)


def random_time_varying(batch: int, T: int, n: int, m: int, seed: int = 0, dtype=np.float64):
    """Fully time-varying well-conditioned problems: every (trajectory, step) has its own matrices."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((batch, T, n, n)) * (0.9 / np.sqrt(n))
    B = rng.standard_normal((batch, T, n, m))
    M = rng.standard_normal((batch, T, n, n))
    N = rng.standard_normal((batch, T, m, m))
    Q = M @ np.swapaxes(M, -1, -2) / n + np.eye(n)
    R = N @ np.swapaxes(N, -1, -2) / m + np.eye(m)
    return A.astype(dtype), B.astype(dtype), Q.astype(dtype), R.astype(dtype)
