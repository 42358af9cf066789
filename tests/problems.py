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


def random_ilqr_model(batch: int, T: int, n: int, m: int, seed: int = 0):
    """Random quadratic model along a trajectory for the iLQR backward pass: (AffineDynamics, QuadraticCost, Vf) fields.

    f_x ~ contraction-ish, f_u ~ N(0,1); stacked cost Hessian H = M M^T/(n+m) + I (positive definite); random
    gradients; terminal v_xx = M M^T/n + I."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((batch, T, n))
    f_x = rng.standard_normal((batch, T, n, n)) * (0.9 / np.sqrt(n))
    f_u = rng.standard_normal((batch, T, n, m))
    M = rng.standard_normal((batch, T, n + m, n + m))
    H = M @ np.swapaxes(M, -1, -2) / (n + m) + np.eye(n + m)
    c = rng.standard_normal((batch, T))
    c_x = rng.standard_normal((batch, T, n))
    c_u = rng.standard_normal((batch, T, m))
    c_xx = np.ascontiguousarray(H[..., :n, :n])
    c_ux = np.ascontiguousarray(H[..., n:, :n])
    c_uu = np.ascontiguousarray(H[..., n:, n:])
    Mv = rng.standard_normal((batch, n, n))
    v_xx = Mv @ np.swapaxes(Mv, -1, -2) / n + np.eye(n)
    v_x = rng.standard_normal((batch, n))
    v = rng.standard_normal(batch)
    return (f, f_x, f_u), (c, c_x, c_u, c_xx, c_ux, c_uu), (v, v_x, v_xx)
