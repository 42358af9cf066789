#!/usr/bin/env python3
"""NumPy model of zopt_amd/csrc/ns16.h: the PD projection  a -> V max(w, eps) V^T  (reference ilqrUtils.py:217-225) computed
without an eigen-decomposition, as  eps I + (X + sign(X) X) / 2,  X = a - eps I,  with the same arithmetic order the kernel uses
(products as X^T Y, symmetrised updates, Frobenius scaling, the (quintic, cubic) pair rule and iteration caps).

Prints the worst relative error against numpy.linalg.eigh over adversarial spectra and the number of 16x16x4 MFMAs a
full 16 x 16 tile needs (4 per product).  CPU only."""
import numpy as np

QA, QB, QC = 3.4445, -4.7750, 2.0315
MAX_PAIRS, MAX_CUBIC = 18, 14


def eigh_projection(a, eps=1e-3):
    w, v = np.linalg.eigh(a)
    return (v * np.maximum(w, eps)) @ v.T


def ns_projection(A, k, eps=1e-3, stats=None):
    """A: 16x16 with the symmetric matrix in its leading k x k block (zeros elsewhere)."""
    sym = lambda M: 0.5 * (M + M.T)
    I = np.zeros((16, 16))
    I[:k, :k] = np.eye(k)
    A = sym(A)
    # indices with an exactly zero row/column: eigenvalue 0, decoupled -- they get eps on the diagonal and stay out of the iteration
    Ir = I * (np.abs(A).sum(0) != 0)[None, :]
    I_full, I = I, Ir
    X = A - eps * I
    s = np.linalg.norm(X)
    Z = X / s if s > 0 else X
    pairs = cubic = mfma = 0
    while s > 0:
        Z2 = Z.T @ Z
        mfma += 4
        F = np.sum((I - Z2) ** 2)
        if F > 0.9 and pairs < MAX_PAIRS:
            if F > 0.9:      # (always, in a booster group) a second quintic rides along: one reduction and transpose per 8 products
                Z4 = Z2.T @ Z2
                Z = (QA * I + QB * Z2 + QC * Z4).T @ Z
                Z2 = Z.T @ Z
                mfma += 12
            Z4 = Z2.T @ Z2
            Z = (QA * I + QB * Z2 + QC * Z4).T @ Z        # W bitwise symmetric: W^T Z = W Z; symmetrised after the cubic step
            Z2 = Z.T @ Z
            Z = sym((1.5 * I - 0.5 * Z2).T @ Z)
            mfma += 16
            pairs += 1
        else:
            # cubic steps in twos (one reduction / one symmetrisation per two steps); F' = 0.5625 F^2 after a step
            if F < 1e-16 or cubic + 1 >= MAX_CUBIC:
                Z = sym(Z.T @ (1.5 * I - 0.5 * Z2))
                mfma += 4
                cubic += 1
                break
            Z = Z.T @ (1.5 * I - 0.5 * Z2)
            Z2 = Z.T @ Z
            Z = sym(Z.T @ (1.5 * I - 0.5 * Z2))
            mfma += 12
            cubic += 2
            if 0.5625 * F * F < 1e-18 or cubic >= MAX_CUBIC:
                break
    if stats is not None:
        stats.append((pairs, cubic, mfma + 4))
    return sym(eps * I_full + 0.5 * (X + Z.T @ X))


def spectrum(kind, k, rng):
    if kind == 0:
        return rng.standard_normal(k) * 10 ** rng.uniform(-3, 3)
    if kind == 1:   # half the eigenvalues within 1e-9 of eps
        return np.concatenate([rng.standard_normal(k // 2), 1e-3 + rng.standard_normal(k - k // 2) * 1e-9])
    if kind == 2:   # 16 decades of magnitude, random signs
        return 10.0 ** rng.uniform(-14, 2, k) * rng.choice([-1, 1], k)
    if kind == 3:   # rank one (after rotation: dense, eigenvalue 0 with multiplicity k - 1)
        lam = np.zeros(k)
        lam[0] = rng.standard_normal()
        return lam
    if kind == 4:   # two dominant eigenvalues
        lam = rng.standard_normal(k)
        lam[:2] = 1e3
        return lam
    return 1e-3 + 10.0 ** rng.uniform(-16, -2, k) * rng.choice([-1, 1], k)   # everything hugging eps


def main():
    rng = np.random.default_rng(1)
    worst, stats = [0.0] * 6, []
    for trial in range(900):
        k, kind = int(rng.integers(2, 17)), trial % 6
        Q, _ = np.linalg.qr(rng.standard_normal((k, k)))
        a = (Q * spectrum(kind, k, rng)) @ Q.T
        a = 0.5 * (a + a.T)
        A = np.zeros((16, 16))
        A[:k, :k] = a
        P, R = ns_projection(A, k, stats=stats), eigh_projection(a)
        worst[kind] = max(worst[kind], np.abs(P[:k, :k] - R).max() / max(np.abs(R).max(), 1e-300))
    st = np.array(stats)
    print("adversarial spectra: worst relative error per kind", ["%.1e" % w for w in worst])
    print("  (pairs, cubic steps, MFMAs) mean", st.mean(0).round(1), "max", st.max(0))
    worst, stats = 0.0, []
    for _ in range(400):
        k = int(rng.integers(4, 17))
        M = rng.standard_normal((k, k))
        a = 0.5 * (M + M.T) * 10 ** rng.uniform(-1, 2)
        A = np.zeros((16, 16))
        A[:k, :k] = a
        worst = max(worst, np.abs(ns_projection(A, k, stats=stats)[:k, :k] - eigh_projection(a)).max() / np.abs(eigh_projection(a)).max())
    st = np.array(stats)
    print("dense random symmetric: worst relative error %.1e" % worst)
    print("  (pairs, cubic steps, MFMAs) mean", st.mean(0).round(1), "max", st.max(0))


if __name__ == "__main__":
    main()
