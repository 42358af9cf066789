#!/usr/bin/env python3
"""Model of a scaled Newton-Schulz (Chen-Chow) schedule for the PD projection, against tools/ns_psd_model.py (the shipped iteration):
products / reductions per matrix and accuracy against eigh on adversarial spectra, on matrices dumped from a DDP solve (/tmp/ddp_mats.npy,
if present) and on dense random matrices.  DESIGN.md section 7 quotes its output; not part of the product."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import ns_psd_model as M

def schedule(l0=1e-13):
    """universal Chen-Chow schedule: (l_k, a_k, b_k) with Z <- Z (a I - b Z^2)"""
    out = []
    l = l0
    while 1 - l > 1e-9:
        rho2 = 3.0 / (1 + l + l * l); rho = np.sqrt(rho2)
        a, b = 1.5 * rho, 0.5 * rho * rho2
        out.append((l, a, b))
        l = 0.5 * rho * l * (3 - rho2 * l * l)
    out.append((l, 1.5, 0.5))      # one plain step: 1 - l^2 ~ 2e-9 -> 1e-17
    return out
SCHED = schedule()
LS = np.array([s[0] for s in SCHED])

def proj(A, k, eps=1e-3, stats=None, safety=1.0, sym_every=2):
    sym = lambda X: 0.5 * (X + X.T)
    I = np.zeros((16, 16)); I[:k, :k] = np.eye(k)
    A = sym(A)
    Ir = I * (np.abs(A).sum(0) != 0)[None, :]
    I_full, I = I, Ir
    X = A - eps * I
    s = np.linalg.norm(X)
    Z = X / s if s > 0 else X
    prods = checks = steps = 0
    if s > 0:
        l0 = min(eps / s * safety, 1.0)
        k0 = max(int(np.searchsorted(LS, l0, side='right')) - 1, 0)   # largest k with l_k <= l0
        n = 0
        for (l, a, b) in SCHED[k0:]:
            Z2 = Z.T @ Z
            Z = (a * I - b * Z2).T @ Z
            prods += 2; steps += 1; n += 1
            if n % sym_every == 0:
                Z = sym(Z)
        Z = sym(Z)
        # final check + fallback: plain adaptive loop of the product
        Z2 = Z.T @ Z; prods += 1
        F = np.sum((I - Z2) ** 2); checks += 1
        fb = 0
        while F > 1e-16 and fb < 40:
            # fallback: the product's groups / cubic pairs (modelled as: QQC group if F > 0.9 else two cubic steps)
            if F > 0.9:
                for _ in range(2):
                    Z4 = Z2.T @ Z2
                    Z = (M.QA * I + M.QB * Z2 + M.QC * Z4).T @ Z
                    Z2 = Z.T @ Z; prods += 3
                Z = sym((1.5 * I - 0.5 * Z2).T @ Z); prods += 1
            else:
                Z = Z.T @ (1.5 * I - 0.5 * Z2); Z2 = Z.T @ Z
                Z = sym(Z.T @ (1.5 * I - 0.5 * Z2)); prods += 3
            Z2 = Z.T @ Z; prods += 1
            F = np.sum((I - Z2) ** 2); checks += 1; fb += 1
    if stats is not None:
        stats.append((steps, prods + 1, checks))
    return sym(eps * I_full + 0.5 * (X + Z.T @ X))

def run(name, **kw):
    rng = np.random.default_rng(1)
    worst, stats = [0.0] * 6, []
    for trial in range(600):
        k, kind = int(rng.integers(2, 17)), trial % 6
        Q, _ = np.linalg.qr(rng.standard_normal((k, k)))
        a = (Q * M.spectrum(kind, k, rng)) @ Q.T
        a = 0.5 * (a + a.T)
        A = np.zeros((16, 16)); A[:k, :k] = a
        P, R = proj(A, k, stats=stats, **kw), M.eigh_projection(a)
        worst[kind] = max(worst[kind], np.abs(P[:k, :k] - R).max() / max(np.abs(R).max(), 1e-300))
    st = np.array(stats)
    print(name, "adv worst", ["%.1e" % w for w in worst], "(steps,products,checks)", st.mean(0).round(2), "max", st.max(0))
    if __import__('os').path.exists('/tmp/ddp_mats.npy'):
        mats = np.load('/tmp/ddp_mats.npy')
        stats, worst = [], 0.0
        for a in mats[::3]:
            P, R = proj(a, 16, stats=stats, **kw), M.eigh_projection(a)
            worst = max(worst, np.abs(P - R).max() / np.abs(R).max())
        st = np.array(stats)
        print(name, "DDP worst %.1e" % worst, "(steps,products,checks)", st.mean(0).round(2), "max", st.max(0))
    rng = np.random.default_rng(5); stats=[]; worst=0
    for _ in range(200):
        a = rng.standard_normal((16,16)); a = a + a.T
        P, R = proj(a, 16, stats=stats, **kw), M.eigh_projection(a)
        worst = max(worst, np.abs(P - R).max() / np.abs(R).max())
    st = np.array(stats)
    print(name, "dense random worst %.1e" % worst, "(steps,products,checks)", st.mean(0).round(2))
print(len(SCHED), "schedule entries")
run("CC sym2       ")
run("CC sym1       ", sym_every=1)
run("CC sym2 saf.5 ", safety=0.5)
