"""CPU model of the MFMA tile algebra the Riccati kernels are written in (lqr_backward_tiled_f32.hip).

The kernels never move data between lanes for their matrix products: they rely on the fact that, with the hardware's operand /
result lane layouts, `acc + X^T Y` of two tiles held in the accumulator layout ("D layout") is 4 MFMA instructions whose operands
are the tile registers themselves.  This file restates the documented lane semantics of `v_mfma_f32_16x16x4_f32` (A: lane l
supplies A[l & 15][l >> 4]; B: lane l supplies B[l >> 4][l & 15]; D: lane l, register r holds D[4 (l >> 4) + r][l & 15]) and of
`v_mfma_f64_16x16x4_f64` (D register r holds D[4 r + (l >> 4)][l & 15]) in NumPy and checks, on random data:
  * the closure property op(X, Y) = X^T Y for both layouts,
  * one full Riccati step of the fp32 tile kernel (same sequence of tile products and LDS transposes as the HIP code) against the
    plain formulas of zopt/lqrUtils.py:167-170,
  * the four-block layout of v_mfma_f64_4x4x4_4b probed on the hardware (profiles/r01_probe_mfma_4x4x4.txt).
"""
import numpy as np


# ---- instruction models: each takes per-lane operand vectors (64,) and a per-lane accumulator (64, 4) ----------------------
def mfma_f32_16x16x4(a, b, c):
    """D = A(16x4) B(4x16) + C; lane l: a -> A[l&15][l>>4], b -> B[l>>4][l&15], c[l][r] -> C[4(l>>4)+r][l&15]."""
    lane = np.arange(64)
    A = np.zeros((16, 4)); B = np.zeros((4, 16)); C = np.zeros((16, 16))
    A[lane & 15, lane >> 4] = a
    B[lane >> 4, lane & 15] = b
    for r in range(4):
        C[4 * (lane >> 4) + r, lane & 15] = c[:, r]
    D = A @ B + C
    out = np.empty((64, 4))
    for r in range(4):
        out[:, r] = D[4 * (lane >> 4) + r, lane & 15]
    return out


def mfma_f64_16x16x4(a, b, c):
    """Same A / B lanes; accumulator register r of lane l holds C[4 r + (l >> 4)][l & 15]."""
    lane = np.arange(64)
    A = np.zeros((16, 4)); B = np.zeros((4, 16)); C = np.zeros((16, 16))
    A[lane & 15, lane >> 4] = a
    B[lane >> 4, lane & 15] = b
    for r in range(4):
        C[4 * r + (lane >> 4), lane & 15] = c[:, r]
    D = A @ B + C
    out = np.empty((64, 4))
    for r in range(4):
        out[:, r] = D[4 * r + (lane >> 4), lane & 15]
    return out


def tile_f32(X):      # 16x16 matrix -> D layout of the fp32 instruction: t[l][r] = X[4(l>>4)+r][l&15]
    lane = np.arange(64)
    return np.stack([X[4 * (lane >> 4) + r, lane & 15] for r in range(4)], axis=1)


def untile_f32(t):
    lane = np.arange(64)
    X = np.empty((16, 16))
    for r in range(4):
        X[4 * (lane >> 4) + r, lane & 15] = t[:, r]
    return X


def tile_f64(X):      # fp64 instruction: t[l][r] = X[4 r + (l >> 4)][l & 15]
    lane = np.arange(64)
    return np.stack([X[4 * r + (lane >> 4), lane & 15] for r in range(4)], axis=1)


def untile_f64(t):
    lane = np.arange(64)
    X = np.empty((16, 16))
    for r in range(4):
        X[4 * r + (lane >> 4), lane & 15] = t[:, r]
    return X


def op(mfma, x, y, acc):
    """acc + X^T Y: four MFMAs, operands = the tile registers themselves (register s is K-step s)."""
    for s in range(4):
        acc = mfma(x[:, s], y[:, s], acc)
    return acc


def test_closure_property_of_both_layouts():
    rng = np.random.default_rng(0)
    X, Y, C = rng.standard_normal((3, 16, 16))
    got32 = untile_f32(op(mfma_f32_16x16x4, tile_f32(X), tile_f32(Y), tile_f32(C)))
    got64 = untile_f64(op(mfma_f64_16x16x4, tile_f64(X), tile_f64(Y), tile_f64(C)))
    assert np.max(np.abs(got32 - (X.T @ Y + C))) <= 1e-13
    assert np.max(np.abs(got64 - (X.T @ Y + C))) <= 1e-13


def _tiles(M, rt, ct):     # (16 rt x 16 ct) matrix -> [rt][ct] D-layout tiles
    return [[tile_f32(M[16 * i:16 * i + 16, 16 * j:16 * j + 16]) for j in range(ct)] for i in range(rt)]


def _untiles(T):
    return np.block([[untile_f32(t) for t in row] for row in T])


def test_tiled_riccati_step_matches_the_reference_formulas():
    """The product sequence of lqr_backward_tiled_f32.hip at NT = 2 (n = 32, m = 16), in fp64 arithmetic of the model:
    Y_B, S, Y_A, L (here: numpy solve), Acl / -RL / W via LDS-transposed operands, V'."""
    rng = np.random.default_rng(1)
    NT, n, m = 2, 32, 16
    A = rng.standard_normal((n, n)) * (0.9 / np.sqrt(n))
    B = rng.standard_normal((n, m))
    Q = rng.standard_normal((n, n))                 # nonsymmetric on purpose: the kernel must be exact for it too
    R = rng.standard_normal((m, m)) + 4 * np.eye(m)
    V = rng.standard_normal((n, n)) + 4 * np.eye(n)
    mf = mfma_f32_16x16x4
    zero = np.zeros((64, 4))
    Vt, Ft, Qt = _tiles(V, NT, NT), _tiles(np.hstack([A, B]), NT, NT + 1), _tiles(Q, NT, NT)
    Rt = tile_f32(R)
    # Y_B = V^T B ; S = Y_B^T F + [0 | R]
    YB = [sum_op(mf, [(Vt[K][I], Ft[K][NT]) for K in range(NT)], zero) for I in range(NT)]
    S = [sum_op(mf, [(YB[K], Ft[K][J]) for K in range(NT)], Rt if J == NT else zero) for J in range(NT + 1)]
    Smat = np.hstack([untile_f32(t) for t in S])                      # (16, n + 16): [Sux | Suu]
    assert np.max(np.abs(Smat[:, :n] - B.T @ V @ A)) <= 1e-11 and np.max(np.abs(Smat[:, n:] - (R + B.T @ V @ B))) <= 1e-11
    YA = [[sum_op(mf, [(Vt[K][I], Ft[K][J]) for K in range(NT)], zero) for J in range(NT)] for I in range(NT)]
    L = np.linalg.solve(Smat[:, n:], Smat[:, :n])
    NL = [tile_f32(-L[:, 16 * J:16 * J + 16]) for J in range(NT)]
    T_ = lambda t: tile_f32(untile_f32(t).T)                           # what the LDS round trip does
    NRL = [op(mf, T_(Rt), NL[J], zero) for J in range(NT)]            # R (-L)
    Acl = [[op(mf, T_(Ft[K][NT]), NL[J], Ft[K][J]) for J in range(NT)] for K in range(NT)]     # A + B (-L)
    W = [[op(mf, T_(YB[K]), NL[J], YA[K][J]) for J in range(NT)] for K in range(NT)]           # Y_A + Y_B (-L)
    Vn = [[sum_op(mf, [(W[K][I], Acl[K][J]) for K in range(NT)], op(mf, NL[I], NRL[J], Qt[I][J])) for J in range(NT)]
          for I in range(NT)]
    Acl_ref = A - B @ L
    assert np.max(np.abs(_untiles(Acl) - Acl_ref)) <= 1e-11
    assert np.max(np.abs(_untiles(W) - V.T @ Acl_ref)) <= 1e-10
    V_ref = Q + L.T @ R @ L + Acl_ref.T @ V @ Acl_ref                 # lqrUtils.py:169
    assert np.max(np.abs(_untiles(Vn) - V_ref)) <= 1e-10 * np.max(np.abs(V_ref))


def sum_op(mfma, pairs, acc):
    for x, y in pairs:
        acc = op(mfma, x, y, acc)
    return acc


def test_four_block_fp64_mfma_layout():
    """v_mfma_f64_4x4x4_4b as probed on gfx950: block q = (lane & 15) >> 2; A_q[i][k] at lane 16 k + 4 q + i, B_q[k][j] at
    16 k + 4 q + j, D_q[i][j] at 16 i + 4 q + j.  With B = the registers of a 16x16x4 B operand, the four blocks are the four
    4-column groups of a 4 x 16 product -- the [Sux | Suu] rows of the LQR step."""
    rng = np.random.default_rng(2)
    lane = np.arange(64)
    Aq = rng.standard_normal((4, 4, 4))            # [q][i][k]
    Bq = rng.standard_normal((4, 4, 4))            # [q][k][j]
    a = np.empty(64); b = np.empty(64)
    for k in range(4):
        for q in range(4):
            for i in range(4):
                a[16 * k + 4 * q + i] = Aq[q, i, k]
                b[16 * k + 4 * q + i] = Bq[q, k, i]
    d = np.empty(64)
    for l in lane:                                  # the probe's finding, applied literally
        i, q, j = l >> 4, (l & 15) >> 2, l & 3
        d[l] = sum(a[16 * k + 4 * q + i] * b[16 * k + 4 * q + j] for k in range(4))
    for q in range(4):
        D = Aq[q] @ Bq[q]
        for i in range(4):
            for j in range(4):
                assert abs(d[16 * i + 4 * q + j] - D[i, j]) <= 1e-13
    # B operand of the big instruction (lane (g, c) holds F[g][c] for K-step 0) read as four blocks: S[u][c] = sum_k Y[k][u] F[k][c]
    F = rng.standard_normal((4, 16)); Y = rng.standard_normal((4, 4))
    bF = F[lane >> 4, lane & 15]
    aY = Y[lane >> 4, lane & 3]                     # A_q[u][k] = Y[k][u] at lane (g = k, c = 4 q + u)
    S = np.array([[sum(aY[16 * k + 4 * ((c >> 2)) + u] * bF[16 * k + c] for k in range(4)) for c in range(16)] for u in range(4)])
    assert np.max(np.abs(S - Y.T @ F)) <= 1e-13
