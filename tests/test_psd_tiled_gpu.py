"""GPU parity tests of the multi-tile PD projection (psd_tiled.hip): `ensurePositiveDefinite`, `conditionQuadraticCost`,
`conditionValueFunction` (reference ilqrUtils.py:217-234, 254-257) for 16 < k <= 64 against the oracle's `eigh` restatement -- dense
indefinite matrices, nonsymmetric inputs (eigh symmetrises), already-PD inputs, rank-deficient ones with eigenvalues hugging the clamp,
structurally zero rows / columns -- and, end to end, the generic-callable `iterativeLqr` at a shape beyond the one-tile kernels
(n = 20, m = 6: the tiled sweep, the tiled projection of the stacked 26 x 26 cost Hessian) against the oracle loop."""
import numpy as np
import pytest

from oracle import zopt_oracle as zo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import ilqrUtils, models, pytrees
    return ilqrUtils, models, pytrees


@pytest.mark.parametrize("k", [17, 24, 32, 33, 40, 48, 57, 64])
def test_ensurePositiveDefinite_large(mods, k):
    ilqr = mods[0]
    rng = np.random.default_rng(k)
    M = rng.standard_normal((6, k, k))
    A = M + np.swapaxes(M, -1, -2)                      # symmetric indefinite: the clamp is active
    A[0] = M[0]                                          # nonsymmetric input: eigh symmetrises it
    A[1] = M[1] @ M[1].T + np.eye(k)                     # already PD: (numerically) unchanged
    A[2] = np.diag(np.linspace(-1.0, 2.0, k))            # diagonal
    U, _ = np.linalg.qr(rng.standard_normal((k, k)))     # rank-deficient with eigenvalues at and around the clamp
    w = np.zeros(k)
    w[:5] = [3.0, 1e-3, 1.5e-3, -2.0, 5e-4]
    A[3] = (U * w) @ U.T
    A[4][:, 3] = 0.0                                     # a structurally zero row / column pair
    A[4][3, :] = 0.0
    out = ilqr.ensurePositiveDefinite(A)
    ref = zo.ensurePositiveDefinite(A)
    assert out.shape == A.shape
    for i in range(6):
        assert np.max(np.abs(out[i] - ref[i])) <= 2e-11 * max(1.0, np.max(np.abs(ref[i]))), i
    wmin = np.linalg.eigvalsh(0.5 * (out + np.swapaxes(out, -1, -2)))
    assert np.min(wmin) >= 1e-3 * (1 - 1e-6)
    assert out[1] == pytest.approx(A[1], rel=1e-11, abs=1e-12)
    assert out[4][3, 3] == pytest.approx(1e-3, abs=0) and not out[4][3, :3].any()     # decoupled zero index: eps on the diagonal


def test_ensurePositiveDefinite_large_zero_and_nonfinite(mods):
    ilqr = mods[0]
    A = np.zeros((3, 20, 20))
    A[1] = np.nan
    A[2, 0, 0] = np.inf
    out = ilqr.ensurePositiveDefinite(A)
    assert out[0] == pytest.approx(1e-3 * np.eye(20), abs=0)
    assert np.all(np.isnan(out[1]))
    assert not np.all(np.isfinite(out[2]))
    with pytest.raises(ValueError):
        ilqr.ensurePositiveDefinite(np.eye(65))


@pytest.mark.parametrize("n,m", [(12, 6), (20, 6), (40, 8), (48, 16)])
def test_conditionQuadraticCost_large(mods, n, m):
    ilqr, _, pt = mods
    rng = np.random.default_rng(n + m)
    b, N = 2, 3
    M = rng.standard_normal((b, N, n + m, n + m))
    H = M + np.swapaxes(M, -1, -2)
    cost = (rng.standard_normal((b, N)), rng.standard_normal((b, N, n)), rng.standard_normal((b, N, m)),
            np.ascontiguousarray(H[..., :n, :n]), np.ascontiguousarray(H[..., n:, :n]), np.ascontiguousarray(H[..., n:, n:]))
    out = ilqr.conditionQuadraticCost(pt.QuadraticCostFunction(*cost))
    ref = zo.conditionQuadraticCost(zo.QuadraticCostFunction(*cost))
    for name in ("c_xx", "c_ux", "c_uu"):
        assert _rel(getattr(out, name), getattr(ref, name)) <= 2e-11
    Vf = ilqr.conditionValueFunction(pt.QuadraticValueFunction(0.0, np.zeros(n), H[0, 0, :n, :n]))
    assert _rel(Vf.v_xx, zo.ensurePositiveDefinite(H[0, 0, :n, :n])) <= 2e-11


def test_generic_iterativeLqr_beyond_the_one_tile_shapes(mods):
    """`iterativeLqr` with torch callables at n = 20, m = 6 (a chain of damped pendulum-like cells with cubic stiffness): expansions
    and rollouts by torch.func on the GPU, backward pass on the tiled sweep, cost / terminal Hessians through the tiled projection;
    against the oracle loop (complex-step Jacobians, NumPy rollouts): same `converged`, cost to 1e-8, controls and gains to 1e-6."""
    import torch
    ilqr = mods[0]
    n, m, N = 20, 6, 12
    rng = np.random.default_rng(3)
    K = 0.15 * rng.standard_normal((n, n)) / np.sqrt(n)
    Bm = 0.3 * rng.standard_normal((n, m))
    dt = 0.1

    def f_np(x, u):
        return x + dt * (K @ x - 0.05 * x ** 3 + Bm @ u)

    tK, tB = torch.as_tensor(K, device="cuda"), torch.as_tensor(Bm, device="cuda")

    def f_t(x, u):
        return x + dt * (tK @ x - 0.05 * x ** 3 + tB @ u)

    Q, R, Qf = np.eye(n), 0.5 * np.eye(m), 5 * np.eye(n)
    tQ, tR, tQf = (torch.as_tensor(M, device="cuda") for M in (Q, R, Qf))
    x0 = rng.uniform(-1.5, 1.5, (2, n))
    ug = np.zeros((2, N, m))
    traj, L, J, conv = ilqr.iterativeLqr(f_t, lambda x, u: x @ tQ @ x + u @ tR @ u, lambda x: x @ tQf @ x, x0, ug)
    assert L.shape == (2, N, m, n)
    for i in range(2):
        rt, rL, rJ, rc = zo.iterativeLqr(f_np, Q, R, Qf, x0[i], ug[i])
        assert bool(conv[i]) == rc
        assert J[i] == pytest.approx(rJ, rel=1e-8)
        assert _rel(traj.uTraj[i], rt.uTraj) <= 1e-6 and _rel(L[i], rL) <= 1e-6
