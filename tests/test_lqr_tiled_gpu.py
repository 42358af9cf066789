"""GPU parity tests for the fp32 MFMA tile kernel of discreteFiniteHorizonLqr (reference lqrUtils.py:144-173 in JAX's
default fp32 mode; BASELINE configs[4]: n=64, m=16, T=200 fp32).

Every call goes zopt_amd.lqrUtils -> ctypes -> zm_lqr_backward_f32 -> HIP kernel; the oracle is only the checker.

Tolerance (fp32): the kernel and an fp32 run of the oracle both carry rounding of order eps32 * cond per step, so they are
compared with the fp64 oracle on the SAME fp32-rounded inputs:
    max|L_gpu - L_f64| <= max(RTOL32 * max|L_f64|, 4 * max|L_np32 - L_f64|),   RTOL32 = 2e-4
(i.e. the kernel may be at most a small factor worse than NumPy's own fp32 recursion; measured: comparable or better).
"""
import numpy as np
import pytest

from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu

RTOL32 = 2e-4


@pytest.fixture(scope="module")
def lqr():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import lqrUtils
    return lqrUtils


def _check(Lg, A, B, Q, R, T):
    assert Lg.dtype == np.float32
    L64 = zo.discreteFiniteHorizonLqr(*(x.astype(np.float64) for x in (A, B, Q, R)), T)
    L32 = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
    scale = np.max(np.abs(L64))
    e_gpu, e_np = np.max(np.abs(Lg - L64)), np.max(np.abs(L32 - L64))
    assert np.all(np.isfinite(Lg))
    assert e_gpu <= max(RTOL32 * scale, 4 * e_np), (e_gpu / scale, e_np / scale)
    return e_gpu / scale, e_np / scale


@pytest.mark.parametrize("n,m,T,batch", [
    (64, 16, 20, 3),      # BASELINE configs[4] tile shape (exact tiles)
    (48, 16, 7, 2), (32, 16, 9, 3), (16, 16, 5, 4),
    (64, 7, 6, 2), (33, 16, 5, 2), (20, 5, 8, 3), (13, 3, 6, 2), (17, 1, 4, 2), (5, 9, 6, 3), (12, 16, 3, 2),
    (1, 5, 3, 2), (64, 16, 1, 2), (40, 12, 2, 1),
])
def test_time_varying_matches_oracle(lqr, n, m, T, batch):
    A, B, Q, R = problems.random_time_varying(batch, T, n, m, seed=100 + n + m, dtype=np.float32)
    Lg = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    assert Lg.shape == (batch, T, m, n)
    _check(Lg, A, B, Q, R, T)


def test_config5_shape_full_horizon(lqr):
    """n=64, m=16, T=200 (BASELINE configs[4]) on random stable LTI systems tiled over the horizon; besides the oracle,
    L[0] of the long horizon must agree with the infinite-horizon gain (SciPy DARE, the solver-independent gate of SURVEY 8c)."""
    import scipy.linalg
    n, m, T, batch = 64, 16, 200, 4
    A1, B1, Q1, R1 = problems.random_lti_systems(batch, n, m, seed=3, dtype=np.float32)
    A, B, Q, R = problems.tile_over_horizon(A1, B1, Q1, R1, T)
    Lg = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    _check(Lg, A, B, Q, R, T)
    for i in range(batch):
        a, b, q, r = (x[i].astype(np.float64) for x in (A1, B1, Q1, R1))
        P = scipy.linalg.solve_discrete_are(a, b, q, r)
        Linf = np.linalg.solve(r + b.T @ P @ b, b.T @ P @ a)
        assert np.max(np.abs(Lg[i, 0] - Linf)) <= 5e-4 * np.max(np.abs(Linf))


def test_batch_of_many_and_torch_path(lqr):
    """More trajectories than SIMDs get distinct answers; torch fp32 tensors in -> torch fp32 tensor out on the device."""
    import torch
    n, m, T, nsys = 64, 16, 12, 8
    A1, B1, Q1, R1 = problems.random_lti_systems(nsys, n, m, seed=5, dtype=np.float32)
    reps = 160                                   # 1280 trajectories > 1024 SIMDs
    idx = np.arange(nsys * reps) % nsys
    A, B, Q, R = problems.tile_over_horizon(A1[idx], B1[idx], Q1[idx], R1[idx], T)
    tA, tB, tQ, tR = (torch.as_tensor(x, device="cuda") for x in (A, B, Q, R))
    Lt = lqr.discreteFiniteHorizonLqr(tA, tB, tQ, tR, T)
    assert isinstance(Lt, torch.Tensor) and Lt.is_cuda and Lt.dtype == torch.float32 and Lt.shape == (nsys * reps, T, m, n)
    Lg = Lt.cpu().numpy()
    _check(Lg[:nsys], A[:nsys], B[:nsys], Q[:nsys], R[:nsys], T)
    assert np.array_equal(Lg.reshape(reps, nsys, T, m, n), np.broadcast_to(Lg[:nsys], (reps, nsys, T, m, n)))


def test_partial_pivoting_needed(lqr):
    """R = cyclic permutation matrix + small noise, weak B: Suu = R + B^T V B is well conditioned but its diagonal is ~100x
    smaller than its off-diagonal entries, so elimination without row exchanges loses the answer in fp32;
    jnp.linalg.solve (LU, partial pivoting) and this kernel do not."""
    rng = np.random.default_rng(9)
    n, m, T, batch = 32, 16, 4, 3
    A, B, Q, _ = problems.random_time_varying(batch, T, n, m, seed=9, dtype=np.float64)
    B *= 0.02
    R = np.roll(np.eye(m), 1, axis=1)[None, None] + 0.01 * rng.standard_normal((batch, T, m, m))
    A, B, Q, R = (np.ascontiguousarray(x, dtype=np.float32) for x in (A, B, Q, R))
    Lg = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    L64 = zo.discreteFiniteHorizonLqr(*(x.astype(np.float64) for x in (A, B, Q, R)), T)
    L32 = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
    scale = np.max(np.abs(L64))
    assert scale > 1e-2
    assert np.max(np.abs(Lg - L64)) <= max(RTOL32 * scale, 4 * np.max(np.abs(L32 - L64)))


def test_small_fp32_shapes_keep_the_fp64_tile_path(lqr):
    """fp32 inputs with n <= 12, m <= 4 are computed in fp64 and rounded once: at least as accurate as an fp32 recursion."""
    A, B, Q, R = problems.random_time_varying(3, 10, 12, 4, seed=2, dtype=np.float32)
    Lg = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 10)
    assert Lg.dtype == np.float32
    L64 = zo.discreteFiniteHorizonLqr(*(x.astype(np.float64) for x in (A, B, Q, R)), 10)
    assert np.max(np.abs(Lg - L64)) <= 2e-7 * np.max(np.abs(L64))


def test_unsupported_shapes_raise(lqr):
    A, B, Q, R = problems.random_time_varying(1, 2, 65, 4, seed=1, dtype=np.float32)
    with pytest.raises(ValueError):
        lqr.discreteFiniteHorizonLqr(A, B, Q, R, 2)
    A, B, Q, R = problems.random_time_varying(1, 2, 20, 17, seed=1, dtype=np.float64)
    with pytest.raises(ValueError):
        lqr.discreteFiniteHorizonLqr(A, B, Q, R, 2)


@pytest.mark.parametrize("n,m,T,batch", [(64, 16, 6, 3), (20, 4, 9, 2), (13, 1, 5, 2), (12, 5, 7, 3), (33, 7, 4, 2), (3, 16, 3, 2),
                                         (48, 12, 30, 2), (16, 16, 8, 3), (32, 16, 12, 2), (48, 16, 5, 2), (49, 3, 4, 2),
                                         (64, 16, 40, 2), (57, 9, 5, 3), (64, 1, 3, 2), (50, 16, 1, 2)])
def test_fp64_beyond_tile16_lds_kernel(lqr, n, m, T, batch):
    """fp64 inputs outside n <= 12, m <= 4 run the fp64 MFMA tile kernel: three tile rows with a prefetched operand set up to n = 48,
    four tile rows without it (PREFETCH = false, lqr_tiled_core.h) for 48 < n <= 64: 1e-10."""
    A, B, Q, R = problems.random_time_varying(batch, T, n, m, seed=50 + n + m, dtype=np.float64)
    Lg = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    Lr = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
    assert Lg.dtype == np.float64 and Lg.shape == (batch, T, m, n)
    assert np.max(np.abs(Lg - Lr)) <= 1e-10 * np.max(np.abs(Lr))


def test_fp64_lds_kernel_pivoting_and_nonsymmetric(lqr):
    rng = np.random.default_rng(12)
    n, m, T, batch = 24, 8, 4, 3
    A, B, Q, _ = problems.random_time_varying(batch, T, n, m, seed=12, dtype=np.float64)
    B *= 0.02
    Q = Q + 0.2 * rng.standard_normal(Q.shape)                          # nonsymmetric Q (and hence V)
    R = np.roll(np.eye(m), 1, axis=1)[None, None] + 0.01 * rng.standard_normal((batch, T, m, m))
    Lg = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    Lr = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
    assert np.max(np.abs(Lg - Lr)) <= 1e-10 * np.max(np.abs(Lr))


def test_fp64_four_tile_rows_pivoting_and_nonsymmetric(lqr):
    """48 < n <= 64 in fp64 (four tile rows, no prefetched operand set): nonsymmetric Q and a permuted R that forces the pivoted solve."""
    rng = np.random.default_rng(13)
    n, m, T, batch = 56, 8, 4, 2
    A, B, Q, _ = problems.random_time_varying(batch, T, n, m, seed=13, dtype=np.float64)
    B *= 0.02
    Q = Q + 0.2 * rng.standard_normal(Q.shape)
    R = np.roll(np.eye(m), 1, axis=1)[None, None] + 0.01 * rng.standard_normal((batch, T, m, m))
    Lg = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    Lr = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
    assert np.max(np.abs(Lg - Lr)) <= 1e-10 * np.max(np.abs(Lr))


@pytest.mark.parametrize("n,m", [(24, 8), (60, 12)])
def test_fp64_tile_kernel_and_lds_coverage_kernel_agree(lqr, n, m):
    """The LDS coverage kernel stays selectable (ZOPT_AMD_LQR_PATH=lds, a product fallback switch): same gains as the tile kernels at
    three and at four tile rows."""
    import subprocess, sys, json, os
    A, B, Q, R = problems.random_time_varying(2, 5, n, m, seed=78)
    La = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 5)
    code = ("import numpy as np, json, sys; sys.path.insert(0, %r); from tests import problems; from zopt_amd import lqrUtils;"
            "A,B,Q,R = problems.random_time_varying(2, 5, %d, %d, seed=78); print(json.dumps(lqrUtils.discreteFiniteHorizonLqr(A,B,Q,R,5).tolist()))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n, m))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZOPT_AMD_LQR_PATH="lds"), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-500:]
    Lb = np.array(json.loads(out.stdout.strip().splitlines()[-1]))
    assert np.max(np.abs(La - Lb)) <= 1e-11 * np.max(np.abs(Lb))


def test_fp64_tile_kernel_and_lds_kernel_agree(lqr, monkeypatch):
    """The two fp64 paths for medium sizes (MFMA tiles / LDS coverage kernel, ZOPT_AMD_LQR_PATH=lds) give the same gains."""
    import subprocess, sys, json, os
    A, B, Q, R = problems.random_time_varying(2, 6, 24, 8, seed=77)
    La = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 6)
    code = ("import numpy as np, json, sys; sys.path.insert(0, %r); from tests import problems; from zopt_amd import lqrUtils;"
            "A,B,Q,R = problems.random_time_varying(2, 6, 24, 8, seed=77); print(json.dumps(lqrUtils.discreteFiniteHorizonLqr(A,B,Q,R,6).tolist()))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZOPT_AMD_LQR_PATH="lds"), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-500:]
    Lb = np.array(json.loads(out.stdout.strip().splitlines()[-1]))
    assert np.max(np.abs(La - Lb)) <= 1e-11 * np.max(np.abs(Lb))
