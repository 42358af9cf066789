"""GPU parity tests for K1 lqr_backward (discreteFiniteHorizonLqr, reference lqrUtils.py:144-173).

Every call goes zopt_amd.lqrUtils -> ctypes -> C ABI -> HIP kernel; the oracle is only the checker.
Tolerance (fp64): max|L_gpu - L_oracle| <= 1e-10 * max|L_oracle| (measured ~1e-14: same formulas, different
summation order / FMA contraction inside the MFMA).
"""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu

RTOL = 1e-10
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _rel(Lg, Lr):
    return np.max(np.abs(Lg - Lr)) / max(np.max(np.abs(Lr)), 1e-300)


@pytest.fixture(scope="module")
def lqr():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import lqrUtils
    return lqrUtils


def test_kat_identity(lqr):
    """reference tests/test_lqrUtils.py:61-69: L[1] = 0.5 I, L[0] = 0.6 I."""
    N = 2
    I = np.repeat(np.eye(2)[None], N, axis=0)
    L = lqr.discreteFiniteHorizonLqr(I, I, I, I, N)
    assert L.shape == (2, 2, 2)
    assert L == pytest.approx(np.array(KATS["A1_discreteFiniteHorizonLqr"]["L"]), rel=1e-12, abs=1e-15)


@pytest.mark.parametrize("n,m,T,batch", [
    (12, 4, 50, 64),     # BASELINE config 2 shape
    (4, 1, 50, 1),       # BASELINE config 1 (plumbing) shape
    (8, 4, 100, 5),      # the reference's actual demo shape (demos/discreteFiniteHorizonLqr.py:14-35)
    (2, 2, 3, 3), (1, 1, 5, 2), (3, 4, 6, 7), (5, 2, 11, 4), (7, 3, 9, 3), (9, 1, 4, 2), (11, 4, 7, 9), (12, 1, 3, 2),
    (12, 4, 1, 3), (12, 4, 2, 3), (12, 4, 3, 3), (12, 4, 4, 2), (6, 3, 1, 1),
])
def test_parity_time_varying(lqr, n, m, T, batch):
    A, B, Q, R = problems.random_time_varying(batch, T, n, m, seed=1000 * n + 10 * m + T)
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    Lr = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
    assert L.shape == (batch, T, m, n) and L.dtype == np.float64
    assert _rel(L, Lr) <= RTOL


def test_reference_shapes_without_batch_axis(lqr):
    A, B, Q, R = problems.random_time_varying(1, 50, 4, 1, seed=4)
    L = lqr.discreteFiniteHorizonLqr(A[0], B[0], Q[0], R[0], 50)
    assert L.shape == (50, 1, 4)
    assert _rel(L, zo.discreteFiniteHorizonLqr(A[0], B[0], Q[0], R[0], 50)) <= RTOL


def test_extra_leading_axes(lqr):
    A, B, Q, R = problems.random_time_varying(6, 8, 12, 4, seed=9)
    rs = lambda X: X.reshape((2, 3) + X.shape[1:])
    L = lqr.discreteFiniteHorizonLqr(rs(A), rs(B), rs(Q), rs(R), 8)
    assert L.shape == (2, 3, 8, 4, 12)
    assert _rel(L.reshape(6, 8, 4, 12), zo.discreteFiniteHorizonLqr(A, B, Q, R, 8)) <= RTOL


def test_pivoting_paths(lqr):
    """R with dominant off-diagonals forces row swaps in the m x m LU (partial pivoting, as getrf)."""
    rng = np.random.default_rng(42)
    batch, T, n, m = 16, 6, 12, 4
    A, B, Q, R = problems.random_time_varying(batch, T, n, m, seed=77)
    P = np.eye(m)[[2, 0, 3, 1]]
    R = R @ P * 3.0 + 0.1 * rng.standard_normal(R.shape)   # nonsymmetric, permuted-dominant
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    Lr = zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
    assert np.all(np.isfinite(Lr))
    assert _rel(L, Lr) <= 1e-9


def test_cheap_control_needs_joseph_form(lqr):
    """R = 1e-6 I, unstable A: a Schur-form update loses ~1e-5; the Joseph form (lqrUtils.py:169) must hold 1e-10."""
    A1, B1, Q1, R1 = problems.random_lti_systems(8, 12, 4, seed=5, rho=1.2)
    R1[:] = 1e-6 * np.eye(4)
    A, B, Q, R = problems.tile_over_horizon(A1, B1, Q1, R1, 50)
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 50)
    assert _rel(L, zo.discreteFiniteHorizonLqr(A, B, Q, R, 50)) <= 1e-9


def test_config2_full_size_against_c_oracle(lqr):
    """BASELINE config 2 at full size: 4096 random LTI systems, n=12 m=4 T=50 fp64, every gain checked."""
    A1, B1, Q1, R1 = problems.random_lti_systems(4096, 12, 4, seed=0)
    A, B, Q, R = problems.tile_over_horizon(A1, B1, Q1, R1, 50)
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 50)
    Lr = c_oracle.lqr_backward(A, B, Q, R)
    assert _rel(L, Lr) <= RTOL
    # size-independent property: LTI => L[0] approaches the stationary (DARE) gain, solver-independent check
    import scipy.linalg as spl
    for i in (0, 1, 4095):
        V = spl.solve_discrete_are(A1[i], B1[i], Q1[i], R1[i])
        Ld = np.linalg.solve(R1[i] + B1[i].T @ V @ B1[i], B1[i].T @ V @ A1[i])
        assert np.max(np.abs(L[i, 0] - Ld)) <= 1e-6 * np.max(np.abs(Ld))


def test_golden_config2_first_systems(lqr):
    """Committed golden vectors (tests/golden/lqr_config2_first8.npz: inputs + oracle gains, see make_golden.py)."""
    path = os.path.join(os.path.dirname(__file__), "golden", "lqr_config2_first8.npz")
    g = np.load(path)
    A, B, Q, R = problems.tile_over_horizon(g["A"], g["B"], g["Q"], g["R"], int(g["T"]))
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, int(g["T"]))
    assert _rel(L, g["L"]) <= RTOL


def test_singular_system_propagates_nonfinite_without_fault(lqr):
    """JAX never raises on a singular solve; it returns inf/NaN.  R = 0 and B = 0 => Suu = 0."""
    T, n, m = 4, 12, 4
    A = np.repeat(np.eye(n)[None, None], T, axis=1)
    B = np.zeros((1, T, n, m))
    Q = np.repeat(np.eye(n)[None, None], T, axis=1)
    R = np.zeros((1, T, m, m))
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    assert not np.any(np.isfinite(L))


def test_torch_tensors_stay_on_device(lqr):
    import torch
    A, B, Q, R = problems.random_time_varying(32, 20, 12, 4, seed=3)
    tA, tB, tQ, tR = (torch.as_tensor(x, device="cuda") for x in (A, B, Q, R))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tL = lqr.discreteFiniteHorizonLqr(tA, tB, tQ, tR, 20)
    s.synchronize()
    assert tL.is_cuda and tL.dtype == torch.float64
    assert _rel(tL.cpu().numpy(), zo.discreteFiniteHorizonLqr(A, B, Q, R, 20)) <= RTOL


def test_fp32_inputs_follow_input_dtype(lqr):
    A, B, Q, R = problems.random_time_varying(4, 12, 12, 4, seed=8, dtype=np.float32)
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, 12)
    assert L.dtype == np.float32
    Lr = zo.discreteFiniteHorizonLqr(A.astype(np.float64), B.astype(np.float64), Q.astype(np.float64),
                                     R.astype(np.float64), 12)
    assert _rel(L, Lr) <= 1e-5   # fp32 tolerance


def test_host_pointer_entry_point(lqr):
    """zm_lqr_backward_host_f64: NumPy pointers straight through the C ABI (the binding INTEGRATION.md shows)."""
    import ctypes
    from zopt_amd import _lib
    A, B, Q, R = problems.random_time_varying(5, 7, 8, 4, seed=21)
    L = np.empty((5, 7, 4, 8))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = _lib.lib().zm_lqr_backward_host_f64(p(A), p(B), p(Q), p(R), p(L), 5, 7, 8, 4)
    assert rc == 0
    assert _rel(L, zo.discreteFiniteHorizonLqr(A, B, Q, R, 7)) <= RTOL


def test_unsupported_shape_raises_valueerror(lqr):
    A, B, Q, R = problems.random_time_varying(1, 3, 65, 5, seed=2)      # beyond n <= 64, m <= 16
    with pytest.raises(ValueError):
        lqr.discreteFiniteHorizonLqr(A, B, Q, R, 3)


@pytest.mark.parametrize("n,m,T,batch", [(12, 4, 50, 33), (8, 4, 7, 5), (12, 4, 1, 3), (12, 4, 2, 2)])
def test_fp32_storage_fast_path_is_fp64_arithmetic_on_fp32_arrays(lqr, n, m, T, batch):
    """fp32 inputs at the fast-path shapes run K1 on 4-byte ring elements (zm_lqr_backward_f32 -> lqr_backward_dma_f64<..., float>):
    operands widened as they are read, fp64 arithmetic, L_k narrowed as it is stored.  So the result is the fp64 oracle's on the same
    (fp32-representable) inputs rounded once -- to within one fp32 ulp -- on every ring phase (horizons 1, 2, 7, 50); and the fp32 tile
    kernel (ZOPT_AMD_LQR_F32=tile, fp32 arithmetic) stays within the fp32 tolerance of it."""
    A, B, Q, R = problems.random_time_varying(batch, T, n, m, seed=n + T, dtype=np.float32)
    L = lqr.discreteFiniteHorizonLqr(A, B, Q, R, T)
    assert L.dtype == np.float32 and L.shape == (batch, T, m, n)
    Lr = zo.discreteFiniteHorizonLqr(*(X.astype(np.float64) for X in (A, B, Q, R)), T)
    ulp = np.spacing(np.abs(Lr).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(L.astype(np.float64) - Lr) <= 0.5 * ulp + 1e-12 * np.abs(Lr).max())
