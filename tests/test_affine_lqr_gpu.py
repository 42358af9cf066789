"""GPU parity tests for K2 lqr_backward_affine (bilinearAffineLqr, reference lqrUtils.py:207-262)."""
import json
import os

import numpy as np
import pytest

from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def lqr():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import lqrUtils
    return lqrUtils


def test_kat_bilinearAffineLqr(lqr):
    """reference tests/test_lqrUtils.py:82-98: L = I both steps, l[1] = 1.5, l[0] = 1."""
    k = KATS["A2_bilinearAffineLqr"]
    N = k["N"]
    I = np.repeat(np.eye(2)[None], N, axis=0)
    ones = np.ones((N, 2))
    L, l = lqr.bilinearAffineLqr(I, I, ones, I, I, I, ones, ones, np.ones(N), N)
    assert L == pytest.approx(np.array(k["L"]), rel=1e-12, abs=1e-14)
    assert l == pytest.approx(np.array(k["l"]), rel=1e-12, abs=1e-14)


def _problem(batch, T, n, m, seed, nonsym=False):
    rng = np.random.default_rng(seed)
    A, B, Q, R = problems.random_time_varying(batch, T, n, m, seed=seed)
    if nonsym:
        Q = Q + 0.3 * rng.standard_normal(Q.shape)
        R = R + 0.3 * rng.standard_normal(R.shape)
    d = 0.5 * rng.standard_normal((batch, T, n))
    H = 0.2 * rng.standard_normal((batch, T, m, n))
    q = rng.standard_normal((batch, T, n))
    r = rng.standard_normal((batch, T, m))
    q0 = rng.standard_normal((batch, T))
    return A, B, d, Q, R, H, q, r, q0


@pytest.mark.parametrize("n,m,T,batch,nonsym", [
    (8, 4, 100, 4, False),     # the reference's demo shape (demos/bilinearLqrControl.py:23-43)
    (12, 4, 50, 8, False), (12, 4, 20, 3, True), (8, 4, 10, 3, True), (2, 2, 2, 5, False), (4, 1, 30, 2, False),
    (1, 1, 5, 2, False), (5, 3, 7, 3, True), (7, 2, 9, 2, False), (11, 4, 4, 2, True), (12, 4, 1, 2, False), (3, 4, 6, 2, False),
])
def test_parity(lqr, n, m, T, batch, nonsym):
    args = _problem(batch, T, n, m, seed=1000 * n + 10 * m + T, nonsym=nonsym)
    L, l = lqr.bilinearAffineLqr(*args, T)
    Lr, lr = zo.bilinearAffineLqr(*args, T)
    assert L.shape == (batch, T, m, n) and l.shape == (batch, T, m)
    assert _rel(L, Lr) <= 1e-10 and _rel(l, lr) <= 1e-10


def test_reduces_to_plain_lqr_gain(lqr):
    """With d = 0, H = 0, q = r = 0 the gains are those of V' = Q + A'VA - L'SuuL (the Schur-form recursion)."""
    A, B, d, Q, R, H, q, r, q0 = _problem(3, 12, 12, 4, seed=9)
    z = lambda x: np.zeros_like(x)
    L, l = lqr.bilinearAffineLqr(A, B, z(d), Q, R, z(H), z(q), z(r), q0, 12)
    Lr, lr = zo.bilinearAffineLqr(A, B, z(d), Q, R, z(H), z(q), z(r), q0, 12)
    assert _rel(L, Lr) <= 1e-10 and not l.any()
    Lj = zo.discreteFiniteHorizonLqr(A, B, Q, R, 12)
    assert _rel(L, Lj) <= 1e-8        # same optimum as the Joseph-form recursion (different rounding path)


def test_no_batch_axis_and_torch(lqr):
    import torch
    args = _problem(1, 9, 8, 4, seed=4)
    sq = [x[0] for x in args]
    L, l = lqr.bilinearAffineLqr(*sq, 9)
    Lr, lr = zo.bilinearAffineLqr(*sq, 9)
    assert L.shape == (9, 4, 8) and l.shape == (9, 4)
    assert _rel(L, Lr) <= 1e-10 and _rel(l, lr) <= 1e-10
    t = [torch.as_tensor(x, device="cuda") for x in args]
    Lt, lt = lqr.bilinearAffineLqr(*t, 9)
    assert Lt.is_cuda and _rel(Lt.cpu().numpy()[0], Lr) <= 1e-10


@pytest.mark.parametrize("n,T", [(12, 1), (12, 2), (12, 3), (12, 50), (8, 1), (8, 2), (8, 3), (8, 100)])
def test_dma_ring_and_register_kernels_agree(n, T):
    """zm_lqr_backward_affine_f64 stages (n in {8, 12}, m = 4, 16-B aligned operands) through an LDS ring by DMA and
    runs every other case with register prefetch: both against the oracle, on every ring phase (T around the ring
    depth); the register kernel is reached by handing the same data over at 8-B-aligned addresses."""
    import ctypes
    import torch
    from zopt_amd import _lib
    m, batch = 4, 7
    A, B, d, Q, R, H, q, r, q0 = _problem(batch, T, n, m, seed=31 * n + T, nonsym=True)
    Lr, lr = zo.bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, T)
    outs = []
    for shift in (0, 1):
        dev = []
        for X in (A, B, d, Q, R, H, q, r):
            buf = torch.zeros(X.size + 2, dtype=torch.float64, device="cuda")
            buf[shift:shift + X.size] = torch.as_tensor(np.ascontiguousarray(X).ravel(), device="cuda")
            t = buf[shift:shift + X.size]
            assert t.data_ptr() % 16 == 8 * shift
            dev.append(t)
        dL = torch.empty((batch, T, m, n), dtype=torch.float64, device="cuda")
        dl = torch.empty((batch, T, m), dtype=torch.float64, device="cuda")
        rc = _lib.lib().zm_lqr_backward_affine_f64(*[t.data_ptr() for t in dev], dL.data_ptr(), dl.data_ptr(), batch, T, n, m,
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "zm_lqr_backward_affine_f64")
        torch.cuda.synchronize()
        L, l = dL.cpu().numpy(), dl.cpu().numpy()
        assert _rel(L, Lr) <= 1e-10 and _rel(l, lr) <= 1e-10
        outs.append((L, l))
    assert _rel(outs[0][0], outs[1][0]) <= 1e-12 and _rel(outs[0][1], outs[1][1]) <= 1e-12
