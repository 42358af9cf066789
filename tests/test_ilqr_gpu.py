"""GPU parity tests for K3 ilqr_backward (backwardPass_ilqr, reference ilqrUtils.py:153-181).

zopt_amd.ilqrUtils -> ctypes -> C ABI -> HIP kernel; the NumPy oracle is only the checker.
Tolerance (fp64): max|err| <= 1e-10 * max|ref| on l and L (measured ~1e-14)."""
import json
import os

import numpy as np
import pytest

from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu
RTOL = 1e-10
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def ilqr():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import ilqrUtils
    return ilqrUtils


def test_kat_riccati_step(ilqr):
    """reference tests/test_ilqrUtils.py:56-81 as a 1-step backward pass: l = 0, L = -0.5 I (exact)."""
    from zopt_amd import pytrees as pt
    I2 = np.eye(2)
    dyn = pt.AffineDynamics(np.zeros((1, 2)), I2[None], I2[None])
    cost = pt.QuadraticCostFunction(np.zeros(1), np.zeros((1, 2)), np.zeros((1, 2)), I2[None], np.zeros((1, 2, 2)), I2[None])
    Vf = pt.QuadraticValueFunction(0.0, np.zeros(2), I2)
    pol = ilqr.backwardPass_ilqr(dyn, cost, Vf)
    assert isinstance(pol, pt.AffinePolicy)
    assert np.all(pol.l[0] == np.array(KATS["A3_riccatiStep_ilqr"]["l"]))
    assert np.all(pol.L[0] == np.array(KATS["A3_riccatiStep_ilqr"]["L"]))


def test_kat_two_step_identity(ilqr):
    """reference tests/test_ilqrUtils.py:84-107 inputs (N=2 identities): hand-derived Riccati values L[1]=-0.5 I, L[0]=-0.6 I."""
    from zopt_amd import pytrees as pt
    N = 2
    I = np.repeat(np.eye(2)[None], N, axis=0)
    pol = ilqr.backwardPass_ilqr(pt.AffineDynamics(np.zeros((N, 2)), I, I),
                                 pt.QuadraticCostFunction(np.zeros(N), np.zeros((N, 2)), np.zeros((N, 2)), I,
                                                          np.zeros((N, 2, 2)), I),
                                 pt.QuadraticValueFunction(0.0, np.zeros(2), np.eye(2)))
    assert pol.L[1] == pytest.approx(-0.5 * np.eye(2), rel=1e-14)
    assert pol.L[0] == pytest.approx(-0.6 * np.eye(2), rel=1e-14)
    assert np.all(pol.l == 0)


@pytest.mark.parametrize("n,m,T,batch", [
    (12, 4, 100, 16),   # BASELINE config 4 shape (quadcopter n=12, m=4, T=100)
    (12, 4, 30, 5), (8, 4, 10, 3), (4, 1, 20, 2), (2, 2, 3, 4), (1, 1, 4, 2), (5, 3, 7, 3), (7, 2, 5, 2),
    (9, 4, 6, 3), (11, 1, 3, 2), (12, 3, 1, 2), (12, 4, 2, 2), (3, 4, 5, 3), (6, 4, 3, 1),
])
def test_parity_random_models(ilqr, n, m, T, batch):
    dyn, cost, Vf = problems.random_ilqr_model(batch, T, n, m, seed=100 * n + 10 * m + T)
    pol = ilqr.backwardPass_ilqr(dyn, cost, Vf)
    ref = zo.backwardPass_ilqr(zo.AffineDynamics(*dyn), zo.QuadraticCostFunction(*cost), zo.QuadraticValueFunction(*Vf))
    assert pol.l.shape == (batch, T, m) and pol.L.shape == (batch, T, m, n)
    assert _rel(pol.L, ref.L) <= RTOL
    assert _rel(pol.l, ref.l) <= RTOL


def test_reference_shapes_without_batch_axis(ilqr):
    dyn, cost, Vf = problems.random_ilqr_model(1, 9, 12, 4, seed=3)
    sq = lambda t: tuple(x[0] for x in t)
    pol = ilqr.backwardPass_ilqr(sq(dyn), sq(cost), sq(Vf))
    ref = zo.backwardPass_ilqr(zo.AffineDynamics(*sq(dyn)), zo.QuadraticCostFunction(*sq(cost)),
                               zo.QuadraticValueFunction(*sq(Vf)))
    assert pol.L.shape == (9, 4, 12) and pol.l.shape == (9, 4)
    assert _rel(pol.L, ref.L) <= RTOL and _rel(pol.l, ref.l) <= RTOL


def test_lq_problem_equals_riccati(ilqr):
    """For an LQ problem (cost x'Qx + u'Ru => c_xx = 2Q, c_uu = 2R, quirk Q7) the iLQR gains are minus the LQR gains
    of discreteFiniteHorizonLqr with terminal value Q (lqrUtils.py:144-173 vs ilqrUtils.py:153-181)."""
    from zopt_amd import lqrUtils
    b, T, n, m = 6, 12, 12, 4
    A, B, Q, R = problems.random_time_varying(b, T + 1, n, m, seed=5)
    # LQR: stage costs Q[0..T-1], terminal Q[T]  == arrays of length T+1 with B-row T unused by construction below
    dyn = (np.zeros((b, T, n)), A[:, :T], B[:, :T])
    cost = (np.zeros((b, T)), np.zeros((b, T, n)), np.zeros((b, T, m)), 2 * Q[:, :T], np.zeros((b, T, m, n)), 2 * R[:, :T])
    Vf = (np.zeros(b), np.zeros((b, n)), 2 * Q[:, T])
    pol = ilqr.backwardPass_ilqr(dyn, cost, Vf)
    ref = zo.backwardPass_ilqr(zo.AffineDynamics(*dyn), zo.QuadraticCostFunction(*cost), zo.QuadraticValueFunction(*Vf))
    assert _rel(pol.L, ref.L) <= RTOL
    assert np.max(np.abs(pol.l)) == 0.0


def test_pivoting_and_torch(ilqr):
    import torch
    dyn, cost, Vf = problems.random_ilqr_model(8, 6, 12, 4, seed=11)
    c, c_x, c_u, c_xx, c_ux, c_uu = cost
    P = np.eye(4)[[2, 0, 3, 1]]
    c_uu = c_uu @ P * 3.0                 # permuted-dominant nonsymmetric: forces the pivoted LU path
    cost = (c, c_x, c_u, c_xx, c_ux, c_uu)
    ref = zo.backwardPass_ilqr(zo.AffineDynamics(*dyn), zo.QuadraticCostFunction(*cost), zo.QuadraticValueFunction(*Vf))
    t = lambda tup: tuple(torch.as_tensor(np.asarray(x), device="cuda") for x in tup)
    pol = ilqr.backwardPass_ilqr(t(dyn), t(cost), t(Vf))
    assert pol.L.is_cuda
    assert _rel(pol.L.cpu().numpy(), ref.L) <= 1e-9 and _rel(pol.l.cpu().numpy(), ref.l) <= 1e-9


def test_riccatiStep_known_answers_and_value_function():
    """reference tests/test_ilqrUtils.py:56-81 (riccatiStep_ilqr, exact ==) and :110-135 (riccatiStep_ddp, rel 1e-3):
    the single-step helpers return the new value function next to the policy."""
    from zopt_amd import ilqrUtils, pytrees
    I, Z, z2 = np.eye(2), np.zeros((2, 2)), np.zeros(2)
    dyn = pytrees.AffineDynamics(z2, I, I)
    cost = pytrees.QuadraticCostFunction(0.0, z2, z2, I, Z, I)
    val = pytrees.QuadraticValueFunction(0.0, z2, I)
    V, pol = ilqrUtils.riccatiStep_ilqr(dyn, cost, val)
    assert V.v == 0 and np.all(V.v_x == 0) and np.array_equal(V.v_xx, 1.5 * I)
    assert np.all(pol.l == 0) and np.array_equal(pol.L, -0.5 * I)
    zz = np.zeros((2, 2, 2))
    V2, pol2 = ilqrUtils.riccatiStep_ddp(pytrees.QuadraticDynamics(z2, I, I, zz, zz, zz), cost, val)
    assert V2.v == 0 and np.all(V2.v_x == 0)
    assert V2.v_xx == pytest.approx(1.5 * I, rel=1e-3) and pol2.L == pytest.approx(-0.5 * I, rel=1e-3)
    # random batched steps against the oracle's step functions (value function included)
    rng = np.random.default_rng(17)
    for n, m in ((12, 4), (5, 2), (8, 4)):
        b = 6
        f = rng.standard_normal((b, n))
        f_x = rng.standard_normal((b, n, n)) * (0.9 / np.sqrt(n))
        f_u = rng.standard_normal((b, n, m))
        M = rng.standard_normal((b, n + m, n + m))
        H = M @ np.swapaxes(M, -1, -2) / (n + m) + np.eye(n + m)
        c_xx, c_ux, c_uu = H[:, :n, :n].copy(), H[:, n:, :n].copy(), H[:, n:, n:].copy()
        c, c_x, c_u = rng.standard_normal(b), rng.standard_normal((b, n)), rng.standard_normal((b, m))
        Mv = rng.standard_normal((b, n, n))
        v, v_x, v_xx = rng.standard_normal(b), rng.standard_normal((b, n)), Mv @ np.swapaxes(Mv, -1, -2) / n + np.eye(n)
        Vg, pg = ilqrUtils.riccatiStep_ilqr(pytrees.AffineDynamics(f, f_x, f_u),
                                            pytrees.QuadraticCostFunction(c, c_x, c_u, c_xx, c_ux, c_uu),
                                            pytrees.QuadraticValueFunction(v, v_x, v_xx))
        f_xx, f_ux, f_uu = (0.1 * rng.standard_normal((b, n) + s_) for s_ in ((n, n), (m, n), (m, m)))
        f_xx = 0.5 * (f_xx + np.swapaxes(f_xx, -1, -2))
        f_uu = 0.5 * (f_uu + np.swapaxes(f_uu, -1, -2))
        Vd, pd = ilqrUtils.riccatiStep_ddp(pytrees.QuadraticDynamics(f, f_x, f_u, f_xx, f_ux, f_uu),
                                           pytrees.QuadraticCostFunction(c, c_x, c_u, c_xx, c_ux, c_uu),
                                           pytrees.QuadraticValueFunction(v, v_x, v_xx))
        for i in range(b):
            Vr, pr = zo.riccatiStep_ilqr(zo.AffineDynamics(f[i], f_x[i], f_u[i]),
                                         zo.QuadraticCostFunction(c[i], c_x[i], c_u[i], c_xx[i], c_ux[i], c_uu[i]),
                                         zo.QuadraticValueFunction(v[i], v_x[i], v_xx[i]))
            assert abs(Vg.v[i] - Vr.v) <= 1e-10 * max(1.0, abs(Vr.v))
            assert _rel(Vg.v_x[i], Vr.v_x) <= 1e-10 and _rel(Vg.v_xx[i], Vr.v_xx) <= 1e-10
            assert _rel(pg.l[i], pr.l) <= 1e-10 and _rel(pg.L[i], pr.L) <= 1e-10
            Vr, pr = zo.riccatiStep_ddp(zo.QuadraticDynamics(f[i], f_x[i], f_u[i], f_xx[i], f_ux[i], f_uu[i]),
                                        zo.QuadraticCostFunction(c[i], c_x[i], c_u[i], c_xx[i], c_ux[i], c_uu[i]),
                                        zo.QuadraticValueFunction(v[i], v_x[i], v_xx[i]))
            assert abs(Vd.v[i] - Vr.v) <= 1e-9 * max(1.0, abs(Vr.v))
            assert _rel(Vd.v_x[i], Vr.v_x) <= 1e-9 and _rel(Vd.v_xx[i], Vr.v_xx) <= 1e-9
            assert _rel(pd.l[i], pr.l) <= 1e-9 and _rel(pd.L[i], pr.L) <= 1e-9


@pytest.mark.parametrize("n,T", [(12, 1), (12, 2), (12, 3), (12, 4), (12, 7), (12, 100), (8, 1), (8, 2), (8, 5), (8, 33)])
@pytest.mark.parametrize("shared", [False, True])
def test_dma_ring_and_register_kernels_agree(n, T, shared):
    """zm_ilqr_backward_ex_f64 stages (n in {8, 12}, m = 4, 16-B aligned operands) through an LDS ring by DMA and runs
    every other case with register prefetch: both against the oracle, on every ring phase (T around the ring depth 3),
    with per-step and shared (time-invariant) Hessians and an active mask; the register kernel is reached by handing
    the same data over at 8-B-aligned addresses."""
    import ctypes
    import torch
    from zopt_amd import _lib
    m, batch = 4, 9
    dyn, cost, Vf = problems.random_ilqr_model(batch, T, n, m, seed=7 * n + T)
    c, c_x, c_u, c_xx, c_ux, c_uu = cost
    v, v_x, v_xx = Vf
    if shared:   # one Hessian (and terminal Hessian) for every trajectory and step
        c_xx = np.broadcast_to(c_xx[0, 0], c_xx.shape).copy()
        c_ux = np.broadcast_to(c_ux[0, 0], c_ux.shape).copy()
        c_uu = np.broadcast_to(c_uu[0, 0], c_uu.shape).copy()
        v_xx = np.broadcast_to(v_xx[0], v_xx.shape).copy()
    ref = zo.backwardPass_ilqr(zo.AffineDynamics(*dyn), zo.QuadraticCostFunction(c, c_x, c_u, c_xx, c_ux, c_uu),
                               zo.QuadraticValueFunction(v, v_x, v_xx))
    act = np.ones(batch, dtype=np.int32)
    act[[2, 5]] = 0
    host = [dyn[1], dyn[2], c_x, c_u] + ([c_xx[0, 0], c_ux[0, 0], c_uu[0, 0]] if shared else [c_xx, c_ux, c_uu]) + \
           [v_x, v_xx[0] if shared else v_xx]
    dact = torch.as_tensor(act, device="cuda")
    outs = []
    for shift in (0, 1):    # 0: 16-B aligned (DMA ring); 1: every operand at +8 B (register kernel)
        dev = []
        for X in host:
            buf = torch.zeros(X.size + 2, dtype=torch.float64, device="cuda")
            buf[shift:shift + X.size] = torch.as_tensor(np.ascontiguousarray(X).ravel(), device="cuda")
            t = buf[shift:shift + X.size]
            assert t.data_ptr() % 16 == 8 * shift
            dev.append(t)
        dl = torch.full((batch, T, m), 77.0, dtype=torch.float64, device="cuda")
        dL = torch.full((batch, T, m, n), 77.0, dtype=torch.float64, device="cuda")
        rc = _lib.lib().zm_ilqr_backward_ex_f64(*[t.data_ptr() for t in dev], dact.data_ptr(), 1 if shared else 0,
                                                dl.data_ptr(), dL.data_ptr(), batch, T, n, m,
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "zm_ilqr_backward_ex_f64")
        torch.cuda.synchronize()
        l, L = dl.cpu().numpy(), dL.cpu().numpy()
        on = act == 1
        assert np.all(l[~on] == 77.0) and np.all(L[~on] == 77.0)      # masked trajectories keep their policy
        assert _rel(L[on], ref.L[on]) <= RTOL and _rel(l[on], ref.l[on]) <= RTOL
        outs.append((l, L))
    assert _rel(outs[0][1], outs[1][1]) <= 1e-12 and _rel(outs[0][0], outs[1][0]) <= 1e-12


@pytest.mark.parametrize("n,ddp", [(12, False), (12, True), (5, False), (5, True)])
def test_sweeps_over_an_id_list_match_the_full_calls(n, ddp):
    """zm_ilqr_backward_list_f64 / zm_ddp_backward_list_f64 run one wave per LISTED trajectory: the listed ones get bit for bit
    what the masked full-batch call writes (DMA-ring kernel at n = 12, register kernel at n = 5), the others keep their policy,
    listed-but-inactive ones are skipped, an empty list is a no-op, an over-long list is rejected."""
    import ctypes
    import torch
    from zopt_amd import _lib
    lib = _lib.lib()
    m, B, T = (4 if n == 12 else 3), 10, 9
    dyn, cost, Vf = problems.random_ilqr_model(B, T, n, m, seed=50 + n)
    rng = np.random.default_rng(n)
    dev = [torch.as_tensor(np.ascontiguousarray(X), device="cuda") for X in (dyn[1], dyn[2], cost[1], cost[2], cost[3], cost[4], cost[5],
                                                                             Vf[1], Vf[2])]
    z = [torch.as_tensor(0.1 * rng.standard_normal(s), device="cuda") for s in ((B, T, n, n, n), (B, T, n, m, n), (B, T, n, m, m))]
    z[0] = 0.5 * (z[0] + z[0].transpose(-1, -2)).contiguous()
    z[2] = 0.5 * (z[2] + z[2].transpose(-1, -2)).contiguous()
    p = lambda t: t.data_ptr()
    ids = torch.tensor([7, 1, 4], dtype=torch.int32, device="cuda")
    act = torch.ones(B, dtype=torch.int32, device="cuda")
    act[4] = 0

    def call(lst, cnt, a):
        l = torch.full((B, T, m), 7.0, dtype=torch.float64, device="cuda")
        L = torch.full((B, T, m, n), 7.0, dtype=torch.float64, device="cuda")
        if ddp:
            rc = lib.zm_ddp_backward_list_f64(p(dev[0]), p(dev[1]), p(z[0]), p(z[1]), p(z[2]), *[p(d) for d in dev[2:]],
                                              lst, cnt, a, 0, p(l), p(L), B, T, n, m, None)
        else:
            rc = lib.zm_ilqr_backward_list_f64(*[p(d) for d in dev], lst, cnt, a, 0, p(l), p(L), B, T, n, m, None)
        torch.cuda.synchronize()
        return rc, l, L

    rc, lf, Lf = call(None, 0, None)
    assert rc == 0
    for cnt in (0, 3):
        rc, l, L = call(p(ids), cnt, p(act))
        assert rc == 0
        done = [7, 1] if cnt else []
        for b in range(B):
            if b in done:
                assert torch.equal(l[b], lf[b]) and torch.equal(L[b], Lf[b])
            else:
                assert bool((l[b] == 7.0).all()) and bool((L[b] == 7.0).all())
    assert call(p(ids), B + 1, None)[0] == _lib.ZM_EINVAL
