"""GPU parity tests of the large-state sweeps (sweep_tiled_f64.hip): `backwardPass_ilqr` (reference ilqrUtils.py:153-181) and
`bilinearAffineLqr` (lqrUtils.py:207-262) beyond the tile-16 shapes -- 12 < n <= 48 or 4 < m <= 16, fp64 MFMA tiles -- against the
oracle at the shapes discreteFiniteHorizonLqr already took: one, two and three 16-wide state tiles, ragged n and m, nonsymmetric
weights and value Hessians (the reference's formulas do not symmetrise), and a case whose Q_uu needs row exchanges."""
import numpy as np
import pytest

from oracle import zopt_oracle as zo
from tests import problems

pytestmark = pytest.mark.gpu
RTOL = 1e-10
SHAPES = [(16, 4, 12, 3), (24, 8, 9, 2), (32, 8, 7, 3), (48, 16, 6, 2), (13, 5, 8, 3), (20, 3, 5, 2), (40, 16, 4, 2), (12, 5, 6, 2),
          (33, 1, 5, 2), (48, 4, 30, 2), (16, 16, 3, 2)]


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import ilqrUtils, lqrUtils
    return ilqrUtils, lqrUtils


def _nonsym_model(batch, T, n, m, seed):
    rng = np.random.default_rng(seed + 7)
    dyn, cost, Vf = problems.random_ilqr_model(batch, T, n, m, seed=seed)
    c, c_x, c_u, c_xx, c_ux, c_uu = cost
    v, v_x, v_xx = Vf
    cost = (c, c_x, c_u, c_xx + 0.2 * rng.standard_normal(c_xx.shape), c_ux, c_uu + 0.2 * rng.standard_normal(c_uu.shape))
    Vf = (v, v_x, v_xx + 0.2 * rng.standard_normal(v_xx.shape))
    return dyn, cost, Vf


def _oracle_ilqr(dyn, cost, Vf):
    ls, Ls = [], []
    for b in range(dyn[1].shape[0]):
        p = zo.backwardPass_ilqr(zo.AffineDynamics(*(x[b] for x in dyn)), zo.QuadraticCostFunction(*(x[b] for x in cost)),
                                 zo.QuadraticValueFunction(*(x[b] for x in Vf)))
        ls.append(p.l)
        Ls.append(p.L)
    return np.stack(ls), np.stack(Ls)


@pytest.mark.parametrize("nonsym", [False, True])
@pytest.mark.parametrize("n,m,T,batch", SHAPES)
def test_backwardPass_ilqr_large_states(mods, n, m, T, batch, nonsym):
    ilqr = mods[0]
    seed = 100 * n + 10 * m + T
    dyn, cost, Vf = _nonsym_model(batch, T, n, m, seed) if nonsym else problems.random_ilqr_model(batch, T, n, m, seed=seed)
    pol = ilqr.backwardPass_ilqr(dyn, cost, Vf)
    lr, Lr = _oracle_ilqr(dyn, cost, Vf)
    assert pol.l.shape == (batch, T, m) and pol.L.shape == (batch, T, m, n)
    assert _rel(pol.L, Lr) <= RTOL and _rel(pol.l, lr) <= RTOL


def test_backwardPass_ilqr_large_states_pivoting_and_torch(mods):
    """A permuted-dominant nonsymmetric c_uu makes the multipliers of the unpivoted elimination exceed the growth bound: the wave
    falls back to LU with partial pivoting in LDS -- the arithmetic jnp.linalg.solve performs."""
    import torch
    ilqr = mods[0]
    n, m = 24, 8
    dyn, cost, Vf = problems.random_ilqr_model(4, 6, n, m, seed=11)
    c, c_x, c_u, c_xx, c_ux, c_uu = cost
    P = np.eye(m)[[2, 0, 3, 1, 7, 4, 6, 5]]
    cost = (c, c_x, c_u, c_xx, c_ux, c_uu @ P * 30.0)
    lr, Lr = _oracle_ilqr(dyn, cost, Vf)
    t = lambda tup: tuple(torch.as_tensor(np.asarray(x), device="cuda") for x in tup)
    pol = ilqr.backwardPass_ilqr(t(dyn), t(cost), t(Vf))
    assert pol.L.is_cuda
    assert _rel(pol.L.cpu().numpy(), Lr) <= 1e-9 and _rel(pol.l.cpu().numpy(), lr) <= 1e-9


def _affine_problem(batch, T, n, m, seed, nonsym):
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


@pytest.mark.parametrize("nonsym", [False, True])
@pytest.mark.parametrize("n,m,T,batch", SHAPES)
def test_bilinearAffineLqr_large_states(mods, n, m, T, batch, nonsym):
    lqr = mods[1]
    args = _affine_problem(batch, T, n, m, seed=1000 * n + 10 * m + T, nonsym=nonsym)
    L, l = lqr.bilinearAffineLqr(*args, T)
    Lr, lr = zo.bilinearAffineLqr(*args, T)
    assert L.shape == (batch, T, m, n) and l.shape == (batch, T, m)
    assert _rel(L, Lr) <= RTOL and _rel(l, lr) <= RTOL


def test_shapes_beyond_the_tile_kernels_are_refused(mods):
    ilqr, lqr = mods
    with pytest.raises(ValueError):
        lqr.bilinearAffineLqr(*_affine_problem(1, 2, 49, 4, 1, False), 2)
    with pytest.raises(ValueError):
        ilqr.backwardPass_ilqr(*problems.random_ilqr_model(1, 2, 16, 17, seed=1))


@pytest.mark.parametrize("n,m,batch", [(16, 4, 3), (20, 6, 2), (33, 9, 2), (48, 16, 2)])
def test_riccatiStep_ilqr_large_states_returns_the_value_function(mods, n, m, batch):
    """`riccatiStep_ilqr` (reference ilqrUtils.py:153-173) beyond the one-tile shapes: the new value function (v, v_x, v_xx) AND the
    policy of one step, nonsymmetric Hessians included."""
    from zopt_amd import pytrees as pt
    ilqr = mods[0]
    dyn, cost, Vf = _nonsym_model(batch, 1, n, m, seed=77 + n)
    sq = lambda t: tuple(x[:, 0] if x.ndim > 1 and x.shape[1] == 1 else x for x in t)   # one time step per item: drop the T = 1 axis
    dyn1, cost1 = sq(dyn), sq(cost)
    val, pol = ilqr.riccatiStep_ilqr(pt.AffineDynamics(*dyn1), pt.QuadraticCostFunction(*cost1), pt.QuadraticValueFunction(*Vf))
    for b in range(batch):
        rv, rp = zo.riccatiStep_ilqr(zo.AffineDynamics(*(x[b] for x in dyn1)), zo.QuadraticCostFunction(*(x[b] for x in cost1)),
                                     zo.QuadraticValueFunction(*(x[b] for x in Vf)))
        assert _rel(pol.L[b], rp.L) <= RTOL and _rel(pol.l[b], rp.l) <= RTOL
        assert val.v[b] == pytest.approx(rv.v, rel=1e-10, abs=1e-12)
        assert _rel(val.v_x[b], rv.v_x) <= RTOL and _rel(val.v_xx[b], rv.v_xx) <= RTOL


def _ddp_model(batch, T, n, m, seed):
    dyn, cost, Vf = problems.random_ilqr_model(batch, T, n, m, seed=seed)
    rng = np.random.default_rng(seed + 1)
    sym = lambda X: 0.5 * (X + np.swapaxes(X, -1, -2))
    f_xx = 0.2 * sym(rng.standard_normal((batch, T, n, n, n))) / np.sqrt(n)
    f_ux = 0.2 * rng.standard_normal((batch, T, n, m, n)) / np.sqrt(n)
    f_uu = 0.2 * sym(rng.standard_normal((batch, T, n, m, m))) / np.sqrt(n)
    return dyn + (f_xx, f_ux, f_uu), cost, Vf


@pytest.mark.parametrize("n,m,T,batch", [(16, 4, 5, 3), (13, 2, 4, 2), (20, 6, 4, 2), (33, 9, 3, 2), (48, 16, 2, 2), (8, 6, 4, 2)])
def test_backwardPass_ddp_large_states(mods, n, m, T, batch):
    """`backwardPass_ddp` (reference ilqrUtils.py:184-214, 237-251) beyond the one-tile DDP sweep: per time step a torch contraction
    `sum_i v_x[i] d2f_i`, the HIP PD projection of the stacked (n+m)^2 matrix (one tile up to 16, multi-tile beyond) and the HIP tile sweep
    for one step with the projected blocks added to the cost Hessians; against the oracle (eigh) at 1e-9."""
    ilqr = mods[0]
    dyn, cost, Vf = _ddp_model(batch, T, n, m, seed=11 * n + m)
    pol = ilqr.backwardPass_ddp(dyn, cost, Vf)
    for b in range(batch):
        ref = zo.backwardPass_ddp(zo.QuadraticDynamics(*(x[b] for x in dyn)), zo.QuadraticCostFunction(*(x[b] for x in cost)),
                                  zo.QuadraticValueFunction(*(x[b] for x in Vf)))
        assert _rel(pol.L[b], ref.L) <= 1e-9 and _rel(pol.l[b], ref.l) <= 1e-9


def test_riccatiStep_ddp_and_conditionQuadraticDynamics_large_states(mods):
    from zopt_amd import pytrees as pt
    ilqr = mods[0]
    n, m, batch = 20, 6, 3
    dyn, cost, Vf = _ddp_model(batch, 1, n, m, seed=5)
    sq = lambda t: tuple(x[:, 0] for x in t)
    dyn1, cost1 = sq(dyn), sq(cost)
    val, pol = ilqr.riccatiStep_ddp(pt.QuadraticDynamics(*dyn1), pt.QuadraticCostFunction(*cost1), pt.QuadraticValueFunction(*Vf))
    blocks = ilqr.conditionQuadraticDynamics(pt.QuadraticDynamics(*dyn1), Vf[1])
    for b in range(batch):
        rv, rp = zo.riccatiStep_ddp(zo.QuadraticDynamics(*(x[b] for x in dyn1)), zo.QuadraticCostFunction(*(x[b] for x in cost1)),
                                    zo.QuadraticValueFunction(*(x[b] for x in Vf)))
        assert _rel(pol.L[b], rp.L) <= 1e-9 and _rel(pol.l[b], rp.l) <= 1e-9
        assert val.v[b] == pytest.approx(rv.v, rel=1e-9, abs=1e-11) and _rel(val.v_x[b], rv.v_x) <= 1e-9 and _rel(val.v_xx[b], rv.v_xx) <= 1e-9
        rb = zo.conditionQuadraticDynamics(zo.QuadraticDynamics(*(x[b] for x in dyn1)), Vf[1][b])
        for got, ref in zip(blocks, rb):
            assert _rel(got[b], ref) <= 2e-11
    with pytest.raises(ValueError):
        ilqr.backwardPass_ddp(*_ddp_model(1, 2, 49, 2, seed=1))


def test_generic_ddp_beyond_the_one_tile_shapes(mods):
    """`differentialDynamicProgramming` with torch callables at n = 14, m = 5: expansions incl. second derivatives by torch.func, the
    backward pass through the per-step large-shape path; against the oracle's DDP loop."""
    import torch
    ilqr = mods[0]
    n, m, N = 14, 5, 8
    rng = np.random.default_rng(4)
    K = 0.15 * rng.standard_normal((n, n)) / np.sqrt(n)
    Bm = 0.3 * rng.standard_normal((n, m))
    dt = 0.1
    tK, tB = torch.as_tensor(K, device="cuda"), torch.as_tensor(Bm, device="cuda")
    cK, cB = torch.as_tensor(K), torch.as_tensor(Bm)
    f_np = lambda x, u: x + dt * (K @ x - 0.05 * x ** 3 + Bm @ u)
    f_gpu = lambda x, u: x + dt * (tK @ x - 0.05 * x ** 3 + tB @ u)
    f_cpu = lambda x, u: x + dt * (cK @ x - 0.05 * x ** 3 + cB @ u)          # the oracle differentiates a CPU torch restatement
    Q, R, Qf = np.eye(n), 0.5 * np.eye(m), 5 * np.eye(n)
    tQ, tR, tQf = (torch.as_tensor(M, device="cuda") for M in (Q, R, Qf))
    x0 = rng.uniform(-1.5, 1.5, (2, n))
    ug = np.zeros((2, N, m))
    traj, L, J, conv = ilqr.differentialDynamicProgramming(f_gpu, lambda x, u: x @ tQ @ x + u @ tR @ u, lambda x: x @ tQf @ x, x0, ug)
    for i in range(2):
        rt, rL, rJ, rc = zo.differentialDynamicProgramming(f_np, f_cpu, Q, R, Qf, x0[i], ug[i])
        assert bool(conv[i]) == rc and J[i] == pytest.approx(rJ, rel=1e-7)
        assert _rel(traj.uTraj[i], rt.uTraj) <= 1e-5 and _rel(L[i], rL) <= 1e-4
