"""GPU tests of the Quadcopter mirror (reference quadcopter.py:70-201, tests/test_quadcopter.py:46-125):
dynamics evaluation against the oracle restatement, the reference's trim / linearise tests, batched operating points."""
import numpy as np
import pytest

from oracle import zopt_oracle as zo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ac():
    import torch
    assert torch.cuda.is_available()
    from zopt_amd import models
    return models.Quadcopter()


def test_dynamics_evaluation_matches_oracle_and_reference_kats(ac):
    rng = np.random.default_rng(3)
    x12 = 0.4 * rng.standard_normal((5, 12))
    u = np.array([9.807, 0, 0, 0]) + 0.3 * rng.standard_normal((5, 4))
    wind = np.array([3.0, 1.0, 0.0])
    xd = ac.inertialDynamics(x12, u, wind_ned=wind)
    xr = ac.rigidBodyDynamics(x12[:, :8], u, wind_body=(0.5, -0.2, 0.1))
    for b in range(5):
        assert np.max(np.abs(xd[b] - zo.quad_inertialDynamics(x12[b], u[b], wind_ned=wind))) <= 1e-12
        assert np.max(np.abs(xr[b] - zo.quad_rigidBodyDynamics(x12[b, :8], u[b], wind_body=np.array([0.5, -0.2, 0.1])))) <= 1e-12
    # tests/test_quadcopter.py:46-58: at rest with zero thrust the body falls (wDot = g); hover thrust gives zero
    z = ac.rigidBodyDynamics(np.zeros(8), np.zeros(4))
    assert z == pytest.approx([0, 0, 9.807, 0, 0, 0, 0, 0])
    assert ac.rigidBodyDynamics(np.zeros(8), np.array([9.807, 0, 0, 0])) == pytest.approx(np.zeros(8))


def test_trim_reference_cases_and_batch(ac):
    """tests/test_quadcopter.py:89-99: x0[0:3] == uvw0 and |rigidBodyDynamics(x0, u0)| <= 1e-3."""
    for uvw0 in (np.zeros(3), np.array([0.1, 0.2, 0.3])):
        x0, u0 = ac.trim(uvw0)
        assert x0.shape == (8,) and u0.shape == (4,)
        assert x0[0:3] == pytest.approx(uvw0)
        assert zo.quad_rigidBodyDynamics(x0, u0) == pytest.approx(np.zeros(8), abs=1e-3)
    x0, u0 = ac.trim(np.zeros(3))
    assert x0 == pytest.approx(np.zeros(8), abs=1e-9) and u0 == pytest.approx([9.807, 0, 0, 0], abs=1e-9)   # hover
    rng = np.random.default_rng(0)
    uvw = rng.uniform(-3, 3, (4, 50, 3))                    # a family of operating points in one call
    X, U = ac.trim(uvw)
    assert X.shape == (4, 50, 8) and U.shape == (4, 50, 4) and np.array_equal(X[..., :3], uvw)
    for idx in ((0, 0), (1, 7), (3, 49)):
        assert np.max(np.abs(zo.quad_rigidBodyDynamics(X[idx], U[idx]))) <= 1e-8


def test_linearize_continuous_and_discrete(ac):
    """tests/test_quadcopter.py:102-117 plus values: jax.jacobian of rigidBodyDynamics <-> complex-step Jacobian of the oracle."""
    x0, u0 = np.zeros(8), np.array([9.807, 0, 0, 0])
    A, B = ac.linearize(x0, u0)
    assert A.shape == (8, 8) and B.shape == (8, 4) and not (np.any(np.isnan(A)) or np.any(np.isnan(B)))
    _, Ar, Br = zo.jacobians(zo.quad_rigidBodyDynamics, x0, u0)
    assert np.max(np.abs(A - Ar)) <= 1e-12 and np.max(np.abs(B - Br)) <= 1e-12
    Ad, Bd = ac.linearize(x0, u0, dt=1)
    assert np.max(np.abs(Ad - (np.eye(8) + Ar))) <= 1e-12 and np.max(np.abs(Bd - Br)) <= 1e-12
    rng = np.random.default_rng(1)
    xs = 0.3 * rng.standard_normal((6, 8))
    us = u0 + 0.3 * rng.standard_normal((6, 4))
    As, Bs = ac.linearize(xs, us, dt=0.1)
    for b in range(6):
        _, Ar, Br = zo.jacobians(zo.quad_rigidBodyDynamics, xs[b], us[b])
        assert np.max(np.abs(As[b] - (np.eye(8) + 0.1 * Ar))) <= 1e-12 and np.max(np.abs(Bs[b] - 0.1 * Br)) <= 1e-12


def test_trim_linearise_dare_pipeline(ac):
    """The producer feeding the path on the device: trim at forward flight -> linearise -> DARE gain; the gain stabilises the
    discretised linearisation (spectral radius < 1)."""
    from zopt_amd import lqrUtils
    uvw = np.array([[1.0, 0.0, 0.0], [0.5, -0.5, 0.2]])
    X, U = ac.trim(uvw)
    A, B = ac.linearize(X, U, dt=0.05)
    Q, R = np.broadcast_to(np.eye(8), (2, 8, 8)).copy(), np.broadcast_to(np.eye(4), (2, 4, 4)).copy()
    K = lqrUtils.discreteInfiniteHorizonLqr(A, B, Q, R)
    for b in range(2):
        assert np.max(np.abs(np.linalg.eigvals(A[b] - B[b] @ K[b]))) < 1.0
        Kr, _ = zo.discreteInfiniteHorizonLqr(A[b], B[b], Q[b], R[b])
        assert np.max(np.abs(K[b] - Kr)) <= 1e-9 * np.max(np.abs(Kr))
