"""The generated closed-form derivatives of the quadcopter (zopt_amd/csrc/quad_derivs_gen.h, tools/gen_quad_derivs.py) against the
oracle's model on the CPU: first derivatives vs complex-step differentiation of oracle.quad_inertialDynamics, second derivatives
vs torch autograd of the oracle's torch restatement -- with and without wind; plus: the header in the tree is what the generator
produces (nobody edited one without the other)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zopt_oracle as zo  # noqa: E402

PAIRS = [0x00, 0x11, 0x22, 0x24, 0x15, 0x05, 0x23, 0x13, 0x04, 0x66, 0x67, 0x77, 0x68, 0x78, 0x88, 0x06, 0x07, 0x08, 0x16, 0x17, 0x18,
         0x26, 0x27, 0x28, 0x46, 0x47, 0x56, 0x57]


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = tmp_path_factory.mktemp("quad_derivs") / "quad_derivs_shim.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", str(so),
                    os.path.join(ROOT, "tests", "quad_derivs_shim.cpp")], check=True)
    lib = ctypes.CDLL(str(so))
    dp = ctypes.POINTER(ctypes.c_double)
    lib.quad_jacobian.argtypes = [dp, dp, dp, ctypes.c_int, dp]
    lib.quad_hessian_pairs.argtypes = [dp, dp, dp, ctypes.c_int, ctypes.c_int, dp]
    lib.quad_jacobian_packed.argtypes = [dp, dp, dp, ctypes.c_int, ctypes.c_double, dp, ctypes.POINTER(ctypes.c_ubyte)]
    lib.quad_jacobian_packed.restype = ctypes.c_int
    lib.quad_hessian_sparse.argtypes = [dp, dp, dp, ctypes.c_int, ctypes.c_double, dp, ctypes.POINTER(ctypes.c_ushort)]
    lib.quad_hessian_sparse.restype = ctypes.c_int
    lib.quad_all_packed.argtypes = [dp, dp, dp, ctypes.c_int, ctypes.c_double, dp, dp]
    lib.quad_all_packed.restype = None
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _points(seed, count):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        x = rng.standard_normal(12) * np.array([3, 3, 3, 1, 1, 1, 0.6, 0.6, 2.0, 5, 5, 5])
        u = np.array([9.807, 0, 0, 0]) + rng.standard_normal(4)
        yield x, u


@pytest.mark.parametrize("wind", [(0.0, 0.0, 0.0), (3.0, 1.0, -0.5)])
def test_first_derivatives_match_complex_step(shim, wind):
    w = np.array(wind)
    for x, u in _points(1, 20):
        ref = _complex_step_jacobian(x, u, w)
        for still_air in ([0, 1] if not np.any(w) else [0]):      # w = 0: the general form and the still-air form
            J = np.zeros((12, 16))
            shim.quad_jacobian(_p(x), _p(u), _p(w), still_air, _p(J))
            assert np.max(np.abs(J - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


def _complex_step_jacobian(x, u, w):
    ref = np.zeros((12, 16))
    h = 1e-30
    for j in range(16):
        z = np.concatenate([x, u]).astype(complex)
        z[j] += 1j * h
        ref[:, j] = np.imag(zo.quad_inertialDynamics(z[:12], z[12:], w)) / h
    return ref


def test_second_derivatives_match_autograd_and_nothing_else_is_nonzero(shim):
    import torch
    w = np.zeros(3)
    f = zo.quad_euler_step_torch(1.0)                          # x + 1.0 * xd: the second derivatives of xd
    for x, u in _points(2, 6):
        H = np.zeros((len(PAIRS), 12))
        H1 = np.zeros((len(PAIRS), 12))
        shim.quad_hessian_pairs(_p(x), _p(u), _p(w), 1, len(PAIRS), _p(H))
        shim.quad_hessian_pairs(_p(x), _p(u), _p(w), 0, len(PAIRS), _p(H1))
        assert np.max(np.abs(H - H1)) <= 1e-13 * max(1.0, np.max(np.abs(H)))      # still-air form == general form at w = 0
        z = torch.tensor(np.concatenate([x, u]), dtype=torch.float64)
        hes = torch.stack([torch.autograd.functional.hessian(lambda zz, i=i: f(zz[:12], zz[12:])[i], z) for i in range(12)]).numpy()
        seen = np.zeros((16, 16), bool)
        for p, ab in enumerate(PAIRS):
            a, b = ab >> 4, ab & 15
            seen[a, b] = seen[b, a] = True
            assert np.max(np.abs(H[p] - hes[:, a, b])) <= 1e-11 * max(1.0, np.max(np.abs(hes))), (p, a, b)
        assert np.max(np.abs(hes[:, ~seen])) == 0.0          # every undeclared pair: exactly zero


def test_second_derivatives_with_wind_match_differences_of_the_jacobian(shim):
    """the oracle's torch model has no wind: central differences of the complex-step Jacobian instead (error ~ h^2)"""
    w = np.array([3.0, 1.0, -0.5])
    h = 1e-5
    for x, u in _points(3, 4):
        H = np.zeros((len(PAIRS), 12))
        shim.quad_hessian_pairs(_p(x), _p(u), _p(w), 0, len(PAIRS), _p(H))
        z = np.concatenate([x, u])
        full = np.zeros((12, 16, 16))
        for b in range(16):
            zp, zm = z.copy(), z.copy()
            zp[b] += h
            zm[b] -= h
            full[:, :, b] = (_complex_step_jacobian(zp[:12], zp[12:], w) - _complex_step_jacobian(zm[:12], zm[12:], w)) / (2 * h)
        seen = np.zeros((16, 16), bool)
        scale = max(1.0, np.max(np.abs(full)))
        for p, ab in enumerate(PAIRS):
            a, b = ab >> 4, ab & 15
            seen[a, b] = seen[b, a] = True
            assert np.max(np.abs(H[p] - full[:, a, b])) <= 1e-7 * scale, (p, a, b)
        assert np.max(np.abs(full[:, ~seen])) <= 1e-7 * scale


@pytest.mark.parametrize("wind", [(0.0, 0.0, 0.0), (3.0, 1.0, -0.5)])
def test_packed_jacobian_image_rebuilds_the_full_matrices(shim, wind):
    """quad_jac_column_packed + QUAD_JPOS_*: the packed entries at their positions and the identity everywhere else are
    [f_x | f_u] = I + dt d xd / d z (the form the expansion kernel writes and the sweep kernels read)"""
    w, dt = np.array(wind), 0.1
    still = 0 if np.any(w) else 1
    for x, u in _points(5, 8):
        t = np.full(64, np.nan)
        pos = np.zeros(192, dtype=np.uint8)
        nj = shim.quad_jacobian_packed(_p(x), _p(u), _p(w), still, dt, _p(t), pos.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
        pos = pos.reshape(12, 16)
        assert nj == (56 if still else 59) and sorted(pos[pos != 255].tolist()) == list(range(nj)) and np.all(np.isfinite(t[:nj]))
        F = np.hstack([np.eye(12), np.zeros((12, 4))])
        F[pos != 255] = t[pos[pos != 255]]
        ref = np.hstack([np.eye(12), np.zeros((12, 4))]) + dt * _complex_step_jacobian(x, u, w)
        assert np.max(np.abs(F - ref)) <= 1e-13 * max(1.0, np.abs(ref).max())
        assert np.all((ref != np.hstack([np.eye(12), np.zeros((12, 4))])) <= (pos != 255))     # nothing outside the packed set moves


@pytest.mark.parametrize("wind", [(0.0, 0.0, 0.0), (3.0, 1.0, -0.5)])
def test_sparse_second_derivative_image_rebuilds_the_dense_rows(shim, wind):
    """quad_hess_pair2_packed + QUAD_HDENSE_*: the packed entries scattered to pair * 12 + i, zeros elsewhere, are dt times the dense
    rows of quad_hess_pair (the form the DDP expansion writes and the sweep scatters into its LDS image)"""
    w, dt = np.array(wind), 0.1
    still = 0 if np.any(w) else 1
    for x, u in _points(6, 6):
        t = np.full(96, np.nan)
        dense = np.zeros(96, dtype=np.uint16)
        nh = shim.quad_hessian_sparse(_p(x), _p(u), _p(w), still, dt, _p(t), dense.ctypes.data_as(ctypes.POINTER(ctypes.c_ushort)))
        assert nh == (69 if still else 85) and len(set(dense[:nh].tolist())) == nh and np.all(np.isfinite(t[:nh]))
        H = np.zeros((len(PAIRS), 12))
        shim.quad_hessian_pairs(_p(x), _p(u), _p(w), still, len(PAIRS), _p(H))
        R = np.zeros(len(PAIRS) * 12)
        R[dense[:nh]] = t[:nh]
        assert np.array_equal(R.reshape(len(PAIRS), 12), dt * H)          # same expressions, same bits; and nothing nonzero is left out


@pytest.mark.parametrize("wind", [(0.0, 0.0, 0.0), (3.0, 1.0, -0.5)])
@pytest.mark.parametrize("dt", [0.1, 0.0])
def test_straight_line_forms_equal_the_per_column_and_per_pair_forms_bit_for_bit(shim, wind, dt):
    """quad_jac_all_packed / quad_hess_all_packed (one lane evaluates a whole point: expand_quad_points_kernel,
    quad_hessian_points_kernel) write the images of quad_jac_column_packed / quad_hess_pair2_packed: the generator spells every
    multiply-add out and switches the compiler's contraction off, so the two forms round identically"""
    w = np.array(wind)
    still = 0 if np.any(w) else 1
    for x, u in _points(7, 10):
        t = np.full(64, np.nan)
        pos = np.zeros(192, dtype=np.uint8)
        nj = shim.quad_jacobian_packed(_p(x), _p(u), _p(w), still, dt, _p(t), pos.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
        th = np.full(96, np.nan)
        dense = np.zeros(96, dtype=np.uint16)
        nh = shim.quad_hessian_sparse(_p(x), _p(u), _p(w), still, dt, _p(th), dense.ctypes.data_as(ctypes.POINTER(ctypes.c_ushort)))
        aj, ah = np.full(64, np.nan), np.full(96, np.nan)
        shim.quad_all_packed(_p(x), _p(u), _p(w), still, dt, _p(aj), _p(ah))
        assert np.array_equal(aj[:nj], t[:nj]) and np.all(np.isnan(aj[nj:]))
        assert np.array_equal(ah[:nh], th[:nh]) and np.all(np.isnan(ah[nh:]))


def test_regenerating_the_header_reproduces_the_file_in_the_tree(tmp_path):
    """tools/gen_quad_derivs.py --out <tmp>: byte for byte zopt_amd/csrc/quad_derivs_gen.h (nobody edited one without the other)"""
    out = tmp_path / "quad_derivs_gen.h"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_quad_derivs.py"), "--out", str(out)], check=True, timeout=600,
                   capture_output=True)
    assert open(out).read() == open(os.path.join(ROOT, "zopt_amd", "csrc", "quad_derivs_gen.h")).read()


def test_header_is_what_the_generator_writes(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_quad_derivs", os.path.join(ROOT, "tools", "gen_quad_derivs.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.PAIRS == PAIRS
    src = open(os.path.join(ROOT, "zopt_amd", "csrc", "models.h")).read()
    for ab in PAIRS:                                         # the generator's pair table is models.h's
        assert f"0x{ab:02x}" in src.lower()
