"""Drop-in for ``zopt.lqrUtils`` (hot-path functions), running on MI355X HIP kernels.

Same names, positional arguments and return shapes as the reference; arrays may carry extra
LEADING batch axes (new).  NumPy in -> NumPy out; torch ROCm tensors in -> torch ROCm tensors out
(asynchronous on the current stream).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _arrays as arr
from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _shape_error(msg):
    raise ValueError(msg)


def discreteFiniteHorizonLqr(A, B, Q, R, N):
    """Finite-horizon discrete LQR gains by backward Riccati recursion (reference lqrUtils.py:144-173).

    ```
    J = sum_k(x^T Q x + u^T R u);   xNew = A x + B u;   uLqr = -L x
    ```

    Arguments
    ---------
        A : (..., N, n, n) state matrices, time along axis -3: `A[k]`
        B : (..., N, n, m)
        Q : (..., N, n, n)   (terminal value is `Q[-1]`, as in the reference, lqrUtils.py:172)
        R : (..., N, m, m)
        N : horizon

    Returns
    -------
        L : (..., N, m, n) optimal gains `L[k]`
    """
    shp = tuple(np.shape(B)) if not arr.is_torch(B) else tuple(B.shape)
    if len(shp) < 3:
        _shape_error("B must have shape (..., N, n, m)")
    n, m = shp[-2:]
    lead = shp[:-3]
    for name, X, tail in (("A", A, (n, n)), ("B", B, (n, m)), ("Q", Q, (n, n)), ("R", R, (m, m))):
        s = tuple(X.shape) if hasattr(X, "shape") else tuple(np.shape(X))
        if s[-2:] != tail or len(s) < 3 or s[-3] < N or s[:-3] != lead:
            _shape_error(f"{name} has shape {s}, expected {lead + ('>=N',) + tail} with N={N}")
    if N < 1:
        _shape_error("N must be >= 1")
    fp32_in = (arr.is_torch(A) and A.dtype == torch.float32) or (not arr.is_torch(A) and np.asarray(A).dtype == np.float32)
    # dtype follows the input arrays (quirk Q8).  fp32 inputs go to zm_lqr_backward_f32 as they are: the fast-path shapes (n in {8, 12},
    # m = 4) run K1 on fp32 storage with fp64 arithmetic, larger ones (n <= 64, m <= 16) the fp32 MFMA tile kernel; the remaining small
    # shapes are computed in fp64 on the tile-16 kernel and rounded once on output.
    native32 = fp32_in and (n > 12 or m > 4 or ((n in (8, 12)) and m == 4))
    dt = torch.float32 if native32 else torch.float64
    dev = [arr.to_device(X, dt) for X in (A, B, Q, R)]
    batch = 1
    for d in lead:
        batch *= int(d)
    out_dt = torch.float32 if fp32_in else torch.float64
    if batch == 0:
        return arr.result_like(torch.empty(lead + (N, m, n), dtype=out_dt, device=dev[0].device), A)
    # The reference scans xs = arange(N) over the first N steps but starts from V = Q[-1] of the WHOLE array
    # (lqrUtils.py:172).  The kernel takes the last step's Q as terminal value, so over-long inputs get one extra step
    # (A = 0, B = 0, Q = Q[-1], R = I): it returns L = 0 and hands V = Q[-1] to step N-1.
    Tk = N
    if any(x.shape[-3] != N for x in dev):
        dA, dB, dQ, dR = dev
        Tk = N + 1
        eye = torch.eye(m, dtype=dt, device=dA.device).expand(lead + (1, m, m))
        dA = torch.cat([dA[..., :N, :, :], torch.zeros_like(dA[..., :1, :, :])], dim=-3).contiguous()
        dB = torch.cat([dB[..., :N, :, :], torch.zeros_like(dB[..., :1, :, :])], dim=-3).contiguous()
        dQ = torch.cat([dQ[..., :N, :, :], dQ[..., -1:, :, :]], dim=-3).contiguous()
        dR = torch.cat([dR[..., :N, :, :], eye], dim=-3).contiguous()
    else:
        dA, dB, dQ, dR = dev
    dL = torch.empty(lead + (Tk, m, n), dtype=dt, device=dA.device)
    fn = _lib.lib().zm_lqr_backward_f32 if native32 else _lib.lib().zm_lqr_backward_f64
    rc = fn(dA.data_ptr(), dB.data_ptr(), dQ.data_ptr(), dR.data_ptr(), dL.data_ptr(), batch, Tk, n, m,
            ctypes.c_void_p(arr.stream_ptr(dA)))
    _lib.check(rc, "discreteFiniteHorizonLqr")
    if Tk != N:
        dL = dL[..., :N, :, :].contiguous()
    if fp32_in and not native32:
        dL = dL.to(torch.float32)
    return arr.result_like(dL, A)


def discreteInfiniteHorizonLqr(A, B, Q, R, tol=1e-14, maxIter=200000, return_value=False):
    """Discrete-time infinite-horizon LQR gains (reference lqrUtils.py:176-204: SciPy `solve_discrete_are` + one solve).

    ```
    J = sum_k(x^T Q x + u^T R u);   xNew = A x + B u;   uLqr = -L x
    ```
    Here: Riccati value iteration from V = Q on the GPU, one wave per system, until `max|V' - V| <= tol * max|V'|`.
    Accuracy: the fixed point is SciPy's stabilising solution for stabilizable / detectable designs (tested to 1e-10 against
    `solve_discrete_are`); value iteration converges linearly at the closed loop's spectral radius squared, so marginally
    stabilizable designs need many iterations.  Like SciPy (`LinAlgError`), a design whose iteration has not converged within
    `maxIter` steps or whose gain is not finite (not stabilizable) raises `numpy.linalg.LinAlgError` instead of returning a
    non-stationary gain (with `return_value=True` nothing is raised: the caller gets `iterations` and decides).

    Arguments
    ---------
        A : (..., n, n)    B : (..., n, m)    Q : (..., n, n)    R : (..., m, m)     (n <= 64, m <= 16: the tile-16 register
            kernel up to n <= 12, m <= 4, the fp64 MFMA tile kernel on time-invariant operands beyond -- it stops when the GAIN no
            longer changes, `max|L_k - L_{k-1}| <= tol max|L_k|`)

    Returns
    -------
        L : (..., m, n) optimal LQR gains `u = -L x`   (with `return_value=True`: (L, V, iterations), iterations < 0 where the
            cap ended the iteration before its stopping test was met)
    """
    shp = tuple(B.shape) if hasattr(B, "shape") else tuple(np.shape(B))
    if len(shp) < 2:
        _shape_error("B must have shape (..., n, m)")
    n, m = shp[-2:]
    lead = shp[:-2]
    for name, X, tail in (("A", A, (n, n)), ("B", B, (n, m)), ("Q", Q, (n, n)), ("R", R, (m, m))):
        s_ = tuple(X.shape) if hasattr(X, "shape") else tuple(np.shape(X))
        if s_ != lead + tail:
            _shape_error(f"{name} has shape {s_}, expected {lead + tail}")
    dt = torch.float64
    dA, dB, dQ, dR = [arr.to_device(X, dt) for X in (A, B, Q, R)]
    batch = 1
    for d in lead:
        batch *= int(d)
    dL = torch.empty(lead + (m, n), dtype=dt, device=dA.device)
    dP = torch.empty(lead + (n, n), dtype=dt, device=dA.device)
    its = torch.empty(lead, dtype=torch.int32, device=dA.device)
    rc = 0 if batch == 0 else _lib.lib().zm_dare_f64(dA.data_ptr(), dB.data_ptr(), dQ.data_ptr(), dR.data_ptr(), dL.data_ptr(), dP.data_ptr(),
                                its.data_ptr(), batch, n, m, float(tol), int(maxIter), ctypes.c_void_p(arr.stream_ptr(dA)))
    _lib.check(rc, "discreteInfiniteHorizonLqr")
    if batch and not return_value:      # with return_value=True the caller receives `iterations` and judges convergence itself
        # the kernel reports convergence explicitly (iters < 0: the cap ended the loop); a design that meets the tolerance exactly on
        # its last allowed iteration is converged
        bad = (its.reshape(-1) <= 0) | ~torch.isfinite(dL.reshape(batch, -1)).all(dim=1)
        nbad = int(bad.sum())
        if nbad:   # SciPy's solve_discrete_are raises LinAlgError('Failed to find a finite solution.') there (lqrUtils.py:202)
            raise np.linalg.LinAlgError(f"discreteInfiniteHorizonLqr: no converged finite solution for {nbad} of {batch} designs "
                                        f"within maxIter={int(maxIter)} (not stabilizable, or increase maxIter)")
    fp32_in = (arr.is_torch(A) and A.dtype == torch.float32) or (not arr.is_torch(A) and np.asarray(A).dtype == np.float32)
    if fp32_in:
        dL, dP = dL.to(torch.float32), dP.to(torch.float32)
    if return_value:
        return arr.result_like(dL, A), arr.result_like(dP, A), arr.result_like(its, A)
    return arr.result_like(dL, A)


def _check_symmetric(name, X):
    """scipy.linalg.solve_continuous_are rejects nonsymmetric q / r with a ValueError (its _are_validate_args)."""
    if X.numel() and float((X - X.transpose(-1, -2)).abs().max()) > 100 * np.finfo(np.float64).eps * max(float(X.abs().max()), 1.0):
        _shape_error(f"Matrix {name} should be symmetric/hermitian.")


def infiniteHorizonLqr(A, B, Q, R, tol=1e-14, maxIter=60, return_value=False):
    """Continuous-time infinite-horizon LQR gains (reference lqrUtils.py:13-36: SciPy `solve_continuous_are` + one solve).

    ```
    J = int_t(x^T Q x + u^T R u);   xDot = A x + B u;   uLqr = -K x
    ```
    Here: the stabilising solution of the algebraic Riccati equation by the structure-preserving doubling algorithm on the GPU,
    one wave per design (quadratically convergent: ~10 doubling steps), `K = R^-1 B^T P`.
    Accuracy against SciPy's Schur-based solver (tools/fuzz_care.py, 1272 random designs, n <= 16): median deviation 1e-14, 95 %
    below 1e-9; the rest are weakly controllable designs, where `P` is only determined to cond * eps and the doubling iteration's
    residual is up to ~100x SciPy's.  Designs without a finite stabilising solution raise `numpy.linalg.LinAlgError` as SciPy does.

    Arguments
    ---------
        A : (..., n, n)    B : (..., n, m)    Q : (..., n, n)    R : (..., m, m)     (n <= 16, m <= 16; Q, R symmetric)

    Returns
    -------
        K : (..., m, n) LQR gains   (with `return_value=True`: (K, P, doubling steps))
    """
    shp = tuple(B.shape) if hasattr(B, "shape") else tuple(np.shape(B))
    if len(shp) < 2:
        _shape_error("B must have shape (..., n, m)")
    n, m = shp[-2:]
    lead = shp[:-2]
    for name, X, tail in (("A", A, (n, n)), ("B", B, (n, m)), ("Q", Q, (n, n)), ("R", R, (m, m))):
        s_ = tuple(X.shape) if hasattr(X, "shape") else tuple(np.shape(X))
        if s_ != lead + tail:
            _shape_error(f"{name} has shape {s_}, expected {lead + tail}")
    dt = torch.float64
    dA, dB, dQ, dR = [arr.to_device(X, dt) for X in (A, B, Q, R)]
    _check_symmetric("q", dQ)
    _check_symmetric("r", dR)
    batch = 1
    for d in lead:
        batch *= int(d)
    dK = torch.empty(lead + (m, n), dtype=dt, device=dA.device)
    dP = torch.empty(lead + (n, n), dtype=dt, device=dA.device)
    info = torch.ones(lead, dtype=torch.int32, device=dA.device)
    rc = 0 if batch == 0 else _lib.lib().zm_care_f64(dA.data_ptr(), dB.data_ptr(), dQ.data_ptr(), dR.data_ptr(), dK.data_ptr(),
                                                     dP.data_ptr(), info.data_ptr(), batch, n, m, float(tol), int(maxIter),
                                                     ctypes.c_void_p(arr.stream_ptr(dA)))
    _lib.check(rc, "infiniteHorizonLqr")
    if batch and int(info.min()) < 0:    # SciPy: LinAlgError('Failed to find a finite solution.')
        raise np.linalg.LinAlgError("infiniteHorizonLqr: failed to find a finite stabilising solution "
                                    f"({int((info < 0).sum())} of {batch} designs)")
    fp32_in = (arr.is_torch(A) and A.dtype == torch.float32) or (not arr.is_torch(A) and np.asarray(A).dtype == np.float32)
    if fp32_in:
        dK, dP = dK.to(torch.float32), dP.to(torch.float32)
    if return_value:
        return arr.result_like(dK, A), arr.result_like(dP, A), arr.result_like(info, A)
    return arr.result_like(dK, A)


def infiniteHorizonIntegralLqr(A, B, Q, R, Qi, Ci):
    """Infinite-horizon LQR with integral states (reference lqrUtils.py:101-141).

    ```
    J = int_t(z^T Q_i z + x^T Q x + u^T R u);   zDot = C_i x;   xDot = A x + B u
    ```
    The integral-augmented system `[[0, Ci], [0, A]]`, `[[0], [B]]`, `blkdiag(Qi, Q)` (:129-132) goes through `infiniteHorizonLqr`.

    Arguments
    ---------
        A : (..., n, n)   B : (..., n, m)   Q : (..., n, n)   R : (..., m, m)   Qi : (..., ni, ni)
        Ci : (..., ni, n)  -- or (..., n) for a single integral state, as the reference's own test passes it     (ni + n <= 16)

    Returns
    -------
        Ki : (..., m, ni) integral gains      Kp : (..., m, n) proportional gains
    """
    dt = torch.float64
    dA, dB, dQ, dR, dQi, dCi = [arr.to_device(X, dt) for X in (A, B, Q, R, Qi, Ci)]
    if dB.dim() < 2 or dQi.dim() < 2:
        _shape_error("B must have shape (..., n, m) and Qi (..., ni, ni)")
    n, m = dB.shape[-2:]
    ni = dQi.shape[-1]
    lead = tuple(dB.shape[:-2])
    if dCi.dim() == len(lead) + 1:       # np.block promotes a 1-D Ci to one row (:129)
        dCi = dCi.unsqueeze(-2)
    for name, X, tail in (("A", dA, (n, n)), ("Q", dQ, (n, n)), ("R", dR, (m, m)), ("Qi", dQi, (ni, ni)), ("Ci", dCi, (ni, n))):
        if tuple(X.shape) != lead + tail:
            _shape_error(f"{name} has shape {tuple(X.shape)}, expected {lead + tail}")
    z = lambda r, c: torch.zeros(lead + (r, c), dtype=dt, device=dA.device)
    Aw = torch.cat([torch.cat([z(ni, ni), dCi], -1), torch.cat([z(n, ni), dA], -1)], -2)
    Bw = torch.cat([z(ni, m), dB], -2)
    Qw = torch.cat([torch.cat([dQi, z(ni, n)], -1), torch.cat([z(n, ni), dQ], -1)], -2)
    K = infiniteHorizonLqr(Aw.contiguous(), Bw.contiguous(), Qw.contiguous(), dR)
    return arr.result_like(K[..., :, :ni].contiguous(), A), arr.result_like(K[..., :, ni:].contiguous(), A)


class _GainSchedule:
    """`K(t) = R_inv(t) @ B(t).T @ V(t)` with V linearly interpolated between the N grid values, clipped at the ends
    (reference lqrUtils.py:94-97, jaxUtils.py:7-24).  `t`, `V` hold the grid (N,) and the value function (..., N, n, n)."""

    def __init__(self, t, V, B, R_inv, T):
        self.t, self.V, self._B, self._Ri, self._T = t, V, B, R_inv, float(T)

    def value(self, tq):
        N = self.V.shape[-3]
        if N == 1:
            return self.V[..., 0, :, :]
        u = min(max(float(tq) / self._T, 0.0), 1.0) * (N - 1)
        i0 = min(int(u), N - 2)
        w = u - i0
        return self.V[..., i0, :, :] + w * (self.V[..., i0 + 1, :, :] - self.V[..., i0, :, :])

    def __call__(self, tq):
        B, Ri = self._B(tq), self._Ri(tq)
        V = self.value(tq)
        if arr.is_torch(V):
            B, Ri = arr.to_device(B, V.dtype, V.device), arr.to_device(Ri, V.dtype, V.device)
            return Ri @ B.transpose(-1, -2) @ V
        return np.asarray(Ri) @ np.swapaxes(np.asarray(B), -1, -2) @ V


def finiteHorizonLqr(A, B, Q, R_inv, Qf, T, N=50, n_samples=None, rtol=1.4e-8, atol=1.4e-8, max_steps=100000, coef_rtol=1e-6,
                     max_refinements=7):
    """Continuous-time finite-horizon LQR gains by integrating the LQR Hamilton-Jacobi-Bellman (Riccati) equation
    (reference lqrUtils.py:39-98).

    ```
    J = xf^T Qf xf + int_t(x^T Q x + u^T R u);   xDot = A x + B u;   uLqr = -K(t) x
    ```
    Here: `dV/dt = -Q + V B R_inv B^T V - V A - A^T V` backwards from `V(T) = Qf` on the GPU, one wave per design, with the
    adaptive Dormand-Prince 5(4) pair and tolerances of `jax.experimental.ode.odeint` (the reference's integrator), stepping
    exactly onto the output times `linspace(0, T, N)`.  The coefficient callables are host Python (the reference evaluates them
    inside the integrator's right-hand side): they are sampled at `linspace(0, T, n_samples)` and the kernel interpolates linearly
    in between -- exact for time-invariant coefficients (the reference's demos and tests; a single sample then) and for
    piecewise-linear ones.  For other time-varying coefficients the sampling is REFINED until it no longer matters: starting
    from `8 (N - 1) + 1` samples the spacing is halved (interpolation error / 4 each time) until the value function changes by
    less than `coef_rtol * max|V|` between two refinements (default 1e-6: the level at which two runs of the adaptive integrator
    differ anyway), at most `max_refinements` times (`K.n_samples` / `K.coef_change`
    report what was reached).  An explicit `n_samples` switches the refinement off.

    Arguments
    ---------
        A, B, Q, R_inv : callables of time returning (..., n, n), (..., n, m), (..., n, n), (..., m, m)      (n <= 16, m <= 16)
        Qf : (..., n, n) terminal state cost matrix
        T : time horizon;  N : number of output times

    Returns
    -------
        K : callable `K(t)` -> (..., m, n);  `K.t` (N,) and `K.V` (..., N, n, n) expose the integrated value function
    """
    T = float(T)
    N = int(N)
    if not (T > 0.0) or N < 1:
        _shape_error("T must be > 0 and N >= 1")
    ns0 = int(n_samples) if n_samples is not None else 8 * (N - 1) + 1
    if ns0 < 1:
        _shape_error("n_samples must be >= 1")
    dt = torch.float64
    dev = arr.to_device(Qf, dt).device
    dQf = arr.to_device(Qf, dt, dev).contiguous()

    def integrate(ns):
        ts = np.linspace(0.0, T, ns)

        def sample(f):
            vals = [arr.to_device(f(float(tk)), dt, dev) for tk in ts]
            if all(v.shape == vals[0].shape and bool((v == vals[0]).all()) for v in vals[1:]):
                vals = vals[:1]           # time-invariant
            return vals

        sA, sB, sQ, sRi = sample(A), sample(B), sample(Q), sample(R_inv)
        cnt = max(len(sA), len(sB), len(sQ), len(sRi))
        stack = lambda v: torch.stack(v if len(v) == cnt else v * cnt, dim=-3).contiguous()
        dA, dB, dQ, dRi = stack(sA), stack(sB), stack(sQ), stack(sRi)
        if dB.dim() < 3:
            _shape_error("B(t) must have shape (..., n, m)")
        n, m = dB.shape[-2:]
        lead = tuple(dB.shape[:-3])
        for name, X, tail in (("A(t)", dA, (cnt, n, n)), ("Q(t)", dQ, (cnt, n, n)), ("R_inv(t)", dRi, (cnt, m, m))):
            if tuple(X.shape) != lead + tail:
                _shape_error(f"{name} has shape {tuple(X.shape[:-3]) + tuple(X.shape[-2:])}, expected {lead + tail[1:]}")
        if tuple(dQf.shape) != lead + (n, n):
            _shape_error(f"Qf has shape {tuple(dQf.shape)}, expected {lead + (n, n)}")
        batch = 1
        for d in lead:
            batch *= int(d)
        dV = torch.empty(lead + (N, n, n), dtype=dt, device=dev)
        info = torch.ones(lead, dtype=torch.int32, device=dev)
        rc = 0 if batch == 0 else _lib.lib().zm_riccati_ode_f64(dA.data_ptr(), dB.data_ptr(), dRi.data_ptr(), dQ.data_ptr(),
                                                                dQf.data_ptr(), dV.data_ptr(), info.data_ptr(), batch, n, m, cnt, N,
                                                                T, float(rtol), float(atol), int(max_steps),
                                                                ctypes.c_void_p(arr.stream_ptr(dQf)))
        _lib.check(rc, "finiteHorizonLqr")
        return dV, info, cnt

    ns = ns0
    dV, info, cnt = integrate(ns)
    change = 0.0
    if n_samples is None and cnt > 1 and dV.numel():
        # time-varying coefficients: halve the sample spacing until the linear interpolation no longer shows in V
        for _ in range(int(max_refinements)):
            ns = 2 * (ns - 1) + 1
            dV2, info2, _ = integrate(ns)
            fin = torch.isfinite(dV2) & torch.isfinite(dV)
            scale = float(dV2[fin].abs().max()) if bool(fin.any()) else 0.0
            change = float((dV2 - dV)[fin].abs().max()) if bool(fin.any()) else 0.0
            dV, info = dV2, info2
            if change <= float(coef_rtol) * max(scale, 1e-300):
                break
    K = _GainSchedule(np.linspace(0.0, T, N), arr.result_like(dV, Qf), B, R_inv, T)
    K.info = arr.result_like(info, Qf)
    K.n_samples = ns if cnt > 1 else 1
    K.coef_change = change
    return K


def bilinearAffineLqr(A, B, d, Q, R, H, q, r, q0, N):
    """Finite Horizon LQR with bilinear cost and affine dynamics (reference lqrUtils.py:207-262).

    Arguments
    ---------
        A : (..., N, n, n)    B : (..., N, n, m)    d : (..., N, n)
        Q : (..., N, n, n)    R : (..., N, m, m)    H : (..., N, m, n)
        q : (..., N, n)       r : (..., N, m)       q0 : (..., N)   (does not influence the gains, accepted for parity)
        N : horizon depth

    Returns
    -------
        L : (..., N, m, n) optimal lqr gains `L[k]`
        l : (..., N, m) optimal lqr offsets `l[k]`          (law `u = -L x - l`, demos/bilinearLqrControl.py:14)
    """
    shp = tuple(B.shape) if hasattr(B, "shape") else tuple(np.shape(B))
    if len(shp) < 3:
        _shape_error("B must have shape (..., N, n, m)")
    n, m = shp[-2:]
    lead = shp[:-3]
    spec = (("A", A, (n, n)), ("B", B, (n, m)), ("d", d, (n,)), ("Q", Q, (n, n)), ("R", R, (m, m)), ("H", H, (m, n)),
            ("q", q, (n,)), ("r", r, (m,)))
    for name, X, tail in spec:
        s_ = tuple(X.shape) if hasattr(X, "shape") else tuple(np.shape(X))
        if s_[len(s_) - len(tail):] != tail or len(s_) != len(lead) + 1 + len(tail) or s_[:len(lead)] != lead or s_[len(lead)] < N:
            _shape_error(f"{name} has shape {s_}, expected {lead + ('>=N',) + tail} with N={N}")
    if N < 1:
        _shape_error("N must be >= 1")
    dt = torch.float64
    dev = [arr.to_device(X, dt) for name, X, tail in spec]
    batch = 1
    for s_ in lead:
        batch *= int(s_)
    fp32_in = (arr.is_torch(A) and A.dtype == torch.float32) or (not arr.is_torch(A) and np.asarray(A).dtype == np.float32)
    if batch == 0:
        odt = torch.float32 if fp32_in else dt
        return (arr.result_like(torch.empty(lead + (N, m, n), dtype=odt, device=dev[0].device), A),
                arr.result_like(torch.empty(lead + (N, m), dtype=odt, device=dev[0].device), A))
    # Over-long inputs: the reference scans the first N steps from the carry (Q[-1], q[-1], q0[-1]) of the WHOLE arrays
    # (lqrUtils.py:261).  One extra step (A = B = d = H = r = 0, Q = Q[-1], q = q[-1], R = I) reproduces that carry.
    ax = len(lead)
    Tk = N
    if any(t.shape[ax] != N for t in dev):
        Tk = N + 1
        dA, dB, dd, dQ, dR, dH, dq, dr = dev
        head = lambda t: t.narrow(ax, 0, N)
        last = lambda t: t.narrow(ax, t.shape[ax] - 1, 1)
        zero = lambda t: torch.zeros_like(t.narrow(ax, 0, 1))
        eye = torch.eye(m, dtype=dt, device=dA.device).expand(lead + (1, m, m))
        dev = [torch.cat([head(dA), zero(dA)], ax), torch.cat([head(dB), zero(dB)], ax), torch.cat([head(dd), zero(dd)], ax),
               torch.cat([head(dQ), last(dQ)], ax), torch.cat([head(dR), eye], ax), torch.cat([head(dH), zero(dH)], ax),
               torch.cat([head(dq), last(dq)], ax), torch.cat([head(dr), zero(dr)], ax)]
        dev = [t.contiguous() for t in dev]
    dA, dB, dd, dQ, dR, dH, dq, dr = dev
    dL = torch.empty(lead + (Tk, m, n), dtype=dt, device=dA.device)
    dl = torch.empty(lead + (Tk, m), dtype=dt, device=dA.device)
    rc = _lib.lib().zm_lqr_backward_affine_f64(dA.data_ptr(), dB.data_ptr(), dd.data_ptr(), dQ.data_ptr(), dR.data_ptr(),
                                               dH.data_ptr(), dq.data_ptr(), dr.data_ptr(), dL.data_ptr(), dl.data_ptr(),
                                               batch, Tk, n, m, ctypes.c_void_p(arr.stream_ptr(dA)))
    _lib.check(rc, "bilinearAffineLqr")
    if Tk != N:
        dL, dl = dL.narrow(ax, 0, N).contiguous(), dl.narrow(ax, 0, N).contiguous()
    if fp32_in:
        dL, dl = dL.to(torch.float32), dl.to(torch.float32)
    return arr.result_like(dL, A), arr.result_like(dl, A)


def proportionalFeedbackController(x, x0, u0, K):
    """`u = -K (x - x0) + u0` (reference lqrUtils.py:266-269); no controller states."""
    control = -K @ (x - x0) + u0
    dxCtrl = np.array([])
    return control, dxCtrl
