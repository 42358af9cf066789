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
    # dtype follows the input arrays (quirk Q8).  fp32 inputs: the small shapes (n <= 12, m <= 4) are computed in fp64 on the
    # tile-16 kernel and rounded once on output; larger ones (n <= 64, m <= 16) run the native fp32 MFMA tile kernel.
    native32 = fp32_in and (n > 12 or m > 4)
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

    Arguments
    ---------
        A : (..., n, n)    B : (..., n, m)    Q : (..., n, n)    R : (..., m, m)     (n <= 12, m <= 4)

    Returns
    -------
        L : (..., m, n) optimal LQR gains `u = -L x`   (with `return_value=True`: (L, V, iterations))
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
    fp32_in = (arr.is_torch(A) and A.dtype == torch.float32) or (not arr.is_torch(A) and np.asarray(A).dtype == np.float32)
    if fp32_in:
        dL, dP = dL.to(torch.float32), dP.to(torch.float32)
    if return_value:
        return arr.result_like(dL, A), arr.result_like(dP, A), arr.result_like(its, A)
    return arr.result_like(dL, A)


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
