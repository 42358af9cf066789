"""ctypes loader for the plain-C oracle (TEST INFRASTRUCTURE ONLY; see riccati_oracle.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libzopt_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/riccati_oracle.c with gcc (used by __graft_entry__.build())."""
    src = os.path.join(_HERE, "riccati_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.zo_lqr_backward_f64.argtypes = [dp, dp, dp, dp, dp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int]
        _lib.zo_lqr_backward_f64.restype = ctypes.c_int
        _lib.zo_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def lqr_backward(A, B, Q, R, nthreads: int = 0):
    """A (b,T,n,n), B (b,T,n,m), Q (b,T,n,n), R (b,T,m,m) fp64 -> L (b,T,m,n)."""
    A, B, Q, R = (np.ascontiguousarray(x, dtype=np.float64) for x in (A, B, Q, R))
    b, T, n, m = B.shape
    L = np.empty((b, T, m, n), dtype=np.float64)
    rc = lib().zo_lqr_backward_f64(_p(A), _p(B), _p(Q), _p(R), _p(L), b, T, n, m, nthreads)
    if rc != 0:
        raise ValueError(f"zo_lqr_backward_f64 rc={rc}")
    return L


def num_threads() -> int:
    return int(lib().zo_num_threads())
