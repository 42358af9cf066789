"""Host-side helpers named as in ``zopt.jaxUtils`` (reference jaxUtils.py:7-49) so that code written against the reference imports
unchanged.  Nothing here is on the hot path: `interpMapped` is the per-component linear interpolation `finiteHorizonLqr` builds its
gain schedule from (lqrUtils.py:94-96; `zopt_amd.lqrUtils.finiteHorizonLqr` interpolates its device-resident value function itself),
and the `maybeJit*` decorators have nothing to compile here -- the kernels are already native -- and return the function unchanged.
"""
from __future__ import annotations

import numpy as np


def interpMapped(x, xp, fp, left=None, right=None, period=None):
    """n-dimensional linear interpolation (reference jaxUtils.py:7-24: `jnp.interp` vmapped over the rows of `fp`).

    Arguments
    ---------
        x : scalar or array of query points
        xp : (N,) sorted sample points
        fp : (n, N) values of n functions at the sample points
        left, right, period : as `numpy.interp` (default: clipped at the ends)

    Returns
    -------
        (n,) + shape(x) interpolated values
    """
    fp = np.asarray(fp)
    if fp.ndim != 2:
        raise ValueError("fp must have shape (n, len(xp))")
    return np.stack([np.interp(x, xp, row, left=left, right=right, period=period) for row in fp])


def maybeJitCls(func):
    """Class method decorator of the reference (jaxUtils.py:27-36): jit if `self.jittable`.  Native kernels: returned as is."""
    return func


def maybeJit(func, cond):
    """`jax.jit(func) if cond else func` in the reference (jaxUtils.py:39-41).  Native kernels: returned as is."""
    return func
