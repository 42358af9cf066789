"""A NumPy-backed stand-in for the handful of `jax` names the reference's hot path uses -- FIXTURE TOOLING ONLY.

zprihoda/zopt is pure Python over jax; jax is not installed in the build container (and cannot be).  The array-level hot
path (`zopt/lqrUtils.py`, `zopt/ilqrUtils.py`, `zopt/pytrees.py`, the model in `zopt/quadcopter.py`) only needs array ops,
`linalg.solve` / `eigh`, `lax.scan` / `while_loop`, `vmap`, `tree.map` -- all of which have an exact NumPy meaning.  This
module registers in-memory modules named `jax`, `jax.numpy`, ... that give them that meaning, so that the REFERENCE'S OWN
SOURCE FILES can be imported from /root/reference and executed on nonsymmetric, time-varying inputs
(tests/golden/make_reference_fixtures.py).  What comes out is labelled "reference source, NumPy semantics": it is the
reference's code and operation order, evaluated by NumPy / LAPACK in IEEE fp64 -- not XLA output.

Semantics that differ between the two libraries and are therefore emulated explicitly:
  * `jnp.linalg.eigh(a)` symmetrises its input, (a + a^H) / 2, by default (`symmetrize_input=True`); NumPy reads one triangle.
  * `lax.scan(..., reverse=True)` visits xs back to front but stacks the per-step outputs in xs order.
  * `jnp.argmin` treats NaN as the minimum (first NaN wins) -- as `numpy.argmin` does.
Autodiff (`grad`, `jacobian`, `hessian`) and `odeint` have no NumPy meaning: they return callables that raise when CALLED
(constructing `Quadcopter()` wraps `jacobian` without calling it).

Nothing here is used by the product, by the `-m gpu` tests or on the GPU box; nothing of the reference is copied.
"""
from __future__ import annotations

import sys
import types

import numpy as np


def _is_namedtuple(x):
    return isinstance(x, tuple) and hasattr(type(x), "_fields")


def tree_map(f, tree, *rest):
    """jax.tree.map for the pytrees the reference uses: (named) tuples, lists, dicts, array leaves."""
    if _is_namedtuple(tree):
        return type(tree)(*[tree_map(f, t, *[tuple.__getitem__(r, i) for r in rest])
                            for i, t in enumerate(tuple.__iter__(tree))])
    if isinstance(tree, (tuple, list)):
        return type(tree)(tree_map(f, t, *[r[i] for r in rest]) for i, t in enumerate(tree))
    if isinstance(tree, dict):
        return {k: tree_map(f, v, *[r[k] for r in rest]) for k, v in tree.items()}
    return f(tree, *rest)


def _stack(outs):
    """Stack a list of identically structured pytrees leaf-wise along a new leading axis."""
    first = outs[0]
    if _is_namedtuple(first):
        cols = list(zip(*[list(tuple.__iter__(o)) for o in outs]))
        return type(first)(*[_stack(list(c)) for c in cols])
    if isinstance(first, (tuple, list)):
        cols = list(zip(*outs))
        return type(first)(_stack(list(c)) for c in cols)
    return np.stack([np.asarray(o) for o in outs])


def vmap(f, in_axes=0, out_axes=0):
    assert out_axes == 0

    def mapped(*args):
        axes = tuple(in_axes) if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        assert all(ax in (0, None) for ax in axes)
        n = next(len(a) for a, ax in zip(args, axes) if ax is not None)
        return _stack([f(*[a if ax is None else a[i] for a, ax in zip(args, axes)]) for i in range(n)])
    return mapped


def scan(f, init, xs=None, length=None, reverse=False):
    n = len(xs) if xs is not None else length
    carry, ys = init, [None] * n
    for i in (range(n - 1, -1, -1) if reverse else range(n)):
        carry, ys[i] = f(carry, None if xs is None else xs[i])
    return carry, _stack(ys)


def while_loop(cond_fun, body_fun, init_val):
    val = init_val
    while bool(cond_fun(val)):
        val = body_fun(val)
    return val


def jit(fun=None, **_kw):
    return fun


def _no_autodiff(name):
    def factory(fun, *a, **k):
        def call(*args, **kwargs):
            raise NotImplementedError(f"jax.{name} has no NumPy stand-in (autodiff); the fixtures do not use it")
        return call
    return factory


def _eigh(a, *args, **kwargs):
    a = np.asarray(a)
    return np.linalg.eigh(0.5 * (a + np.conj(np.swapaxes(a, -1, -2))))   # jnp.linalg.eigh: symmetrize_input=True


def install():
    """Registers the stand-in modules in sys.modules (refuses to shadow a real jax)."""
    if "jax" in sys.modules and not getattr(sys.modules["jax"], "_zopt_amd_standin", False):
        raise RuntimeError("a real jax is loaded; the stand-in is only for containers without it")
    jnp = types.ModuleType("jax.numpy")
    for k in dir(np):
        if not k.startswith("__"):
            setattr(jnp, k, getattr(np, k))
    la = types.ModuleType("jax.numpy.linalg")
    for k in dir(np.linalg):
        if not k.startswith("__"):
            setattr(la, k, getattr(np.linalg, k))
    la.eigh = _eigh
    jnp.linalg = la
    lax = types.ModuleType("jax.lax")
    lax.scan, lax.while_loop = scan, while_loop
    tree = types.ModuleType("jax.tree")
    tree.map = tree_map
    config = types.SimpleNamespace(update=lambda *a, **k: None)
    ode = types.ModuleType("jax.experimental.ode")
    ode.odeint = _no_autodiff("experimental.ode.odeint")(None)
    exp = types.ModuleType("jax.experimental")
    exp.ode = ode
    jax = types.ModuleType("jax")
    jax._zopt_amd_standin = True
    jax.numpy, jax.lax, jax.tree, jax.config, jax.experimental = jnp, lax, tree, config, exp
    jax.vmap, jax.jit = vmap, jit
    jax.grad, jax.jacobian, jax.hessian = _no_autodiff("grad"), _no_autodiff("jacobian"), _no_autodiff("hessian")
    jax.jacfwd, jax.jacrev = _no_autodiff("jacfwd"), _no_autodiff("jacrev")
    for name, mod in (("jax", jax), ("jax.numpy", jnp), ("jax.numpy.linalg", la), ("jax.lax", lax), ("jax.tree", tree),
                      ("jax.experimental", exp), ("jax.experimental.ode", ode)):
        sys.modules[name] = mod
    return jax
