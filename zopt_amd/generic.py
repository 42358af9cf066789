"""Generic-callable path of the iLQR / DDP drivers and the rollout functions (SURVEY 7 "hard part (iii)", 8(b)).

The reference takes arbitrary JAX callables `dynamics(x, u) -> x+`, `runningCost(x, u) -> c`, `terminalCost(x) -> cf` and
differentiates them by JAX autodiff (pytrees.py:72-81, 100-115, 139-153, 180-194).  A HIP kernel cannot call Python, so the fused
driver needs a *registered device model* (zopt_amd.models).  For everything else this module plays the part JAX plays in the
reference -- and only that part: the callables are **torch** functions of single points (a torch restatement of the user's jnp
function), evaluated and differentiated ON THE GPU with `torch.func` (vmap / jacrev / hessian) to PRODUCE the arrays
(f_x, f_u, f_xx, ..., c_x, ..., the 16 line-search rollouts); the sweeps themselves -- backwardPass_ilqr / backwardPass_ddp, the PD
projections -- are the HIP kernels, exactly as in the registered-model path.  Nothing here solves anything on the CPU, and nothing
runs without the HIP library.

This path is launch-bound Python (hundreds of small torch kernels per iteration); it exists for coverage of the reference's
interface, not for speed -- register a device model for that.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _arrays as arr

LINESEARCH_ALPHAS = 0.5 ** np.arange(16)


def is_callable_model(f):
    return callable(f) and not hasattr(f, "c_struct")


def _vv(f, in_dims=0):
    """f over two leading axes (batch, time)"""
    return torch.func.vmap(torch.func.vmap(f, in_dims=in_dims), in_dims=in_dims)


def _dev(a):
    arr.require_gpu()
    return arr.to_device(a, torch.float64)


# ---------------------------------------------------------------------------------------------------------------------------------
# expansions (the role of jax.jacobian / jax.hessian in pytrees.py)
# ---------------------------------------------------------------------------------------------------------------------------------
def expand_dynamics(dynFun, xT, uT, second_order):
    """xT (B, N+1, n), uT (B, N, m) device tensors -> [f, f_x, f_u] (+ [f_xx, f_ux, f_uu]) along the trajectory
    (AffineDynamics / QuadraticDynamics.from_trajectory, pytrees.py:139-153, 180-194)."""
    x, u = xT[:, :-1], uT
    f = _vv(dynFun)(x, u)
    f_x, f_u = _vv(torch.func.jacrev(dynFun, argnums=(0, 1)))(x, u)
    outs = [f, f_x.contiguous(), f_u.contiguous()]
    if second_order:
        # jax.hessian(dynFun, (0, 1)): ((f_xx, f_xu), (f_ux, f_uu)); the reference keeps f_xx (n,n,n), f_ux (n,m,n), f_uu (n,m,m)
        H = _vv(torch.func.jacfwd(torch.func.jacrev(dynFun, argnums=(0, 1)), argnums=(0, 1)))(x, u)
        (f_xx, _f_xu), (f_ux, f_uu) = H
        outs += [f_xx.contiguous(), f_ux.contiguous(), f_uu.contiguous()]
    return outs


def expand_cost(runningCost, xT, uT):
    """[c, c_x, c_u, c_xx, c_ux, c_uu] along the trajectory (QuadraticCostFunction.from_trajectory, pytrees.py:100-115)"""
    x, u = xT[:, :-1], uT
    c = _vv(runningCost)(x, u)
    c_x, c_u = _vv(torch.func.jacrev(runningCost, argnums=(0, 1)))(x, u)
    (c_xx, _c_xu), (c_ux, c_uu) = _vv(torch.func.jacfwd(torch.func.jacrev(runningCost, argnums=(0, 1)), argnums=(0, 1)))(x, u)
    return [c, c_x.contiguous(), c_u.contiguous(), c_xx.contiguous(), c_ux.contiguous(), c_uu.contiguous()]


def expand_terminal(terminalCost, xf):
    """[v, v_x, v_xx] at xf (B, n) (QuadraticValueFunction.fromTerminalCostFunction, pytrees.py:72-81)"""
    vm = torch.func.vmap
    v = vm(terminalCost)(xf)
    v_x = vm(torch.func.jacrev(terminalCost))(xf)
    v_xx = vm(torch.func.hessian(terminalCost))(xf)
    return [v, v_x.contiguous(), v_xx.contiguous()]


# ---------------------------------------------------------------------------------------------------------------------------------
# rollouts (trajectoryRollout / forwardPass2, ilqrUtils.py:33-66, 116-150)
# ---------------------------------------------------------------------------------------------------------------------------------
def rollout(x0, dynFun, l, L, xPrev, uPrev, alphas, runningCost=None, terminalCost=None):
    """x0 (B, n), l (B, N, m), L (B, N, m, n), xPrev (B, N+1, n), uPrev (B, N, m), alphas (A,): every trajectory rolled out with
    every step size, `u_k = alpha l_k + L_k (x_k - xPrev_k) + uPrev_k`, `x_{k+1} = dynFun(x_k, u_k)` (pytrees.py:215-220,
    ilqrUtils.py:59-61).  Returns (xTraj (B, A, N+1, n), uTraj (B, A, N, m), J (B, A) or None)."""
    B, N = l.shape[0], l.shape[1]
    A = alphas.shape[0]
    step = _vv(dynFun)
    x = x0[:, None, :].expand(B, A, x0.shape[-1]).contiguous()
    xs, us = [x], []
    J = torch.zeros((B, A), dtype=x0.dtype, device=x0.device) if runningCost is not None else None
    rc = _vv(runningCost) if runningCost is not None else None
    for k in range(N):
        dx = x - xPrev[:, None, k, :]
        u = (alphas[None, :, None] * l[:, None, k, :] + torch.einsum("bij,baj->bai", L[:, k], dx)) + uPrev[:, None, k, :]
        if rc is not None:
            J = J + rc(x, u)
        x = step(x, u)
        xs.append(x)
        us.append(u)
    if terminalCost is not None:
        J = J + _vv(terminalCost)(x)
    return torch.stack(xs, dim=2), torch.stack(us, dim=2), J


def argmin_nan_wins(J):
    """jnp.argmin over the last axis: a NaN beats every number (also -inf), the first index wins among equals (ilqrUtils.py:147)"""
    isn = torch.isnan(J)
    first_nan = torch.argmax(isn.to(torch.int8), dim=-1)
    return torch.where(isn.any(dim=-1), first_nan, torch.argmin(torch.where(isn, torch.full_like(J, float("inf")), J), dim=-1))


def forward_pass2(x0, dynFun, runningCost, terminalCost, l, L, xPrev, uPrev):
    """-> (xTraj (B, N+1, n), uTraj (B, N, m), J (B,)): the rollout of minimum cost among the 16 step sizes 0.5**j"""
    al = torch.as_tensor(LINESEARCH_ALPHAS, dtype=x0.dtype, device=x0.device)
    xs, us, J = rollout(x0, dynFun, l, L, xPrev, uPrev, al, runningCost, terminalCost)
    idx = argmin_nan_wins(J)
    b = torch.arange(x0.shape[0], device=x0.device)
    return xs[b, idx], us[b, idx], J[b, idx]


# ---------------------------------------------------------------------------------------------------------------------------------
# the drivers (ilqrUtils.py:260-397)
# ---------------------------------------------------------------------------------------------------------------------------------
def solve(dynamics, runningCost, terminalCost, x0, uGuess, maxIter, tol, ddp):
    """iterativeLqr / differentialDynamicProgramming for torch callables on a batch: x0 (B, n), uGuess (B, N, m) device tensors.
    Per trajectory exactly the reference's loop (:290-327 / :360-397); converged trajectories keep their result while the others
    go on (under jax.vmap every lane would run to the slowest).  Returns (xTraj, uTraj, L, J, converged)."""
    from . import ilqrUtils as iu
    from . import pytrees as pt
    B, N, m = uGuess.shape
    n = x0.shape[-1]
    dev, dt = x0.device, x0.dtype
    L = torch.zeros((B, N, m, n), dtype=dt, device=dev)
    l = uGuess.clone()
    zx, zu = torch.zeros((B, N + 1, n), dtype=dt, device=dev), torch.zeros((B, N, m), dtype=dt, device=dev)
    one = torch.ones(1, dtype=dt, device=dev)
    xs, us, J = rollout(x0, dynamics, l, L, zx, zu, one, runningCost, terminalCost)        # policy = (uGuess, 0), alpha = 1  (:293-298)
    xT, uT, J = xs[:, 0].contiguous(), us[:, 0].contiguous(), J[:, 0].contiguous()
    converged = torch.zeros(B, dtype=torch.bool, device=dev)
    for _ in range(int(maxIter)):
        act = torch.nonzero(~converged).flatten()
        if act.numel() == 0:
            break
        xa, ua = xT[act].contiguous(), uT[act].contiguous()
        dyn = expand_dynamics(dynamics, xa, ua, ddp)
        cost = expand_cost(runningCost, xa, ua)
        Vf = expand_terminal(terminalCost, xa[:, -1])
        qc = iu.conditionQuadraticCost(pt.QuadraticCostFunction(*cost))                      # HIP: PD projection   (:312, :222-234)
        qv = iu.conditionValueFunction(pt.QuadraticValueFunction(*Vf))                      # HIP                  (:313, :254-257)
        if ddp:
            pol = iu.backwardPass_ddp(pt.QuadraticDynamics(*dyn), qc, qv)                   # HIP sweep K4         (:373)
        else:
            pol = iu.backwardPass_ilqr(pt.AffineDynamics(*dyn), qc, qv)                     # HIP sweep K3         (:315)
        xn, un, Jn = forward_pass2(x0[act], dynamics, runningCost, terminalCost, pol.l, pol.L, xa, ua)   # (:316)
        cv = (J[act] - Jn).abs() <= tol                                                     # NaN compares false   (:318)
        xT[act], uT[act], J[act], L[act] = xn, un, Jn, pol.L
        converged[act] = cv
    return xT, uT, L, J, converged
