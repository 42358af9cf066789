"""ctypes binding of the C ABI in ``include/zopt_amd.h`` (``zopt_amd/csrc/libzopt_amd.so``).

The library is loaded lazily; a missing library is a hard error (``ZoptAmdError``) -- the product
never falls back to a CPU path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# ZOPT_AMD_LIB: an alternative build of the same library (A/B measurements of kernel variants), else the in-tree one
LIB_PATH = os.environ.get("ZOPT_AMD_LIB") or os.path.join(CSRC, "libzopt_amd.so")

ZM_OK = 0
ZM_EINVAL = -1
ZM_EUNSUPPORTED = -2

# every symbol include/zopt_amd.h declares: name -> (restype, argtypes)
_c_dp = ctypes.c_void_p
SYMBOLS = {
    "zm_version": (ctypes.c_int, []),
    "zm_last_error": (ctypes.c_char_p, []),
    "zm_lqr_backward_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "zm_lqr_backward_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "zm_lqr_backward_f32": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "zm_dare_f64": (ctypes.c_int, [_c_dp] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                   ctypes.c_void_p]),
    "zm_care_f64": (ctypes.c_int, [_c_dp] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                   ctypes.c_void_p]),
    "zm_riccati_ode_f64": (ctypes.c_int, [_c_dp] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]),
    "zm_lqr_backward_host_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int]),
    "zm_lqr_backward_affine_f64": (ctypes.c_int, [_c_dp] * 10 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                 ctypes.c_void_p]),
    "zm_ilqr_backward_f64": (ctypes.c_int, [_c_dp] * 11 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                           ctypes.c_void_p]),
    # (model*, cost*, x0, l, L, xPrev, uPrev, alphas, n_alpha, active, xTraj, uTraj, J, alpha_idx, batch, T, stream)
    "zm_rollout_linesearch_f64": (ctypes.c_int, [_c_dp] * 8 + [ctypes.c_int] + [_c_dp] * 5 +
                                  [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "zm_rollout_linesearch_list_f64": (ctypes.c_int, [_c_dp] * 8 + [ctypes.c_int] + [_c_dp] + [ctypes.c_int64] + [_c_dp] * 5 +
                                       [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "zm_ilqr_accept_f64": (ctypes.c_int, [_c_dp, ctypes.c_int64] + [_c_dp] * 8 + [ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    # (model*, xTraj, uTraj, active, f, f_x, f_u, batch, T, stream)
    "zm_linearize_dynamics_f64": (ctypes.c_int, [_c_dp] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    # (cost*, n, m, xTraj, uTraj, active, c, c_x, c_u, v, v_x, c_xx, c_ux, c_uu, v_xx, batch, T, stream)
    "zm_quadratize_cost_f64": (ctypes.c_int, [_c_dp, ctypes.c_int, ctypes.c_int] + [_c_dp] * 12 +
                               [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    # (f_x, f_u, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, active, shared_hessian, l, L, batch, T, n, m, stream)
    "zm_ilqr_backward_ex_f64": (ctypes.c_int, [_c_dp] * 10 + [ctypes.c_int] + [_c_dp] * 2 +
                                [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    # (A, B, Q, R, Qf, rho, N, n, m, K, Minv, stream)
    "zm_condition_dynamics_f64": (ctypes.c_int, [_c_dp] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                              ctypes.c_void_p]),
    "zm_quadcopter_trim_f64": (ctypes.c_int, [_c_dp] * 6 + [ctypes.c_int64, ctypes.c_double, ctypes.c_void_p]),
    "zm_mpc_setup_f64": (ctypes.c_int, [_c_dp] * 5 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp,
                                                    ctypes.c_void_p]),
    # (A, B, K, Minv, x_lb, x_ub, u_lb, u_ub, x0, rho, eps_abs, eps_rel, max_iter, ws, xTraj, uTraj, status, iters, resid,
    #  batch, N, n, m, stream)
    "zm_mpc_solve_f64": (ctypes.c_int, [_c_dp] * 9 + [ctypes.c_double] * 4 + [ctypes.c_int] + [_c_dp] * 6 +
                         [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "zm_mpc_solve_adaptive_f64": (ctypes.c_int, [_c_dp] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double] + [_c_dp] * 5 +
                                  [ctypes.c_double] * 4 + [ctypes.c_int] * 2 + [_c_dp] * 6 +
                                  [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "zm_mpc_solve_relaxed_f64": (ctypes.c_int, [_c_dp] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double] + [_c_dp] * 5 +
                                 [ctypes.c_double] * 4 + [ctypes.c_int] * 2 + [_c_dp] * 6 +
                                 [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "zm_mpc_solve_warm_f64": (ctypes.c_int, [_c_dp] * 9 + [ctypes.c_double] * 4 + [ctypes.c_int] * 2 + [_c_dp] * 6 +
                              [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    # (f_x, f_u, f_xx, f_ux, f_uu, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, active, shared_hessian, l, L, batch, T, n, m, stream)
    "zm_riccati_value_f64": (ctypes.c_int, [_c_dp] * 19 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                           ctypes.c_void_p]),
    "zm_ddp_backward_f64": (ctypes.c_int, [_c_dp] * 13 + [ctypes.c_int] + [_c_dp] * 2 +
                            [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    # (model*, xTraj, uTraj, active, f_xx, f_ux, f_uu, batch, T, stream)
    "zm_quadratic_dynamics_f64": (ctypes.c_int, [_c_dp] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "zm_ilqr_backward_list_f64": (ctypes.c_int, [_c_dp] * 10 + [ctypes.c_int64, _c_dp, ctypes.c_int] + [_c_dp] * 2 +
                                  [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "zm_ddp_backward_list_f64": (ctypes.c_int, [_c_dp] * 13 + [ctypes.c_int64, _c_dp, ctypes.c_int] + [_c_dp] * 2 +
                                 [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    # (model*, xTraj, uTraj, list, count, active, f, f_x, f_u, batch, T, stream)
    "zm_linearize_dynamics_list_f64": (ctypes.c_int, [_c_dp] * 4 + [ctypes.c_int64] + [_c_dp] * 4 +
                                       [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "zm_quadratize_cost_list_f64": (ctypes.c_int, [_c_dp, ctypes.c_int, ctypes.c_int] + [_c_dp] * 3 + [ctypes.c_int64] +
                                    [_c_dp] * 10 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "zm_quadratic_dynamics_list_f64": (ctypes.c_int, [_c_dp] * 4 + [ctypes.c_int64] + [_c_dp] * 4 +
                                       [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "zm_model_hessian_pairs": (ctypes.c_int, [_c_dp, _c_dp, _c_dp]),
    # (model*, xTraj, uTraj, list, count, active, H, batch, T, stream)
    "zm_quadratic_dynamics_pairs_list_f64": (ctypes.c_int, [_c_dp] * 4 + [ctypes.c_int64] + [_c_dp] * 2 +
                                             [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    # (model*, f_x, f_u, H, c_x, c_u, c_xx, c_ux, c_uu, vf_x, vf_xx, list, count, active, shared_hessian, l, L, batch, T, stream)
    "zm_ddp_backward_pairs_list_f64": (ctypes.c_int, [_c_dp] * 12 + [ctypes.c_int64, _c_dp, ctypes.c_int] + [_c_dp] * 2 +
                                       [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "zm_ilqr_solve_workspace_f64": (ctypes.c_int64, [_c_dp, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    # (model*, cost*, x0, uGuess, ddp, max_iter, tol, sync_every, workspace, workspace_doubles, iwork, xTraj, uTraj, L, J, converged,
    #  iterations (host), batch, T, stream)
    "zm_ilqr_solve_f64": (ctypes.c_int, [_c_dp] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, _c_dp, ctypes.c_int64] +
                          [_c_dp] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "zm_shutdown": (ctypes.c_int, []),
    "zm_ilqr_solve_trace_f64": (ctypes.c_int, [_c_dp] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, _c_dp, ctypes.c_int64] +
                                [_c_dp] * 7 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, _c_dp, _c_dp]),
    "zm_model_nonlinear_mask": (ctypes.c_int, [_c_dp, _c_dp]),
    "zm_psd_project_f64": (ctypes.c_int, [_c_dp, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]),
    "zm_condition_cost_f64": (ctypes.c_int, [_c_dp] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                          ctypes.c_void_p]),
}


class ZoptAmdError(RuntimeError):
    """The HIP library is missing / failed; there is deliberately no fallback."""


_lib = None


LAB_LIB_PATH = os.path.join(CSRC, "libzopt_amd_lab.so")   # -DZM_LAB build: the A/B switches of the kernel lab compiled in (tests, tools)


def build(verbose: bool = False, lab: bool = True) -> str:
    """Compile the HIP sources for gfx950 in-tree (``make -C zopt_amd/csrc``): the product library and, with `lab`, the
    -DZM_LAB build of the same sources that the A/B tests load through ZOPT_AMD_LIB."""
    for target in ([], ["lab"]) if lab else ([],):
        out = subprocess.run(["make", "-C", CSRC, f"-j{max(1, min(8, os.cpu_count() or 1))}"] + target, capture_output=True, text=True)
        if verbose or out.returncode != 0:
            print(out.stdout)
            print(out.stderr)
        if out.returncode != 0:
            raise ZoptAmdError("building libzopt_amd.so failed (hipcc --offload-arch=gfx950)")
    return LIB_PATH


def lib():
    """The loaded shared library with argtypes set; raises ZoptAmdError if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZoptAmdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(zopt_amd has no CPU fallback)")
        # torch first: its wheel bundles its own HIP runtime, and the library must bind to THAT copy (same SONAME) -- loaded the
        # other way round, two runtimes end up in the process and the first launch fails with "no ROCm-capable device"
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise ZoptAmdError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        import atexit
        atexit.register(handle.zm_shutdown)      # pinned words / events of the drivers, while the HIP runtime is still up
    return _lib


def check(rc: int, what: str):
    """Map a C-ABI return code to the reference's error conventions (ValueError for bad shapes)."""
    if rc == ZM_OK:
        return
    msg = lib().zm_last_error().decode("utf-8", "replace")
    if rc in (ZM_EINVAL, ZM_EUNSUPPORTED):
        raise ValueError(f"{what}: {msg}")
    raise ZoptAmdError(f"{what}: HIP error {rc}: {msg}")
