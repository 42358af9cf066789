#!/usr/bin/env python3
"""Rates of the large-state kernels added in round 3 (not the driver's bench.py line):
  * sweep_tiled_f64 -- backwardPass_ilqr / bilinearAffineLqr on fp64 MFMA tiles (n <= 48, m <= 16): horizon-steps/s, algorithmic GB/s
  * lqr_backward_tiled<TileF64, 4> -- discreteFiniteHorizonLqr fp64 at four tile rows (48 < n <= 64)
  * rollout_wide -- forwardPass2 of large linear models (16 step sizes + the winner's re-roll per trajectory)
usage: python tools/bench_sweep_tiled.py [--batch 2048] [--T 50] [--reps 8]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 4]       # lower quartile: steady state without the single best launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--reps", type=int, default=8)
    args = ap.parse_args()
    import torch
    from zopt_amd import _lib
    lib = _lib.lib()
    b, T = args.batch, args.T
    g = torch.Generator(device="cuda").manual_seed(5)
    rn = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float64, generator=g)   # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: t.data_ptr()   # noqa: E731
    for n, m in ((16, 4), (32, 8), (48, 16)):
        f_x, f_u = rn(b, T, n, n) * (0.9 / n ** 0.5), rn(b, T, n, m)
        M = rn(b, T, n + m, n + m)
        H = M @ M.transpose(-1, -2) / (n + m) + torch.eye(n + m, device="cuda", dtype=torch.float64)
        c_xx, c_ux, c_uu = H[..., :n, :n].contiguous(), H[..., n:, :n].contiguous(), H[..., n:, n:].contiguous()
        c_x, c_u, d = rn(b, T, n), rn(b, T, m), 0.3 * rn(b, T, n)
        v_x, v_xx = rn(b, n), c_xx[:, -1].contiguous()
        l, L = torch.empty(b, T, m, device="cuda", dtype=torch.float64), torch.empty(b, T, m, n, device="cuda", dtype=torch.float64)
        del M, H
        ilqr = lambda: _lib.check(lib.zm_ilqr_backward_f64(p(f_x), p(f_u), p(c_x), p(c_u), p(c_xx), p(c_ux), p(c_uu), p(v_x), p(v_xx),   # noqa: E731
                                                           p(l), p(L), b, T, n, m, st), "ilqr")
        aff = lambda: _lib.check(lib.zm_lqr_backward_affine_f64(p(f_x), p(f_u), p(d), p(c_xx), p(c_uu), p(c_ux), p(c_x), p(c_u), p(L),   # noqa: E731
                                                                p(l), b, T, n, m, st), "affine")
        for name, fn, extra in (("backwardPass_ilqr", ilqr, 0), ("bilinearAffineLqr", aff, n)):
            t = timed(fn, args.reps)
            bps = 8 * (2 * n * n + 2 * n * m + m * m + n + m + extra + m * n + m)   # operands read once + (l, L) written
            print(json.dumps({"kernel": "sweep_tiled_f64", "op": name, "n": n, "m": m, "batch": b, "T": T, "ms": t * 1e3,
                              "horizon_steps_per_s": b * T / t, "algorithmic_GBps": b * T * bps / t / 1e9,
                              "finite": bool(torch.isfinite(L).all().item())}))
        del f_x, f_u, c_xx, c_ux, c_uu, c_x, c_u, d, v_x, v_xx, l, L
        torch.cuda.empty_cache()
    # forwardPass2 of a large linear model
    import numpy as np
    from zopt_amd import models
    for n, m in ((24, 8), (64, 16)):
        rng = np.random.default_rng(n)
        model = models.LinearModel(rng.standard_normal((n, n)) * (0.9 / np.sqrt(n)), rng.standard_normal((n, m)))
        cost = models.QuadraticCost(np.eye(n), np.eye(m), 10 * np.eye(n))
        md, cs = model.c_struct(), cost.c_struct()
        x0, l, L = rn(b, n), 0.3 * rn(b, T, m), 0.2 * rn(b, T, m, n) / n ** 0.5
        xp, up = rn(b, T + 1, n), 0.3 * rn(b, T, m)
        al = torch.as_tensor(0.5 ** np.arange(16), device="cuda")
        xT, uT = torch.empty(b, T + 1, n, device="cuda", dtype=torch.float64), torch.empty(b, T, m, device="cuda", dtype=torch.float64)
        J, idx = torch.empty(b, device="cuda", dtype=torch.float64), torch.empty(b, device="cuda", dtype=torch.int32)
        fp2 = lambda: _lib.check(lib.zm_rollout_linesearch_f64(ctypes.addressof(md), ctypes.addressof(cs), p(x0), p(l), p(L), p(xp), p(up), p(al),   # noqa: E731
                                                               16, None, p(xT), p(uT), p(J), p(idx), b, T, st), "rollout")
        t = timed(fp2, args.reps)
        print(json.dumps({"kernel": "rollout_wide", "op": "forwardPass2", "n": n, "m": m, "batch": b, "T": T, "ms": t * 1e3,
                          "rollout_steps_per_s": b * 17 * T / t, "finite": bool(torch.isfinite(J).all().item())}))


if __name__ == "__main__":
    main()
