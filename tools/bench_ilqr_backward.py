#!/usr/bin/env python3
"""Kernel benchmarks for the sweep kernel family at the BASELINE configs[3] shape (8192 trajectories, T=100, n=12, m=4, fp64):
K3 backwardPass_ilqr (ilqrUtils.py:153-181) with per-step cost Hessians (3 752 B per horizon step) and with the
trajectory-independent Hessians the fused iLQR driver uses; K4 backwardPass_ddp (:184-214, + 19 968 B per step of second-order
dynamics); K2 bilinearAffineLqr (lqrUtils.py:207-262) at (12,4) and at the reference demo's (8,4)."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from zopt_amd import _lib
    b, T, n, m = args.batch, args.T, 12, 4
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float64, generator=g)
    f_x = rn(b, T, n, n) * (0.9 / n ** 0.5)
    f_u = rn(b, T, n, m)
    M = rn(b, T, n + m, n + m)
    H = M @ M.transpose(-1, -2) / (n + m) + torch.eye(n + m, device="cuda", dtype=torch.float64)
    c_xx, c_ux, c_uu = H[..., :n, :n].contiguous(), H[..., n:, :n].contiguous(), H[..., n:, n:].contiguous()
    c_x, c_u = rn(b, T, n), rn(b, T, m)
    Mv = rn(b, n, n)
    v_xx = Mv @ Mv.transpose(-1, -2) / n + torch.eye(n, device="cuda", dtype=torch.float64)
    v_x = rn(b, n)
    l = torch.empty((b, T, m), device="cuda", dtype=torch.float64)
    L = torch.empty((b, T, m, n), device="cuda", dtype=torch.float64)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = _lib.lib()
    sh = [c_xx[0, 0].contiguous(), c_ux[0, 0].contiguous(), c_uu[0, 0].contiguous(), v_xx[0].contiguous()]

    def full():
        _lib.check(lib.zm_ilqr_backward_f64(f_x.data_ptr(), f_u.data_ptr(), c_x.data_ptr(), c_u.data_ptr(), c_xx.data_ptr(),
                                            c_ux.data_ptr(), c_uu.data_ptr(), v_x.data_ptr(), v_xx.data_ptr(), l.data_ptr(),
                                            L.data_ptr(), b, T, n, m, st), "full")

    def shared():
        _lib.check(lib.zm_ilqr_backward_ex_f64(f_x.data_ptr(), f_u.data_ptr(), c_x.data_ptr(), c_u.data_ptr(),
                                               sh[0].data_ptr(), sh[1].data_ptr(), sh[2].data_ptr(), v_x.data_ptr(),
                                               sh[3].data_ptr(), None, 1, l.data_ptr(), L.data_ptr(), b, T, n, m, st), "shared")

    for name, fn, bytes_step in (("per-step Hessians", full, 8 * (n * n + n * m + n + m + n * n + m * n + m * m + m + m * n)),
                                 ("shared Hessians", shared, 8 * (n * n + n * m + n + m + m + m * n))):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        t = sorted(ts)[len(ts) // 2]
        print(json.dumps({"kernel": "ilqr_backward_t16_f64", "variant": name, "batch": b, "T": T, "us": t * 1e6,
                          "horizon_steps_per_s": b * T / t, "bytes_per_step": bytes_step,
                          "algorithmic_GBps": b * T * bytes_step / t / 1e9, "hbm_frac": b * T * bytes_step / t / 8e12}))


    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        return sorted(ts)[len(ts) // 2]

    # K4: DDP backward pass
    f_xx, f_ux, f_uu = rn(b, T, n, n, n) * 0.05, rn(b, T, n, m, n) * 0.05, rn(b, T, n, m, m) * 0.05
    t = timeit(lambda: _lib.check(lib.zm_ddp_backward_f64(
        f_x.data_ptr(), f_u.data_ptr(), f_xx.data_ptr(), f_ux.data_ptr(), f_uu.data_ptr(), c_x.data_ptr(), c_u.data_ptr(),
        c_xx.data_ptr(), c_ux.data_ptr(), c_uu.data_ptr(), v_x.data_ptr(), v_xx.data_ptr(), None, 0, l.data_ptr(), L.data_ptr(),
        b, T, n, m, st), "ddp"))
    bs = 8 * (n * n + n * m + n + m + n * n + m * n + m * m + m + m * n + n * (n * n + m * n + m * m))
    print(json.dumps({"kernel": "ilqr_backward_t16_f64<MODE 2> (DDP)", "batch": b, "T": T, "us": t * 1e6,
                      "horizon_steps_per_s": b * T / t, "bytes_per_step": bs, "algorithmic_GBps": b * T * bs / t / 1e9,
                      "hbm_frac": b * T * bs / t / 8e12}))
    del f_xx, f_ux, f_uu
    # K4 as the fused DDP driver runs it: packed second derivatives of the quadcopter's 28 declared pairs, shared cost Hessians
    # (zm_ddp_backward_pairs_list_f64 -> the LDS-ring kernel ilqr_backward_dma_f64<MODE 2>)
    from zopt_amd import models
    md = models.QuadcopterEuler(0.1).c_struct()
    Hp = rn(b, T, 28, n) * 0.05
    t = timeit(lambda: _lib.check(lib.zm_ddp_backward_pairs_list_f64(
        ctypes.addressof(md), f_x.data_ptr(), f_u.data_ptr(), Hp.data_ptr(), c_x.data_ptr(), c_u.data_ptr(), sh[0].data_ptr(),
        sh[1].data_ptr(), sh[2].data_ptr(), v_x.data_ptr(), sh[3].data_ptr(), None, 0, None, 1, l.data_ptr(), L.data_ptr(), b, T, st),
        "ddp packed"))
    bs = 8 * (n * n + n * m + n + m + 28 * n + m + m * n)
    print(json.dumps({"kernel": "ilqr_backward_dma_f64<MODE 2> (DDP, packed pairs, shared Hessians)", "batch": b, "T": T, "us": t * 1e6,
                      "horizon_steps_per_s": b * T / t, "bytes_per_step": bs, "algorithmic_GBps": b * T * bs / t / 1e9,
                      "hbm_frac": b * T * bs / t / 8e12}))
    del Hp
    # K2: bilinearAffineLqr
    for (n2, m2) in ((12, 4), (8, 4)):
        A2 = rn(b, T, n2, n2) * (0.9 / n2 ** 0.5)
        B2, d2 = rn(b, T, n2, m2), rn(b, T, n2)
        M2 = rn(b, T, n2 + m2, n2 + m2)
        H2 = M2 @ M2.transpose(-1, -2) / (n2 + m2) + torch.eye(n2 + m2, device="cuda", dtype=torch.float64)
        Q2, Hx, R2 = H2[..., :n2, :n2].contiguous(), H2[..., n2:, :n2].contiguous(), H2[..., n2:, n2:].contiguous()
        q2, r2 = rn(b, T, n2), rn(b, T, m2)
        L2 = torch.empty((b, T, m2, n2), device="cuda", dtype=torch.float64)
        l2 = torch.empty((b, T, m2), device="cuda", dtype=torch.float64)
        t = timeit(lambda: _lib.check(lib.zm_lqr_backward_affine_f64(
            A2.data_ptr(), B2.data_ptr(), d2.data_ptr(), Q2.data_ptr(), R2.data_ptr(), Hx.data_ptr(), q2.data_ptr(),
            r2.data_ptr(), L2.data_ptr(), l2.data_ptr(), b, T, n2, m2, st), "affine"))
        bs = 8 * (2 * n2 * n2 + n2 * m2 + n2 + m2 * m2 + m2 * n2 + n2 + m2 + m2 * n2 + m2)
        print(json.dumps({"kernel": f"ilqr_backward_t16_f64<MODE 1> (bilinearAffineLqr n={n2} m={m2})", "batch": b, "T": T,
                          "us": t * 1e6, "horizon_steps_per_s": b * T / t, "bytes_per_step": bs,
                          "algorithmic_GBps": b * T * bs / t / 1e9, "hbm_frac": b * T * bs / t / 8e12}))
        del A2, B2, d2, M2, H2, Q2, Hx, R2, q2, r2, L2, l2


if __name__ == "__main__":
    main()
