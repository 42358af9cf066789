#!/usr/bin/env python3
"""Secondary benchmark (not the driver's bench.py line): BASELINE configs[4] -- discreteFiniteHorizonLqr at n=64, m=16,
T=200, fp32 on ONE GPU's share of the 16384-trajectory batch (2048 by default; inputs generated on the device:
A = 0.9 G/sqrt(n), B ~ N(0,1), Q = MM^T/n + I, R = NN^T/m + I, tiled over the horizon as the reference API takes them).
Reports horizon-steps/s, algorithmic GB/s (41 984 B/step) and fp32 MFMA TFLOP/s (864 MFMAs x 2048 flop per step)."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--T", type=int, default=200)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--fp64", action="store_true", help="time zm_lqr_backward_f64 (fp64 MFMA tile kernel up to n=48, LDS coverage kernel beyond) at the same shape")
    args = ap.parse_args()
    import torch
    from zopt_amd import _lib
    b, T, n, m = args.batch, args.T, args.n, args.m
    g = torch.Generator(device="cuda").manual_seed(3)
    dt = torch.float64 if args.fp64 else torch.float32
    rn = lambda *s: torch.randn(*s, device="cuda", dtype=dt, generator=g)
    A1 = rn(b, n, n) * (0.9 / n ** 0.5)
    B1 = rn(b, n, m)
    M, N = rn(b, n, n), rn(b, m, m)
    Q1 = M @ M.transpose(-1, -2) / n + torch.eye(n, device="cuda", dtype=dt)
    R1 = N @ N.transpose(-1, -2) / m + torch.eye(m, device="cuda", dtype=dt)
    A, B, Q, R = (X[:, None].expand(b, T, *X.shape[1:]).contiguous() for X in (A1, B1, Q1, R1))
    del M, N
    L = torch.empty((b, T, m, n), device="cuda", dtype=dt)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = _lib.lib()
    fn = lib.zm_lqr_backward_f64 if args.fp64 else lib.zm_lqr_backward_f32
    call = lambda: _lib.check(fn(A.data_ptr(), B.data_ptr(), Q.data_ptr(), R.data_ptr(), L.data_ptr(),
                                                      b, T, n, m, st), "bench")
    call()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    steps = b * T
    nt = (n + 15) // 16
    mfma = 4 * nt * nt + 4 * nt * (nt + 1) + 4 * nt ** 3 + 4 * nt * nt + 4 * nt + 4 * nt * nt + 4 * nt * nt + 4 * nt ** 3
    bytes_step = (8 if args.fp64 else 4) * (2 * n * n + 2 * n * m + m * m)
    print(json.dumps({"workload": f"discreteFiniteHorizonLqr n={n} m={m} T={T} batch={b} {'fp64 (MFMA tiles up to n=48, LDS coverage kernel beyond)' if args.fp64 else 'fp32'}", "ms": best * 1e3,
                      "horizon_steps_per_s": steps / best, "algorithmic_GBps": steps * bytes_step / best / 1e9,
                      "mfma_TFLOPs": steps * mfma * 2048 / best / 1e12, "finite": bool(torch.isfinite(L).all().item())}))


if __name__ == "__main__":
    main()
