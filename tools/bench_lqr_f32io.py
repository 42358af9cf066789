#!/usr/bin/env python3
"""BASELINE.md section 3, row "n=12, m=4, fp32" (1 600 B per horizon step): discreteFiniteHorizonLqr on fp32 arrays at the configs[1]
shape (4096 trajectories x T=50) -- K1 on fp32 storage with fp64 arithmetic (zm_lqr_backward_f32 -> lqr_backward_dma_f64<..., float>),
against the fp64 call on the same problem and against the fp32 tile kernel (ZOPT_AMD_LQR_F32=tile in another process)."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    args = ap.parse_args()
    import torch
    from zopt_amd import _lib
    b, T, n, m = args.batch, args.T, 12, 4
    g = torch.Generator(device="cuda").manual_seed(3)
    rn = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float64, generator=g)
    A = rn(b, T, n, n) * (0.9 / n ** 0.5)
    B = rn(b, T, n, m)
    Mq, Mr = rn(b, T, n, n), rn(b, T, m, m)
    Q = Mq @ Mq.transpose(-1, -2) / n + torch.eye(n, device="cuda", dtype=torch.float64)
    R = Mr @ Mr.transpose(-1, -2) / m + torch.eye(m, device="cuda", dtype=torch.float64)
    lib = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for name, dt, fn, es in (("fp64", torch.float64, lib.zm_lqr_backward_f64, 8), ("fp32_storage", torch.float32, lib.zm_lqr_backward_f32, 4)):
        a4 = [X.to(dt).contiguous() for X in (A, B, Q, R)]
        L = torch.empty((b, T, m, n), device="cuda", dtype=dt)
        call = lambda: _lib.check(fn(*[X.data_ptr() for X in a4], L.data_ptr(), b, T, n, m, st), name)
        for _ in range(args.warmup):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            call()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / args.steps
        bps = es * (2 * n * n + n * m + m * m + m * n)
        out[name] = {"us_per_launch": t * 1e6, "horizon_steps_per_s": b * T / t, "bytes_per_step": bps,
                     "algorithmic_GBps": b * T * bps / t / 1e9, "frac_of_8TBps": b * T * bps / t / 8e12}
        if name == "fp64":
            L64 = L.clone()
        else:
            out["fp32_vs_fp64_max_rel"] = float(((L.double() - L64).abs().max() / L64.abs().max()).item())
    print(json.dumps({"workload": f"discreteFiniteHorizonLqr n=12 m=4 T={T} batch={b}: fp64 arrays vs fp32 arrays (fp64 arithmetic)", **out}))


if __name__ == "__main__":
    main()
