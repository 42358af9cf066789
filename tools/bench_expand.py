#!/usr/bin/env python3
"""Times the expansion kernels at the BASELINE configs[3] shape (8192 trajectories x T=100 points, quadcopter): Jacobians alone
(zm_linearize_dynamics_f64), cost gradients alone (zm_quadratize_cost_f64), packed second derivatives."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zopt_amd import _lib, models

b, T = 8192, 100
g = torch.Generator(device="cuda").manual_seed(0)
x = 0.3 * torch.randn(b, T + 1, 12, device="cuda", dtype=torch.float64, generator=g)
u = torch.tensor([9.807, 0, 0, 0], device="cuda", dtype=torch.float64) + 0.3 * torch.randn(b, T, 4, device="cuda", dtype=torch.float64, generator=g)
f_x = torch.empty(b, T, 12, 12, device="cuda", dtype=torch.float64)
f_u = torch.empty(b, T, 12, 4, device="cuda", dtype=torch.float64)
H = torch.empty(b, T, 28, 12, device="cuda", dtype=torch.float64)
c_x, c_u = torch.empty(b, T, 12, device="cuda", dtype=torch.float64), torch.empty(b, T, 4, device="cuda", dtype=torch.float64)
v_x = torch.empty(b, 12, device="cuda", dtype=torch.float64)
md = models.QuadcopterEuler(0.1).c_struct()
cost = models.QuadraticCost(np.eye(12), np.eye(4), 10 * np.eye(12))
cs = cost.c_struct()
lib = _lib.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


t1 = timeit(lambda: _lib.check(lib.zm_linearize_dynamics_f64(ctypes.addressof(md), x.data_ptr(), u.data_ptr(), None, None, f_x.data_ptr(), f_u.data_ptr(), b, T, st), "lin"))
t2 = timeit(lambda: _lib.check(lib.zm_quadratize_cost_f64(ctypes.addressof(cs), 12, 4, x.data_ptr(), u.data_ptr(), None, None, c_x.data_ptr(), c_u.data_ptr(), None, v_x.data_ptr(), None, None, None, None, b, T, st), "cost"))
t3 = timeit(lambda: _lib.check(lib.zm_quadratic_dynamics_pairs_list_f64(ctypes.addressof(md), x.data_ptr(), u.data_ptr(), None, 0, None, H.data_ptr(), b, T, st), "pairs"))
print(json.dumps({"points": b * T, "jacobians_us": t1, "jacobians_TBps": b * T * 1536 / t1 / 1e6, "cost_gradients_us": t2,
                  "packed_second_derivatives_us": t3, "second_TBps": b * T * 2688 / t3 / 1e6}))
