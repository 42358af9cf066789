import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import secondary_bench as sb
from zopt_amd import ilqrUtils
for ddp in (False, True):
    model, cost, x0, ug = sb.config3_problem(ddp=ddp)
    tx0, tug = torch.as_tensor(x0, device="cuda"), torch.as_tensor(ug, device="cuda")
    solve = ilqrUtils.differentialDynamicProgramming if ddp else ilqrUtils.iterativeLqr
    solve(model, cost, cost, tx0[:64], tug[:64])
    ilqrUtils._TRACE = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    solve(model, cost, cost, tx0, tug)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    tr = ilqrUtils._TRACE; ilqrUtils._TRACE = None
    print("DDP" if ddp else "iLQR", f"{t*1e3:.1f} ms", " ".join(f"{i}:{c}" for i, c in tr))
    print("  trajectory-iterations total:", sum(c for _, c in tr) * 4, " iterations:", tr[-1][0] + 4)
