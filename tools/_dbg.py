import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from zopt_amd import _lib, models, ilqrUtils, pytrees
import tests.test_rollout_gpu as tr
for N in (2, 3, 9):
    rng = np.random.default_rng(100 + N)
    batch = 37
    x0, l, L, xPrev, uPrev = tr._quad_problem(rng, batch, N)
    l *= 60.0
    cost = models.QuadraticCost(np.diag(rng.uniform(0.5, 2.0, 12)), np.diag(rng.uniform(0.5, 2.0, 4)), np.diag(rng.uniform(5.0, 20.0, 12)))
    md, cs = models.QuadcopterEuler(0.1).c_struct(), cost.c_struct()
    dev = [torch.as_tensor(np.ascontiguousarray(X), device="cuda") for X in (x0, l, L, xPrev, uPrev)]
    al = torch.as_tensor(0.5 ** np.arange(16), device="cuda")
    lst = torch.arange(batch, dtype=torch.int32, device="cuda")
    act = torch.ones(batch, dtype=torch.int32, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for with_idx in (False, True):
        xT = torch.full((batch, N + 1, 12), -7.0, dtype=torch.float64, device="cuda")
        uT = torch.full((batch, N, 4), -7.0, dtype=torch.float64, device="cuda")
        J = torch.full((batch,), -7.0, dtype=torch.float64, device="cuda")
        idx = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib().zm_rollout_linesearch_list_f64(
            ctypes.addressof(md), ctypes.addressof(cs), *[t.data_ptr() for t in dev], al.data_ptr(), 16, lst.data_ptr(), lst.numel(),
            act.data_ptr(), xT.data_ptr(), uT.data_ptr(), J.data_ptr(), idx.data_ptr() if with_idx else None, batch, N, st), "rollout")
        torch.cuda.synchronize()
        out.append((xT.cpu().numpy(), uT.cpu().numpy(), J.cpu().numpy(), idx.cpu().numpy()))
    (x1, u1, J1, _), (x2, u2, J2, i2) = out
    print("N", N, "x equal", np.array_equal(x1, x2), "u equal", np.array_equal(u1, u2), "J equal", np.array_equal(J1, J2))
    du = np.abs(u1 - u2); dx = np.abs(x1 - x2)
    print("  max du", du.max(), "at", np.unravel_index(du.argmax(), du.shape), "max dx", dx.max(), "max dJ", np.abs(J1 - J2).max(), "idx", i2[:12])
    bad = np.argwhere(du > 0)
    print("  differing u entries:", bad[:10].tolist(), "count", len(bad))
    # oracle check of u for trajectory b at winner alpha
    b = int(bad[0][0]) if len(bad) else 0
    a_ = 0.5 ** i2[b]
    x = x0[b].copy(); 
    for k in range(N):
        u = a_ * l[b, k] + L[b, k] @ (x - xPrev[b, k]) + uPrev[b, k]
        print("   k", k, "u ref", u, "\n       u1", u1[b, k], "\n       u2", u2[b, k])
        x = x2[b, k + 1]
