import sys, ctypes, json
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from zopt_amd import _lib
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
for count in (8192, 8192 * 25):
    M = torch.randn(count, 16, 16, device="cuda", dtype=torch.float64, generator=g)
    A = M + M.transpose(-1, -2)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    B = A.clone()
    lib.zm_psd_project_f64(B.data_ptr(), count, 16, 1e-3, st); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        B.copy_(A)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.zm_psd_project_f64(B.data_ptr(), count, 16, 1e-3, st); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[2]
    print(json.dumps({"count": count, "ms": t, "us_per_matrix_per_wave_slot": t * 1e3 / (count / 2048.0)}))
