set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sweep_tiled_gpu.py tests/test_lqr_tiled_gpu.py tests/test_dare_gpu.py tests/test_affine_lqr_gpu.py tests/test_ilqr_gpu.py -x -q > gpurun_out/r03_t2.log 2>&1; rc=$?
tail -30 gpurun_out/r03_t2.log
[ $rc -eq 0 ] || exit $rc
python - > gpurun_out/r03_n64_f64.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, '.')
from tools import secondary_bench as sb
for n, m in ((64, 16), (48, 16), (56, 8)):
    r = sb.config4_tiled(batch=2048, T=50, n=n, m=m, reps=6, fp64=True)
    print(json.dumps({k: r[k] for k in ("workload", "ms", "horizon_steps_per_s", "mfma_TFLOPs", "finite")}))
PY
cat gpurun_out/r03_n64_f64.txt
