set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ilqr_solve_gpu.py tests/test_ddp_gpu.py tests/test_lqr_gpu.py tests/test_edge_cases_gpu.py tests/test_ilqr_gpu.py tests/test_affine_lqr_gpu.py tests/test_reference_fixtures_gpu.py -x -q > gpurun_out/r03_t6.log 2>&1; rc=$?
tail -5 gpurun_out/r03_t6.log
[ $rc -eq 0 ] || exit $rc
for b in 1024 8192; do python tools/bench_ilqr.py --batch $b --reps 5; done > gpurun_out/r03_ilqr_after_nan.txt 2>&1
cat gpurun_out/r03_ilqr_after_nan.txt
