set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_psd_tiled_gpu.py tests/test_ilqr_solve_gpu.py -x -q -k "not full_size and not non_converging" > gpurun_out/r03_t7.log 2>&1; rc=$?
tail -40 gpurun_out/r03_t7.log
exit $rc
