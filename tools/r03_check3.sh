set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_rollout_wide_gpu.py tests/test_rollout_gpu.py -x -q > gpurun_out/r03_t3.log 2>&1; rc=$?
tail -30 gpurun_out/r03_t3.log
exit $rc
