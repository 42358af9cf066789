set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_mpc_gpu.py -x -q > gpurun_out/r03_t4.log 2>&1; rc=$?
tail -15 gpurun_out/r03_t4.log
[ $rc -eq 0 ] || exit $rc
bash tools/pmc_sweeps.sh r03_pmc_sweeps > gpurun_out/r03_pmc_sweeps.txt 2>&1
tail -40 gpurun_out/r03_pmc_sweeps.txt
