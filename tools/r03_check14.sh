set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sweep_tiled_gpu.py tests/test_reference_fixtures_gpu.py tests/test_psd_tiled_gpu.py -x -q > gpurun_out/r03_t14.log 2>&1; rc=$?
tail -6 gpurun_out/r03_t14.log
[ $rc -eq 0 ] || exit $rc
python tools/bench_sweep_tiled.py --reps 8 2>&1 | grep '^{' | grep sweep_tiled > gpurun_out/r03_sweep_tiled_after.txt
cat gpurun_out/r03_sweep_tiled_after.txt
