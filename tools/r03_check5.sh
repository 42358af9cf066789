set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ilqr_solve_gpu.py tests/test_ddp_gpu.py tests/test_ilqr_tail_gpu.py tests/test_simulator_gpu.py tests/test_generic_gpu.py -x -q > gpurun_out/r03_t5.log 2>&1; rc=$?
tail -8 gpurun_out/r03_t5.log
[ $rc -eq 0 ] || exit $rc
for b in 1024 8192; do python tools/bench_ilqr.py --batch $b --reps 5; python tools/bench_ilqr.py --batch $b --reps 3 --ddp; done > gpurun_out/r03_ilqr_after_poll.txt 2>&1
cat gpurun_out/r03_ilqr_after_poll.txt
