set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r03_tl1024_ilqr -o tl -- python3 $R/tools/bench_ilqr.py --reps 1 --batch 1024 > $R/gpurun_out/r03_tl1024_ilqr.log 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/r03_tl1024_ddp -o tl -- python3 $R/tools/bench_ilqr.py --ddp --reps 1 --batch 1024 > $R/gpurun_out/r03_tl1024_ddp.log 2>&1
cd $R
python3 tools/timeline_solve.py gpurun_out/r03_tl1024_ilqr/tl_results.db > gpurun_out/r03_tl1024_ilqr.txt
python3 tools/timeline_solve.py gpurun_out/r03_tl1024_ddp/tl_results.db > gpurun_out/r03_tl1024_ddp.txt
head -30 gpurun_out/r03_tl1024_ilqr.txt; head -16 gpurun_out/r03_tl1024_ddp.txt
