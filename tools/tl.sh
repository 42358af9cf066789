#!/bin/bash
# usage (on the GPU box): tools/tl.sh <tag>   -- kernel timelines of one iLQR and one DDP solve into gpurun_out/tl_<tag>_{ilqr,ddp}.txt
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$1_ilqr -o tl -- python3 $R/tools/bench_ilqr.py --reps 1 > $R/gpurun_out/tl_$1_ilqr.log 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$1_ddp -o tl -- python3 $R/tools/bench_ilqr.py --ddp --reps 1 > $R/gpurun_out/tl_$1_ddp.log 2>&1
cd $R
python3 tools/timeline_solve.py gpurun_out/tl_$1_ilqr/tl_results.db > gpurun_out/tl_$1_ilqr.txt
python3 tools/timeline_solve.py gpurun_out/tl_$1_ddp/tl_results.db > gpurun_out/tl_$1_ddp.txt
