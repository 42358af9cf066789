#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and instruction counters of the solvers' one-lane-per-point expansion kernels at
# the full batch (three DDP iterations of 8192 x 100 points).   usage (GPU box, repo root):  bash tools/pmc_expand.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_expand}
mkdir -p $OUT
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/tools/bench_ilqr.py --ddp --reps 1 --max-iter 3 --no-warmup > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
for k in expand_quad_points_kernel quad_hessian_points_kernel; do
  echo "== $k (per launch; 8192 x 100 points)"; python3 $R/tools/pmc_summary.py $OUT "$k"
done
