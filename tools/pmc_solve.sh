#!/bin/bash
# PMC counters (separate passes) over one iLQR (or, with --ddp, DDP) solve of BASELINE configs[3]; per-kernel sums via tools/pmc_summary.py.
# usage (GPU box, repo root): bash tools/pmc_solve.sh <outdir-under-gpurun_out> [--ddp]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/tools/bench_ilqr.py --reps 1 --batch ${BATCH:-8192} "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
for k in ilqr_backward_dma rollout_ls_fast rollout_quad_all rollout_quad_reroll expand_quad ilqr_accept; do
  echo "== $k"; python3 $R/tools/pmc_summary.py $OUT $k
done > $OUT/summary.txt
cat $OUT/summary.txt
