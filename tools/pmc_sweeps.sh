#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE) and instruction counters of the sweep kernels (K2 / K3 / K4) in separate rocprofv3 --pmc passes
# over tools/bench_ilqr_backward.py.   usage (GPU box, repo root):  bash tools/pmc_sweeps.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_sweeps}
mkdir -p $OUT
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/tools/bench_ilqr_backward.py --reps 2 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
for k in "ilqr_backward_dma_f64<12, 4, 2, false, 0," "ilqr_backward_dma_f64<12, 4, 3, true, 0," "ilqr_backward_dma_f64<12, 4, 2, false, 1," "ilqr_backward_dma_f64<12, 4, 2, true, 2," "ilqr_backward_t16_f64<3, 2>"; do
  echo "== $k"; python3 $R/tools/pmc_summary.py $OUT "$k"
done
