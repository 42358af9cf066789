#!/bin/bash
# PMC counters of the MPC solve kernel (BASELINE configs[2]) in separate passes.  usage (GPU box, repo root): bash tools/pmc_mpc.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_mpc}
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/tools/bench_mpc.py --eps 1e-2 --rh-steps 0 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $OUT mpc_solve > $OUT/summary.txt
cat $OUT/summary.txt
