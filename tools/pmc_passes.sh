#!/bin/bash
# Collects PMC counters for bench.py's dominant kernel in separate passes (gpurun: --pmc only with --kernel-trace).
# usage (on the GPU box, from the repo root):  bash tools/pmc_passes.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
find $OUT -name "*counter_collection.csv" | head
