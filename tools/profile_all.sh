#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py and the secondary benchmarks; copies the kernel_stats CSVs to gpurun_out/<dir>.
# usage (GPU box, repo root):  bash tools/profile_all.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp
run() {  # name, script args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 "$@" > $OUT/$name.log 2>&1 || echo "$name failed"
  local f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
  echo "== $name"; head -4 $OUT/${name}_kernel_stats.csv | cut -c1-160
}
run bench     $R/bench.py --no-cpu-baseline --no-secondary
run ilqr      $R/tools/bench_ilqr.py --reps 1
run mpc       $R/tools/bench_mpc.py --eps 1e-2
run tiled     $R/tools/bench_lqr_tiled.py --batch 2048 --reps 14     # >= 10 launches: the average is a steady-state one, not a cold-clock sample
run wide      $R/tools/bench_sweep_tiled.py --reps 6
run sweeps    $R/tools/bench_ilqr_backward.py --reps 3
run ddp       $R/tools/bench_ilqr.py --ddp --reps 1
run psd       $R/tools/bench_psd.py
grep -h "^{" $OUT/*.log | cut -c1-400 > $OUT/results.jsonl
