set -x
cd $GRAFT_REPO_ROOT
for w in ilqr ddp n64; do
  python bench.py --workload $w --force-dist > gpurun_out/r03_b_$w.json 2> gpurun_out/r03_b_$w.err || { tail -20 gpurun_out/r03_b_$w.err; exit 1; }
done
python bench.py --workload n64 --scaling strong --force-dist > gpurun_out/r03_b_n64_strong.json 2> gpurun_out/r03_b_n64_strong.err || { tail -20 gpurun_out/r03_b_n64_strong.err; exit 1; }
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_b_default.json 2> gpurun_out/r03_b_default.err || { tail -20 gpurun_out/r03_b_default.err; exit 1; }
