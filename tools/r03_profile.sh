set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail -5 gpurun_out/r03_bench_default.err; exit 1; }
bash tools/profile_all.sh r03_prof > gpurun_out/r03_profile_all.txt 2>&1
tail -30 gpurun_out/r03_profile_all.txt
bash tools/pmc_passes.sh r03_pmc > gpurun_out/r03_pmc_passes.txt 2>&1
python tools/pmc_summary.py gpurun_out/r03_pmc lqr_backward_dma > gpurun_out/r03_k1_pmc.txt 2>&1
cat gpurun_out/r03_k1_pmc.txt
