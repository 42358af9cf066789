set -x
cd $GRAFT_REPO_ROOT
bash tools/pmc_sweeps.sh r03_pmc_sweeps2 > gpurun_out/r03_pmc_sweeps2.txt 2>&1
tail -80 gpurun_out/r03_pmc_sweeps2.txt
bash tools/pmc_mpc.sh r03_pmc_mpc > gpurun_out/r03_pmc_mpc.txt 2>&1
tail -25 gpurun_out/r03_pmc_mpc.txt
