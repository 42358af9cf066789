set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --gpus 2 --share-gpu --backend gloo --steps 20 --warmup 5 > gpurun_out/r03_share2_default.json 2> gpurun_out/r03_share2_default.err || { tail -30 gpurun_out/r03_share2_default.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 3 --share-gpu --backend gloo --workload ilqr --scaling strong > gpurun_out/r03_share3_ilqr_strong.json 2> gpurun_out/r03_share3_ilqr_strong.err || { tail -30 gpurun_out/r03_share3_ilqr_strong.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --backend gloo --workload lqr --scaling strong > gpurun_out/r03_share2_lqr_strong.json 2> gpurun_out/r03_share2_lqr_strong.err || { tail -30 gpurun_out/r03_share2_lqr_strong.err; exit 1; }
wc -c gpurun_out/r03_share*.json
