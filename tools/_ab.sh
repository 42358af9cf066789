cd $GRAFT_REPO_ROOT
python -m pytest tests/test_rollout_gpu.py tests/test_ilqr_tail_gpu.py tests/test_reference_fixtures_gpu.py tests/test_ilqr_solve_gpu.py tests/test_ddp_gpu.py tests/test_simulator_gpu.py -m gpu -q 2>&1 | tail -8
for c in _base ""; do
  echo "== variant $c"
  ZOPT_AMD_LIB=$GRAFT_REPO_ROOT/zopt_amd/csrc/libzopt_amd$c.so python tools/bench_ilqr.py --reps 3 --batch 8192 2>&1 | grep -E "^\{" | cut -c60-230
  ZOPT_AMD_LIB=$GRAFT_REPO_ROOT/zopt_amd/csrc/libzopt_amd$c.so python tools/bench_ilqr.py --reps 3 --batch 8192 --ddp 2>&1 | grep -E "^\{" | cut -c60-250
done
