set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r03_tl8192_ilqr -o tl -- python3 $R/tools/bench_ilqr.py --reps 1 --batch 8192 > $R/gpurun_out/r03_tl8192_ilqr.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/r03_tl8192_ilqr_launches.txt
import sqlite3, sys
sys.path.insert(0, "tools")
from timeline_solve import short
rows = [(int(s), int(e), short(n), int(g)) for s, e, n, g in sqlite3.connect("gpurun_out/r03_tl8192_ilqr/tl_results.db").execute("select start, end, name, grid_x from kernels")]
rows.sort()
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 2_000_000:
        bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
solve = max(bursts, key=lambda b: b[-1][1] - b[0][0])
t0 = solve[0][0]
prev = t0
for s, e, n, g in solve:
    print(f"{(s - t0) / 1e3:10.1f} us  gap {(s - prev) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  grid {g:8d}  {n}")
    prev = e
PY
python3 tools/timeline_solve.py gpurun_out/r03_tl8192_ilqr/tl_results.db > gpurun_out/r03_tl8192_ilqr.txt
head -20 gpurun_out/r03_tl8192_ilqr.txt
