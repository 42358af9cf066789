#!/usr/bin/env python3
"""Per-iteration timeline of one iLQR / DDP solve from a rocprofv3 --kernel-trace CSV.
usage: timeline_solve.py <kernel_trace.csv> [n_last_kernels_is_one_solve: --reps R]
Groups the kernels of the LAST solve (between the last two long idle gaps) by outer iteration (one `rollout_ls_fast_kernel` launch each),
prints per-iteration: active kernel time, idle gaps between kernels, and the per-kernel split for a few iterations."""
import csv
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    n = n.replace("zm::", "").replace("void ", "")
    return n.split("<")[0][:34]


def main():
    rows = []
    if sys.argv[1].endswith(".db"):      # rocprofv3's default rocpd output (sqlite)
        import sqlite3
        for s_, e_, n_, g_ in sqlite3.connect(sys.argv[1]).execute("select start, end, name, grid_x from kernels"):
            rows.append((int(s_), int(e_), short(n_), int(g_)))
    else:
        with open(sys.argv[1]) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r.get("Grid_Size_X", 0) or 0)))
    rows.sort()
    # split into bursts by idle gaps > 2 ms; take the longest burst = one full-size solve
    bursts, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - a[1] > 2_000_000:
            bursts.append(cur)
            cur = []
        cur.append(b)
    bursts.append(cur)
    solve = max(bursts, key=lambda b: b[-1][1] - b[0][0])
    t0 = solve[0][0]
    wall = (solve[-1][1] - t0) / 1e6
    busy = sum(e - s for s, e, *_ in solve) / 1e6
    print(f"solve: {len(solve)} kernels, wall {wall:.2f} ms, kernel-busy {busy:.2f} ms, idle {wall - busy:.2f} ms")
    tot = defaultdict(lambda: [0, 0.0])
    for s, e, n, _g in solve:
        tot[n][0] += 1
        tot[n][1] += (e - s) / 1e6
    for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:36s} {c:5d} launches {ms:8.3f} ms  ({ms / c * 1e3:7.1f} us avg)")
    # iterations: split at each rollout launch
    its, cur = [], []
    for k in solve:
        cur.append(k)
        if k[2].startswith("rollout_ls_fast"):
            its.append(cur)
            cur = []
    if cur:
        its.append(cur)
    print("iteration: wall_us busy_us | per-kernel us")
    prev_end = t0
    for i, it in enumerate(its):
        w = (it[-1][1] - prev_end) / 1e3
        b = sum(e - s for s, e, *_ in it) / 1e3
        prev_end = it[-1][1]
        if i < 6 or i % 10 == 0 or i >= len(its) - 3:
            print(f"  {i:4d}: {w:8.1f} {b:8.1f} | " + " ".join(f"{n[:12]}[{g}]={(e - s) / 1e3:.0f}" for s, e, n, g in it))


if __name__ == "__main__":
    main()
