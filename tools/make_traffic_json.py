#!/usr/bin/env python3
"""profiles/traffic_latest.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh: HBM bytes per launch of the headline
kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes -- FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streaming reads on gfx950 (x2), WRITE_SIZE is exact -- and stamped with the fingerprint of the kernel sources it was
measured on (bench.py quotes the figure only while that fingerprint matches).
usage: python tools/make_traffic_json.py <dir with the counter_collection.csv files> [kernel substring]"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "lqr_backward_dma_f64"
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"] and row["Counter_Name"] in vals:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
assert vals["FETCH_SIZE"] and vals["WRITE_SIZE"], "no FETCH_SIZE / WRITE_SIZE rows for " + kern
fetch_kb = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
write_kb = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
alg = bench.bytes_per_step(12, 4) * 4096 * 50
hbm = int(round((2 * fetch_kb + write_kb) * 1024))
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/pmc_passes.sh), kernel {kern}, bench.py --steps 5 "
                 f"--warmup 2, mean over {len(vals['FETCH_SIZE'])} dispatches",
       "k1_source_sha": bench.k1_source_sha(), "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads -> x2 (MI355X_MICROARCH.md, HBM "
                     "section); WRITE_SIZE exact", "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
       "ratio_traffic_to_algorithmic": hbm / alg}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out))
