#!/usr/bin/env python3
"""Condense the counter CSVs written by scripts/footprint.sh: per (workload, size, counter) the mean value per dispatch of
the op kernel.  python scripts/summarize_footprint.py gpurun_out/footprint_r02 [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_"
rows = defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    tag = os.path.relpath(path, root).split(os.sep)[0]
    wl, lg, _pass = tag.split("_", 2)
    with open(path) as f:
        for r in csv.DictReader(f):
            if want in r["Kernel_Name"] and "rtdp::" in r["Kernel_Name"]:
                rows[(wl, lg, r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("| workload | log2 units | kernel | counter | mean per dispatch | dispatches |")
print("|---|---|---|---|---|---|")
for (wl, lg, k, c), v in sorted(rows.items()):
    print("| %s | %s | `%s` | %s | %.4g | %d |" % (wl, lg, k, c, sum(v) / len(v), len(v)))
