#!/usr/bin/env python3
"""tools/pmc_summary.py DIR -- sum rocprofv3 counter_collection CSVs per kernel name and counter (value sums and dispatch counts)."""
import csv, glob, os, sys, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r.get("Kernel_Name", "?").split("(")[0][:60], r.get("Counter_Name", "?"))
        tot[k][0] += float(r.get("Counter_Value", 0)); tot[k][1] += 1
for (kn, cn), (v, c) in sorted(tot.items()):
    print(f"{kn:62s} {cn:12s} dispatches {c:6d}  sum {v:.6g}  per dispatch {v / max(c, 1):.6g}")
