#!/usr/bin/env python3
"""tools/chain_phase.py DIR -- from a rocprofv3 kernel trace of the two-chain 2^20 pipeline: how are the two chains phased?
For every k_p1_1m launch: which kernels overlap it in time (fraction of its duration spent beside a k_p1_1m / k_p2_1m of the
other chain, or alone)."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_p1_1m" in k or "k_p2_1m" in k:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "p1" if "k_p1_1m" in k else "p2", r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
rows = rows[len(rows) // 4: 3 * len(rows) // 4]   # steady state
tot = {"p1": 0, "p2": 0, "alone": 0}
dur = 0
for i, (s, e, k, q) in enumerate(rows):
    if k != "p1":
        continue
    dur += e - s
    covered = []
    for (s2, e2, k2, q2) in rows[max(0, i - 6): i + 7]:
        if (s2, e2, k2, q2) == (s, e, k, q):
            continue
        a, b = max(s, s2), min(e, e2)
        if b > a:
            tot[k2] += b - a
            covered.append((a, b))
    covered.sort()
    c, last = 0, s
    for a, b in covered:
        a = max(a, last)
        if b > a:
            c += b - a; last = b
    tot["alone"] += (e - s) - c
print("steady-state k_p1_1m launches: %d, mean duration %.1f us" % (sum(1 for r in rows if r[2] == "p1"), dur / max(1, sum(1 for r in rows if r[2] == "p1")) / 1e3))
print("share of a k_p1_1m launch spent beside another k_p1_1m: %.2f, beside a k_p2_1m: %.2f, alone: %.2f" % (tot["p1"] / dur, tot["p2"] / dur, tot["alone"] / dur))
t0 = rows[0][0]
for s, e, k, q in rows[:16]:
    print("  %9.1f us  +%6.1f  %s  queue %s" % ((s - t0) / 1e3, (e - s) / 1e3, k, q))
