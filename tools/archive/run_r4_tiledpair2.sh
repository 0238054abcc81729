#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4tiledpair; mkdir -p $O; rm -f $O/sweep2.jsonl
for rep in 1 2; do
for lg in 16 17 18 19 21; do
  b=$((1 << (32 - lg)))
  timeout -k 10 300 python3 tools/sweep.py --lg $lg --batch $b --reps 5 --set "" --set "xcd_swizzle=1" --set "xcd_swizzle=5" >> $O/sweep2.jsonl
done
done
python3 - <<PY
import json
for l in open("$O/sweep2.jsonl"):
    d = json.loads(l); print(d["lg_n"], "%-26s" % d["setting"], d["group"], d["streams"], d["ms"], d["ms_min"], d["roofline_frac"])
PY
