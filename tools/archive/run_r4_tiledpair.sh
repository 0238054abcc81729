#!/bin/bash
# round 4: the XCD-contiguous (bit 0) and CU-pair (bit 2) block maps on the TILED plans (default there: 0), 32-GiB footprint
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4tiledpair; mkdir -p $O; rm -f $O/sweep.jsonl
for lg in 16 18 19 21 22 23 24; do
  b=$((1 << (32 - lg)))
  timeout -k 10 300 python3 tools/sweep.py --lg $lg --batch $b --reps 5 --set "" --set "xcd_swizzle=1" --set "xcd_swizzle=5" --set "streams=1" --set "streams=1,xcd_swizzle=5" >> $O/sweep.jsonl
done
python3 - <<PY
import json
for l in open("$O/sweep.jsonl"):
    d = json.loads(l); print(d["lg_n"], "%-26s" % d["setting"], d["group"], d["streams"], d["ms"], d["ms_min"], d["roofline_frac"])
PY
