#!/bin/bash
# round 4: (group, chains) re-swept for the two-pass tiled sizes with this round's block maps (32-GiB footprint)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4groups; mkdir -p $O; rm -f $O/sweep.jsonl
for lg in 16 17 18 19 21 22; do
  b=$((1 << (32 - lg))); g=$((1 << (24 - lg)))   # default group: 128 MiB of transforms
  timeout -k 10 300 python3 tools/sweep.py --lg $lg --batch $b --reps 5 --set "" --set "group=$((g/2)),streams=2" --set "group=$((g/2)),streams=3" --set "group=$((g/2)),streams=4" --set "group=$((g*3/4)),streams=2" --set "group=$g,streams=3" --set "group=$((g/4)),streams=4" >> $O/sweep.jsonl
done
python3 - <<PY
import json
for l in open("$O/sweep.jsonl"):
    d = json.loads(l); print(d["lg_n"], "%-26s" % d["setting"], d["group"], d["streams"], d["xcd_swizzle"], d["ms"], d["ms_min"], d["roofline_frac"])
PY
