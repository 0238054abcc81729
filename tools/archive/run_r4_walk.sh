#!/bin/bash
# round 4: the 2^20 ring slab in the "walk" layout ([tile][q/4][k2][q%4][c]: every wave of pass 1 stores a contiguous 16 KiB
# front to back; pass 2 reads 512-byte pieces) against the shipped layout ([tile][K1][c]), one library each, interleaved
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4walk; mkdir -p $O
FWA_LAB_LIBRARY=$PWD/tools/variants/walk.so timeout -k 10 200 python3 tools/variant_parity_check.py
for rep in 1 2 3; do
  timeout -k 10 200 python3 tools/sweep.py --lg 20 --batch 4096 --reps 5 --set "" --set "streams=1" | sed 's/^/shipped /' >> $O/ab.txt
  FWA_LAB_LIBRARY=$PWD/tools/variants/walk.so timeout -k 10 200 python3 tools/sweep.py --lab --lg 20 --batch 4096 --reps 5 --set "" --set "streams=1" | sed 's/^/walk    /' >> $O/ab.txt
done
python3 - <<PY
import json
for l in open("$O/ab.txt"):
    tag, js = l[:8], l[8:]
    d = json.loads(js)
    print(tag, d["setting"], d["ms"], d["ms_min"], d["roofline_frac"])
PY
