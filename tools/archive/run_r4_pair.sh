#!/bin/bash
# round 4: block -> tile maps of the 2^20 kernels in which the two resident workgroups of a CU take adjacent tiles (pair1) or a
# control permutation (pair2), against the shipped XCD-contiguous map; one library each, interleaved
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4pair; mkdir -p $O; rm -f $O/ab.txt
for v in pair1 pair_only_P1 pair_only_P2; do FWA_LAB_LIBRARY=$PWD/tools/variants/$v.so timeout -k 10 200 python3 tools/variant_parity_check.py; done
for rep in 1 2 3; do
  timeout -k 10 200 python3 tools/sweep.py --lg 20 --batch 4096 --reps 5 --set "" --set "streams=1" | sed 's/^/shipped /' >> $O/ab.txt
  for v in pair1 pair_only_P1 pair_only_P2; do
    FWA_LAB_LIBRARY=$PWD/tools/variants/$v.so timeout -k 10 200 python3 tools/sweep.py --lab --lg 20 --batch 4096 --reps 5 --set "" --set "streams=1" | sed "s/^/$(printf '%-8.8s' ${v#pair_})/" >> $O/ab.txt
  done
done
python3 - <<PY
import json
for l in open("$O/ab.txt"):
    d = json.loads(l[8:]); print(l[:8], "%-10s" % d["setting"], d["ms"], d["ms_min"], d["roofline_frac"])
PY
