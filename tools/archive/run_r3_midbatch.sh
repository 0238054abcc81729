#!/bin/bash
# mid-size batches of the colsw sizes: does the new default regress anywhere against round 2's plan (k_p1_gen, 1024 x N2)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3mid; mkdir -p $O; S=$O/sweep_mid_batches_colsw.jsonl; : > $S
for spec in "16 17 10.6" "16 64 10.6" "16 256 10.6" "16 1024 10.6" "17 9 10.7" "17 64 10.7" "17 512 10.7" "18 5 10.8" "18 16 10.8" "18 64 10.8" "18 256 10.8" "19 3 10.9" "19 16 10.9" "19 128 10.9" "24 2 10.7.7" "24 4 10.7.7" "24 16 10.7.7"; do
  set -- $spec
  timeout -k 10 100 python3 tools/sweep.py --lg $1 --batch $2 --reps 15 --set "" --set "factors=$3" >> $S 2>&1 || exit 1
done
echo rc=$?
