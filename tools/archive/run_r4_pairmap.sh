#!/bin/bash
# round 4: the pair map (xcd_swizzle bit 2) as default of one-chain 2^20 plans: bit-identity, then mid batches A/B
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4pairmap; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "n1m or c2_n1m or config_c3 or graph" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
rm -f $O/sweep.jsonl
for b in 16 32 48 64 128 4096; do
  timeout -k 10 200 python3 tools/sweep.py --lg 20 --batch $b --reps 9 --set "streams=1,xcd_swizzle=1" --set "streams=1,xcd_swizzle=5" --set "" >> $O/sweep.jsonl
done
python3 - <<PY
import json
for l in open("$O/sweep.jsonl"):
    d = json.loads(l); print(d["batch"], "%-26s" % d["setting"], "streams", d["streams"], "swz", d["xcd_swizzle"], d["ms"], d["ms_min"], d["roofline_frac"])
PY
