#!/bin/bash
# latency regime: few transforms of 2^16 .. 2^21 -- 1024-point first pass (few, fat workgroups) against balanced factors
set -e
mkdir -p gpurun_out
O=gpurun_out/small_batch.jsonl; : > $O
sw() { lg=$1; b=$2; shift; shift; args=(); for s in "$@"; do args+=(--set "$s"); done
  timeout -k 10 150 python tools/sweep.py --lg $lg --batch $b --reps 31 "${args[@]}" | sed "s/^{/{\"lg\": $lg, /" >> $O; }
for b in 1 4 16 64; do
sw 16 $b "" "factors=8.8" "factors=6.10" "path=8"
sw 17 $b "" "factors=8.9" "factors=9.8" "path=8"
sw 18 $b "" "factors=9.9" "factors=6.6.6" "path=8"
sw 19 $b "" "factors=9.10" "factors=6.6.7"
done
for b in 1 4; do
sw 21 $b "" "factors=7.7.7"
done
