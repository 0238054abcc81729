#!/bin/bash
# factorisation sweep for the three-pass sizes with a 1024-point first pass (k_p1_gen) against the balanced default
set -e
mkdir -p gpurun_out
O=gpurun_out/factor_sweep.jsonl; : > $O
sw() { lg=$1; shift; b=$((1 << (28 - lg))); [ $b -lt 1 ] && b=1; args=(); for s in "$@"; do args+=(--set "$s"); done
  timeout -k 10 150 python tools/sweep.py --lg $lg --batch $b --reps 7 "${args[@]}" | sed "s/^{/{\"lg\": $lg, /" >> $O; }
sw 22 "" "factors=10.6.6" "factors=6.8.8" "factors=8.8.6" "factors=8.6.8"
sw 23 "" "factors=10.6.7" "factors=10.7.6" "factors=7.8.8" "factors=8.8.7"
sw 24 "" "factors=10.7.7" "factors=10.6.8" "factors=10.8.6"
sw 25 "" "factors=10.7.8" "factors=10.8.7" "factors=10.6.9" "factors=10.9.6"
sw 26 "" "factors=10.8.8" "factors=10.7.9" "factors=10.9.7" "factors=10.6.10"
sw 27 "" "factors=10.8.9" "factors=10.9.8" "factors=10.7.10" "factors=10.10.7"
sw 28 "" "factors=10.9.9" "factors=10.8.10" "factors=10.10.8"
# latency cases
timeout -k 10 100 python tools/sweep.py --lg 24 --batch 1 --reps 21 --set "" --set "factors=10.7.7" --set "factors=10.6.8" | sed "s/^{/{\"lg\": 24, \"c\": 5, /" >> $O
timeout -k 10 100 python tools/sweep.py --lg 20 --batch 1 --reps 21 --set "" --set "factors=10.10" --set "factors=10.10,p1_gen=0" | sed "s/^{/{\"lg\": 20, \"c\": 2, /" >> $O
