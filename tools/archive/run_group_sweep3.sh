#!/bin/bash
# group / streams sweep for three-pass sizes (slab = group * n * 8 bytes per chain)
set -e
mkdir -p gpurun_out
O=gpurun_out/group_sweep3.jsonl; : > $O
sw() { lg=$1; shift; b=$((1 << (28 - lg))); args=(); for s in "$@"; do args+=(--set "$s"); done
  timeout -k 10 150 python tools/sweep.py --lg $lg --batch $b --reps 7 "${args[@]}" | sed "s/^{/{\"lg\": $lg, /" >> $O; }
sw 21 "" "group=4" "group=2" "group=1" "group=4,streams=3" "group=2,streams=4" "group=16" "group=8,streams=1"
sw 22 "" "group=2" "group=1" "group=2,streams=3" "group=1,streams=4" "group=8" "group=4,streams=1"
sw 23 "" "group=1" "group=1,streams=3" "group=1,streams=4" "group=4" "group=2,streams=1" "group=2,streams=3"
sw 24 "" "streams=1" "streams=3" "streams=4" "group=2" "group=2,streams=1"
sw 18 "" "group=32" "group=16" "group=32,streams=4" "group=128" "group=64,streams=1" "group=64,streams=3"
sw 16 "" "group=128" "group=64" "group=128,streams=4" "group=512" "group=256,streams=1" "group=256,streams=3"
