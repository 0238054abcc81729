#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rows32 or tiled_groups or refactorised or first_pass_1024 or fuzz" > gpurun_out/rows32_tests.txt 2>&1
O=gpurun_out/rows32_sweep.jsonl; : > $O
sw() { lg=$1; shift; b=$((1 << (28 - lg))); args=(); for s in "$@"; do args+=(--set "$s"); done
  timeout -k 10 150 python tools/sweep.py --lg $lg --batch $b --reps 7 "${args[@]}" | sed "s/^{/{\"lg\": $lg, /" >> $O; }
sw 19 "" "rows32=0" "factors=9.10" "factors=9.10,rows32=0" "factors=8.11"
sw 20 "" "path=7,factors=10.10" "path=7,factors=10.10,rows32=0" "path=7,factors=9.11"
sw 21 "" "factors=7.7.7" "factors=10.11,group=4" "factors=10.11,group=16"
sw 18 "" "factors=9.9" "factors=9.9,rows32=0" "factors=7.11"
sw 17 "" "factors=8.9" "factors=7.10" "factors=6.11"
