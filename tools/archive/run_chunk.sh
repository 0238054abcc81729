#!/bin/bash
# k_chunk (small_reg = 1) against the direct-addressing kernels (small_reg = 3) for n = 4 .. 256
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixture_sizes or size_sweep or known_answers or fuzz or impulse or onlyinverse" > gpurun_out/chunk_tests.txt 2>&1
O=gpurun_out/chunk_sweep.jsonl; : > $O
for lg in 2 3 4 5 6 7 8; do
  timeout -k 10 120 python tools/sweep.py --lg $lg --batch $((1 << (28 - lg))) --reps 7 --set "small_reg=1" --set "small_reg=3" | sed "s/^{/{\"lg\": $lg, /" >> $O
done
