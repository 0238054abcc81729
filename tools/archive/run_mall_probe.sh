#!/bin/bash
# 2^20 two-pass pipeline on batches small enough for input + slab to live in the Infinity Cache: what the passes do
# when NO byte has to come from HBM (upper bound of any scheme that keeps the slab cache-resident)
set -e
mkdir -p gpurun_out
O=gpurun_out/mall_probe.jsonl; : > $O
for b in 4 8 12 16 24 32 64 128 256; do
  timeout -k 10 150 python tools/sweep.py --lg 20 --batch $b --reps 21 --set "" --set "group=4" --set "group=8" --set "streams=1" | sed "s/^{/{\"b\": $b, /" >> $O
done
