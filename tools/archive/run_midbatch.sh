#!/bin/bash
# mid-size batches: default pipeline geometry against the old default (two chains always, group 16 / 128 MiB)
set -e
mkdir -p gpurun_out
O=gpurun_out/midbatch.jsonl; : > $O
for b in 16 24 32 48 64 96 128 192 256 512; do
  timeout -k 10 150 python tools/sweep.py --lg 20 --batch $b --reps 15 --set "" --set "group=16,streams=2" --set "group=8,streams=2" --set "group=16,streams=1" | sed "s/^{/{\"b\": $b, /" >> $O
done
for b in 64 128 192 256 384 512 1024; do
  timeout -k 10 150 python tools/sweep.py --lg 18 --batch $b --reps 15 --set "" --set "streams=2" --set "streams=1" --set "group=32,streams=2" | sed "s/^{/{\"b\": $b, /" >> $O
done
for b in 256 512 1024 2048 4096; do
  timeout -k 10 150 python tools/sweep.py --lg 16 --batch $b --reps 15 --set "" --set "streams=2" --set "streams=1" --set "group=128,streams=2" | sed "s/^{/{\"b\": $b, /" >> $O
done
for b in 2 4 6 8 16; do
  timeout -k 10 150 python tools/sweep.py --lg 24 --batch $b --reps 9 --set "" --set "streams=2" --set "streams=1" | sed "s/^{/{\"b\": $b, /" >> $O
done
