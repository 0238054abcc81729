#!/bin/bash
# p1_gen A/B at the sizes whose first factor is (or can be) 1024
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "first_pass_1024 or tiled_groups or refactorised" > gpurun_out/p1gen_tests.txt 2>&1
O=gpurun_out/p1gen_sweep.jsonl; : > $O
timeout -k 10 120 python tools/sweep.py --lg 18 --batch 1024 --reps 7 --set "p1_gen=1" --set "p1_gen=0" --set "factors=9.9" >> $O
timeout -k 10 120 python tools/sweep.py --lg 19 --batch 512 --reps 7 --set "p1_gen=1" --set "p1_gen=0" --set "factors=9.10" >> $O
timeout -k 10 120 python tools/sweep.py --lg 17 --batch 2048 --reps 7 --set "" --set "factors=10.7,p1_gen=1" --set "factors=10.7,p1_gen=0" >> $O
timeout -k 10 120 python tools/sweep.py --lg 16 --batch 4096 --reps 7 --set "" --set "factors=10.6,p1_gen=1" --set "factors=10.6,p1_gen=0" >> $O
timeout -k 10 120 python tools/sweep.py --lg 21 --batch 128 --reps 7 --set "" --set "factors=10.6.5" --set "factors=10.6.5,p1_gen=0" >> $O || true
timeout -k 10 120 python tools/sweep.py --lg 22 --batch 64 --reps 7 --set "" --set "factors=10.6.6" --set "factors=10.6.6,p1_gen=0" >> $O
timeout -k 10 120 python tools/sweep.py --lg 24 --batch 16 --reps 7 --set "" --set "factors=10.7.7" --set "factors=10.7.7,p1_gen=0" >> $O
timeout -k 10 120 python tools/sweep.py --lg 26 --batch 4 --reps 7 --set "" --set "factors=10.8.8" --set "factors=10.8.8,p1_gen=0" >> $O
timeout -k 10 120 python tools/sweep.py --lg 28 --batch 1 --reps 7 --set "" --set "factors=10.9.9" --set "factors=10.9.9,p1_gen=0" >> $O
timeout -k 10 120 python tools/sweep.py --lg 20 --batch 2048 --reps 5 --set "" --set "path=7,factors=10.10,p1_gen=1" >> $O || true
