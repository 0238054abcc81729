#!/bin/bash
# round 3, VERDICT item 1a: k_colsw (512 x 32 / 256 x 64 column tiles) parity + C3-footprint A/B sweeps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3colsw; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "short_wide or rows32_kernel or first_pass_2048" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
S=$O/sweep_colsw.jsonl; : > $S
timeout -k 10 300 python3 tools/sweep.py --lg 20 --batch 4096 --reps 5 --set "" --set "factors=9.11,colsw=1" --set "factors=9.11,colsw=1,tile_ring=0" --set "factors=8.12,colsw=1" --set "factors=8.12,colsw=1,tile_ring=0" --set "factors=9.11" --set "factors=9.11,colsw=1,xcd_swizzle=1" --set "factors=10.10" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 19 --batch 8192 --reps 5 --set "" --set "factors=9.10,colsw=1" --set "factors=8.11,colsw=1" --set "factors=9.10,colsw=1,tile_ring=0" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 18 --batch 16384 --reps 5 --set "" --set "factors=9.9,colsw=1" --set "factors=8.10,colsw=1" --set "factors=9.9" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 21 --batch 2048 --reps 5 --set "" --set "factors=9.12,colsw=1" --set "factors=9.12,colsw=1,tile_ring=0" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 17 --batch 32768 --reps 5 --set "" --set "factors=9.8,colsw=1" --set "factors=8.9,colsw=1" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 16 --batch 65536 --reps 5 --set "" --set "factors=8.8,colsw=1" --set "factors=9.7,colsw=1" >> $S 2>&1
echo rc=$?
