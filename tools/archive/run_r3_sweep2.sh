#!/bin/bash
# round 3: (group, chains) x pass-A kernel at C3; factorisations of 2^21 .. 2^24 at the 32-GiB footprint
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3sweep2; mkdir -p $O
S=$O/sweep_c3_group_chains_colsw.jsonl; : > $S
timeout -k 10 300 python3 tools/sweep.py --lg 20 --batch 4096 --reps 5 --set "" --set "group=16,streams=3" --set "group=8,streams=4" --set "group=8,streams=3" --set "group=12,streams=2" --set "group=10,streams=3" --set "group=4,streams=8" --set "group=2,streams=8" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 20 --batch 4096 --reps 5 --set "factors=9.11,colsw=1,tile_ring=0" --set "factors=9.11,colsw=1,tile_ring=0,group=16,streams=3" --set "factors=9.11,colsw=1,tile_ring=0,group=8,streams=4" --set "factors=9.11,colsw=1,tile_ring=0,group=8,streams=3" --set "factors=9.11,colsw=1,tile_ring=0,group=12,streams=2" >> $S 2>&1
S=$O/sweep_factors_21_24_32GiB.jsonl; : > $S
timeout -k 10 300 python3 tools/sweep.py --lg 21 --batch 2048 --reps 5 --set "" --set "factors=11.10" --set "factors=9.12,colsw=1" --set "factors=8.7.6,colsw=1" --set "factors=9.6.6,colsw=1" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 22 --batch 1024 --reps 5 --set "" --set "factors=11.11" --set "factors=8.7.7,colsw=1" --set "factors=9.7.6,colsw=1" --set "factors=10.6.6" --set "factors=8.8.6,colsw=1" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 23 --batch 512 --reps 5 --set "" --set "factors=8.8.7,colsw=1" --set "factors=9.7.7,colsw=1" --set "factors=10.7.6" --set "factors=8.7.8,colsw=1" >> $S 2>&1 && \
timeout -k 10 300 python3 tools/sweep.py --lg 24 --batch 256 --reps 5 --set "" --set "factors=8.8.8,colsw=1" --set "factors=9.8.7,colsw=1" --set "factors=9.7.8,colsw=1" --set "factors=8.9.7,colsw=1" >> $S 2>&1
echo rc=$?
