#!/bin/bash
# round 4, job C: the sweep launch shape (64-KiB chunk per workgroup, each wave walks 16 KiB) in k_chunk, k_scale, k_copy
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixture or sweep or chunk or ormalize or copy or known or fuzz or kinds or empty or ragged or identity or batch_independ" > $O/pytest_small.log 2>&1 || { tail -60 $O/pytest_small.log; exit 1; }
tail -3 $O/pytest_small.log
timeout -k 10 400 python3 tools/size_bench.py --lg-min 1 --lg-max 12 --no-latency-shapes > $O/size_small_2GiB.jsonl 2>&1
timeout -k 10 400 python3 tools/size_bench.py --lg-min 1 --lg-max 10 --total-lg 32 --no-latency-shapes > $O/size_small_32GiB.jsonl 2>&1
timeout -k 10 200 python3 tools/kinds_bench.py > $O/kinds_bench.jsonl 2>&1
cut -c1-40,100-200 $O/size_small_2GiB.jsonl; cut -c1-40,100-200 $O/size_small_32GiB.jsonl; grep Normalize $O/kinds_bench.jsonl
echo done
