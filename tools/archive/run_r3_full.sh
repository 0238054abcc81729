#!/bin/bash
# full GPU test suite + size sweeps at both footprints
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3full; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -5 $O/tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/size_bench.py --lg-min 15 --lg-max 24 --total-lg 32 --no-latency-shapes > $O/size_sweep_32GiB.jsonl 2>&1 && \
timeout -k 10 300 python3 tools/size_bench.py --lg-min 1 --lg-max 30 > $O/size_sweep_2GiB.jsonl 2>&1
echo rc=$?
