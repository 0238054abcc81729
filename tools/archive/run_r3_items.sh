#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3items; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "bench_distributed or cpp_ or product_library or failed_retune or laboratory or team or persistent" -s > $O/tests.txt 2>&1; rc=$?
tail -25 $O/tests.txt; grep -n "GB/s\|iterations" $O/tests.txt | head
exit $rc
