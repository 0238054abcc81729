#!/bin/bash
# round 3, GPU batch 1: allocation-kind probe, two-pass size class at C3's footprint (item 3), 2^14/2^15 counters (item 4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b1; mkdir -p $O
timeout -k 10 150 tools/fabric_probe3 256 > $O/probe_alloc_kinds_ring256.txt 2>&1 && \
timeout -k 10 150 tools/fabric_probe3 64 > $O/probe_alloc_kinds_ring64.txt 2>&1 && \
timeout -k 10 300 python3 tools/size_bench.py --lg-min 15 --lg-max 24 --total-lg 32 --no-latency-shapes > $O/size_sweep_32GiB.jsonl 2>&1 && \
timeout -k 10 200 python3 tools/size_bench.py --lg-min 15 --lg-max 24 --total-lg 28 --no-latency-shapes > $O/size_sweep_2GiB.jsonl 2>&1 && \
tools/run_pmc_counters.sh 15 8192 "" $O/pmc_small32_15.txt && \
tools/run_pmc_counters.sh 14 16384 "" $O/pmc_small32_14.txt
echo rc=$?
