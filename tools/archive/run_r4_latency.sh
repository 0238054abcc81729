#!/bin/bash
# tools/run_r4_latency.sh -- VERDICT round 3 item 3: the batch-1 shapes on HEAD against the round-2 tree (tools/variants/r2,
# `git archive fe0473a` built in place), interleaved in one job on one box, then HIP-API + kernel traces of C2 on both.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4lat
mkdir -p $O
for rep in 1 2; do
  timeout -k 10 120 python3 tools/latency_shapes.py --label head >> $O/latency_head.jsonl 2>> $O/err.log
  timeout -k 10 120 python3 tools/latency_shapes.py --tree tools/variants/r2 --label r2 >> $O/latency_r2.jsonl 2>> $O/err.log
done
timeout -k 10 200 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/trace_head -- python3 tools/latency_shapes.py --shapes 20x1 --label head > $O/trace_head.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/trace_r2 -- python3 tools/latency_shapes.py --shapes 20x1 --tree tools/variants/r2 --label r2 > $O/trace_r2.log 2>&1
for t in head r2; do
  python3 tools/hip_trace_summary.py $O/trace_$t > $O/hip_summary_$t.txt
  python3 tools/trace_summary.py $O/trace_$t > $O/kernel_summary_$t.txt
  python3 tools/trace_timeline.py $O/trace_$t 16 > $O/timeline_$t.txt
  rm -rf $O/trace_$t
done
echo done
