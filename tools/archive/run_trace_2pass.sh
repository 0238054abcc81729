#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/trace_2pass.txt; : > $O
for lg in 16 17 18 19 20 22 24; do
  rm -rf gpurun_out/tr_$lg
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$lg -- python3 tools/one_exec.py --lg $lg --batch $((1 << (28 - lg))) --execs 4 > gpurun_out/tr_$lg.log 2>&1
  echo "== lg $lg" >> $O
  python3 tools/trace_summary.py gpurun_out/tr_$lg >> $O
  rm -rf gpurun_out/tr_$lg
done
