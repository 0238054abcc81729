#!/bin/bash
# L2<->fabric traffic (FETCH_SIZE x2 + WRITE_SIZE, KB) per exec for one size of every kernel family, 2 GiB of samples each
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/pmc_sizes.txt; : > $O
for lg in 4 10 15 18 21 24; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_s
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_s -- python3 tools/one_exec.py --lg $lg --batch $((1 << (28 - lg))) --execs 2 > gpurun_out/pmc_s.log 2>&1
    echo "== lg $lg $c (2 execs of 2^28 samples = 2 x 2147483.6 KB read + as much written algorithmically)" >> $O
    python3 tools/pmc_summary.py gpurun_out/pmc_s >> $O
    rm -rf gpurun_out/pmc_s
  done
done
