#!/bin/bash
# tools/run_kernel_evidence.sh OUT "LG LG ..." -- rocprofv3 evidence for one size per kernel family of DESIGN.md section 2 on
# the tree as it is: per size, three passes of `tools/one_exec.py --execs 3` over 2^32 samples --
# `--kernel-trace --stats` (kernel names, durations, the HIP-event time of every exec) and one `--pmc` pass each for
# FETCH_SIZE and WRITE_SIZE (never combined with a trace domain other than the kernel trace).  "norm" = Normalize at n = 1024.
# Raw summaries are appended to OUT; tools/kernel_evidence_table.py turns them into the table of
# profiles/round6/kernel_evidence_all_families.txt.  One pass per size: no sweeps.
set -e
O=${1:-gpurun_out/kernel_evidence_raw.txt}
SIZES=${2:-"6 9 10 13 14 15 17 19 21 22 23 24 norm"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$(dirname $O)"; touch $O
run() {  # lg | norm
  local lg=$1 kind=forward
  if [ "$lg" = norm ]; then lg=10; kind=normalize; fi
  local b=$((1 << (32 - lg)))
  echo "== $kind n = 2^$lg x $b: 3 execs; algorithmic bytes per exec 34359738.4 KB read + as much written" >> $O
  rm -rf gpurun_out/ke
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ke -- python3 tools/one_exec.py --lg $lg --batch $b --execs 3 --kind $kind > gpurun_out/ke.log 2>&1
  grep '^{"one_exec"' gpurun_out/ke.log >> $O
  python3 tools/trace_summary.py gpurun_out/ke | grep -v "k_fill\|k_spin\|rocclr" >> $O
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/ke
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/ke -- python3 tools/one_exec.py --lg $lg --batch $b --execs 3 --kind $kind > gpurun_out/ke.log 2>&1
    python3 tools/pmc_summary.py gpurun_out/ke | grep -v "k_fill\|k_spin\|rocclr" >> $O
  done
  rm -rf gpurun_out/ke
  echo "evidence: $kind 2^$lg done"
}
for s in $SIZES; do run $s; done
echo "done $SIZES" >> $O
