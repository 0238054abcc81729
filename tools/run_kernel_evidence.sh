#!/bin/bash
# tools/run_kernel_evidence.sh OUT -- rocprofv3 evidence for the one-launch kernels that changed in round 5: per-kernel
# durations (--kernel-trace --stats) and L2<->fabric traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes) of three execs
# over 2^32 samples at n = 64, 512, 1024, 2048, 4096 (profiles/round5/kernel_evidence_one_launch_with_wave512.txt is the
# same job while the library had k_wave512: its "default" entry at 512; "wave=0" there is k_small32<9>).
set -e
O=${1:-gpurun_out/kernel_evidence.txt}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$(dirname $O)"; : > $O
run() {  # lg, setting
  local lg=$1 set=$2 b=$((1 << (32 - $1)))
  echo "== n = 2^$lg x $b (${set:-default}): 3 execs; algorithmic bytes per exec 34359738.4 KB read + as much written" >> $O
  rm -rf gpurun_out/ke
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ke -- python3 tools/one_exec.py --lg $lg --batch $b --execs 3 --set "$set" > gpurun_out/ke.log 2>&1
  python3 tools/trace_summary.py gpurun_out/ke | grep -v "k_fill\|k_spin" >> $O
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/ke
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/ke -- python3 tools/one_exec.py --lg $lg --batch $b --execs 3 --set "$set" > gpurun_out/ke.log 2>&1
    python3 tools/pmc_summary.py gpurun_out/ke | grep -v "k_fill\|k_spin" >> $O
  done
  rm -rf gpurun_out/ke
}
run 6 ""
run 9 ""
run 10 ""
run 11 ""
run 12 ""
echo done >> $O
