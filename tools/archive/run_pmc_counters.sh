#!/bin/bash
# tools/run_pmc_counters.sh LG BATCH "SETTING" OUTFILE -- SQ / LDS / TCP / TCC counters of whatever kernels one plan
# setting launches (one counter group per rocprofv3 pass of the same command; --pmc with --kernel-trace only).
set -e
LG=$1; BATCH=$2; SETTING=$3; O=$4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
echo "# lg $LG batch $BATCH setting '$SETTING'" >> $O
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_STALL_sum"; do
  rm -rf gpurun_out/pmc_q
  if timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_q -- python3 tools/one_exec.py --lg $LG --batch $BATCH --execs 2 --set "$SETTING" > gpurun_out/pmc_q.log 2>&1; then
    echo "== $c" >> $O
    python3 tools/pmc_summary.py gpurun_out/pmc_q | grep -v "k_fill\|copyBuffer" >> $O
  else
    echo "== $c : not collected" >> $O
  fi
  rm -rf gpurun_out/pmc_q
done
