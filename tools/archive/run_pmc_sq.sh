#!/bin/bash
# SQ / LDS / memory-unit counters of the 2^20 pass kernels (one counter group per pass of the same command)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/pmc_sq.txt; : > $O
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf gpurun_out/pmc_q
  if timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_q -- python3 tools/one_exec.py --lg 20 --batch 512 --execs 2 --set "streams=1" > gpurun_out/pmc_q.log 2>&1; then
    echo "== $c" >> $O
    python3 tools/pmc_summary.py gpurun_out/pmc_q | grep -v "k_fill\|copyBuffer" >> $O
  else
    echo "== $c : not collected" >> $O
  fi
  rm -rf gpurun_out/pmc_q
done
