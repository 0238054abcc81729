#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3skew; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "rows32 or short_wide or first_pass or tiled or size_sweep or c5 or refactorised or small_tiles" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
: > $O/summary.txt
run() { rm -rf $O/t; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/one_exec.py --lg $1 --batch $2 --execs 3 --set "$3" > $O/t.log 2>&1 || return 1; echo "== 2^$1 x $2  $3" >> $O/summary.txt; python3 tools/trace_summary.py $O/t | grep -v "k_fill\|copyBuffer\|k_spin" >> $O/summary.txt; }
run 19 2048 "streams=1" && run 18 4096 "streams=1" && run 21 512 "streams=1" && run 22 256 "streams=1" && run 23 128 "streams=1" && run 17 8192 "streams=1"
rm -rf $O/t; cat $O/summary.txt
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do for spec in "21 256" "23 64" "19 1024"; do set -- $spec; rm -rf gpurun_out/pmc_q; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_q -- python3 tools/one_exec.py --lg $1 --batch $2 --execs 2 --set "streams=1" > gpurun_out/pmc_q.log 2>&1 && python3 tools/pmc_summary.py gpurun_out/pmc_q | grep -v "k_fill\|copyBuffer\|k_spin"; done; done
