#!/bin/bash
# per-kernel durations (isolated: streams=1) of the two-pass plans with k_colsw against the defaults
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3trace; mkdir -p $O; : > $O/summary.txt
run() {  # lg batch setting
  rm -rf $O/t
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/one_exec.py --lg $1 --batch $2 --execs 3 --set "$3" > $O/t.log 2>&1 || { echo "FAILED $1 $3" >> $O/summary.txt; return 1; }
  echo "== 2^$1 x $2  $3" >> $O/summary.txt
  python3 tools/trace_summary.py $O/t | grep -v "k_fill\|copyBuffer" >> $O/summary.txt
}
run 20 1024 "streams=1" && run 20 1024 "factors=9.11,colsw=1,streams=1" && run 20 1024 "factors=9.11,colsw=1,tile_ring=0,streams=1" && run 20 1024 "factors=8.12,colsw=1,streams=1" && \
run 19 2048 "streams=1" && run 19 2048 "factors=9.10,colsw=1,streams=1" && run 19 2048 "factors=9.10,colsw=1,tile_ring=0,streams=1" && \
run 18 4096 "streams=1" && run 18 4096 "factors=8.10,colsw=1,streams=1" && run 18 4096 "factors=9.9,colsw=1,streams=1" && \
run 21 512 "streams=1" && run 22 256 "streams=1" && run 23 128 "streams=1" && run 16 16384 "streams=1" && run 16 16384 "factors=8.8,colsw=1,streams=1"
rm -rf $O/t
cat $O/summary.txt
