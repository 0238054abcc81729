#!/bin/bash
# round 4, job B: k_p1_pitch (profiles/round4/p1_pitch_negative.patch applied) -- parity, then isolated kernel traces and the sweep rows at 2^21 / 2^22 (32 GiB footprint)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pitch_kernel or last_pass_rows32 or first_pass_1024" > $O/pytest_pitch.log 2>&1 || { tail -60 $O/pytest_pitch.log; exit 1; }
tail -3 $O/pytest_pitch.log
for lg in 21 22; do
  b=$((1 << (32 - lg)))
  timeout -k 10 300 python3 tools/sweep.py --lg $lg --batch $b --reps 7 --set "" --set "p1_pitch=0" --set "streams=1" --set "p1_pitch=0,streams=1" --set "xcd_swizzle=1" >> $O/sweep_pitch.jsonl 2>> $O/err.log
done
cat $O/sweep_pitch.jsonl
for lg in 21 22; do
  b=$((1 << (32 - lg)))
  for s in "streams=1" "p1_pitch=0,streams=1"; do
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/one_exec.py --lg $lg --batch $b --execs 3 --set "$s" > $O/tr.log 2>&1
    echo "== 2^$lg $s" >> $O/trace_isolated_passes_pitch.txt
    python3 tools/trace_summary.py $O/tr >> $O/trace_isolated_passes_pitch.txt
    rm -rf $O/tr
  done
done
cat $O/trace_isolated_passes_pitch.txt
echo done
