# round 5: (group, chains) re-sweep of the tiled two-pass plans 2^16 .. 2^19 with the current kernels
# (odd log2 n at 16 GiB: every Forward plan of an odd size owns a second buffer of the batch's size)
set -e
mkdir -p gpurun_out/r5grp
for lg in 16 17 18 19; do
  per=$(( (1<<lg)*8 ))
  tot=32; [ $((lg % 2)) -eq 1 ] && tot=31
  for s in 2 3 4; do
    args=""
    for mib in 8 16 32 64 128 256; do
      g=$(( (mib<<20)/per ))
      args="$args --set group=$g,streams=$s"
    done
    python tools/sweep.py --lg $lg --batch $((1<<(tot-lg))) --reps 5 --set "" $args >> gpurun_out/r5grp/lg$lg.jsonl 2>> gpurun_out/r5grp/lg$lg.err
  done
  echo done $lg
done
