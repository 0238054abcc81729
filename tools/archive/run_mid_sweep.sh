set -e
for lg in 15 16 17 18 19; do
  b=$((1 << (28 - lg)))
  per=$((1 << (lg + 3)))
  sets=""
  for mib in 8 16 32 64 128; do g=$(( (mib << 20) / per )); [ $g -ge 1 ] && sets="$sets --set group=$g"; done
  timeout -k 10 120 python tools/sweep.py --lg $lg --batch $b --reps 5 $sets
done
timeout -k 10 120 python tools/sweep.py --lg 18 --batch 1024 --reps 5 --set "factors=10.8,group=32" --set "factors=10.8,group=16" --set "factors=10.8,group=8" --set "factors=9.9,group=16" --set "factors=9.9,group=32"
timeout -k 10 120 python tools/sweep.py --lg 19 --batch 512 --reps 5 --set "factors=10.9,group=16" --set "factors=10.9,group=8" --set "factors=10.9,group=4" --set "factors=9.10,group=8"
