set -e
for lg in 16 17 18 19 21 22 24; do
  b=$((1 << (28 - lg)))
  timeout -k 10 120 python tools/sweep.py --lg $lg --batch $b --reps 7 --set "xcd_swizzle=1" --set "xcd_swizzle=0"
done
timeout -k 10 120 python tools/sweep.py --lg 24 --batch 1 --reps 9 --set "xcd_swizzle=1" --set "xcd_swizzle=0"
timeout -k 10 120 python tools/sweep.py --lg 20 --batch 1 --reps 9 --set "xcd_swizzle=1" --set "xcd_swizzle=0" --set "factors=7.7.6" --set "factors=8.6.6" --set "factors=6.6.8"
