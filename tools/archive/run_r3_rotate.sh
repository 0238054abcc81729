#!/bin/bash
# what does the Infinity Cache give the ring?  Same launches, ring footprint x ring_rotate (laboratory knob)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3rotate; mkdir -p $O
timeout -k 10 400 python3 tools/sweep.py --lab --lg 20 --batch 4096 --reps 5 --set "" --set "ring_rotate=2" --set "ring_rotate=4" --set "ring_rotate=16" \
  --set "group=8" --set "group=8,ring_rotate=2" --set "group=8,ring_rotate=8" \
  --set "group=4" --set "group=4,ring_rotate=4" --set "group=4,ring_rotate=16" \
  --set "group=4,streams=4" --set "group=4,streams=4,ring_rotate=2" --set "group=4,streams=4,ring_rotate=8" \
  --set "group=2,streams=4" --set "group=2,streams=4,ring_rotate=16" > $O/sweep_ring_rotate.jsonl 2>&1
echo rc=$?
