#!/bin/bash
# round 4 experiment: the two chain streams confined to complementary halves of the CUs (FWA_CHAIN_CU_SPLIT = 1: mask bits
# [0,128) / [128,256); 2: even / odd bits; 3: nibbles), with and without the pair maps (adjacent tiles on the two residents of a CU)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4split; mkdir -p $O; rm -f $O/ab.txt
V=$PWD/tools/variants
run() { label=$1; shift; env "$@" timeout -k 10 200 python3 tools/sweep.py --lab --lg 20 --batch 4096 --reps 5 --set "" | sed "s/^/$(printf '%-14.14s' $label)/" >> $O/ab.txt; }
for rep in 1 2; do
  run shipped FWA_LAB_LIBRARY=$PWD/fft_wgpu_amd/libfft_wgpu_amd.so
  run split1 FWA_LAB_LIBRARY=$PWD/fft_wgpu_amd/libfft_wgpu_amd.so FWA_CHAIN_CU_SPLIT=1
  run split1+pair16 FWA_LAB_LIBRARY=$V/pair16.so FWA_CHAIN_CU_SPLIT=1
  run split2 FWA_LAB_LIBRARY=$PWD/fft_wgpu_amd/libfft_wgpu_amd.so FWA_CHAIN_CU_SPLIT=2
  run split2+pair1 FWA_LAB_LIBRARY=$V/pair1.so FWA_CHAIN_CU_SPLIT=2
  run split3 FWA_LAB_LIBRARY=$PWD/fft_wgpu_amd/libfft_wgpu_amd.so FWA_CHAIN_CU_SPLIT=3
  run split3+pair16 FWA_LAB_LIBRARY=$V/pair16.so FWA_CHAIN_CU_SPLIT=3
done
python3 - <<PY
import json
for l in open("$O/ab.txt"):
    d = json.loads(l[14:]); print(l[:14], "%-10s" % d["setting"], d["ms"], d["ms_min"], d["roofline_frac"])
PY
