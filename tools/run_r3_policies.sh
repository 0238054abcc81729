#!/bin/bash
# cache-policy A/B of the shipped 2^20 pipeline (round 3 re-check of round 1's choice: user buffer nt, ring stores sc1, ring
# loads default): one laboratory library per variant (tools/variants/lab_*.so, built with -DFWA_*_AUX=...), each timed at C3
# in its own process right after the shipped policies in the same process order.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3pol; mkdir -p $O; S=$O/sweep_c3_cache_policies.jsonl; : > $S
echo '{"variant": "shipped (user nt, ring st sc1, ring ld default)"}' >> $S
timeout -k 10 200 python3 tools/sweep.py --lab --lg 20 --batch 4096 --reps 5 --set "" >> $S 2>&1
for v in nt_ring_st plain_ring_st sc1_ring_ld nt_ring_ld default_user_ld plain_user_st sc1_user_st; do
  echo "{\"variant\": \"$v\"}" >> $S
  FWA_LAB_LIBRARY=$GRAFT_REPO_ROOT/tools/variants/lab_$v.so timeout -k 10 200 python3 tools/sweep.py --lab --lg 20 --batch 4096 --reps 5 --set "" >> $S 2>&1 || exit 1
done
echo '{"variant": "shipped again"}' >> $S
timeout -k 10 200 python3 tools/sweep.py --lab --lg 20 --batch 4096 --reps 5 --set "" >> $S 2>&1
echo rc=$?
