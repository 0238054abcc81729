#!/bin/bash
# cache-policy A/B of the shipped 2^20 pipeline (round 3 re-check of round 1's choice: user buffer nt, ring stores sc1, ring
# loads default): one laboratory library per variant (tools/variants/lab_*.so, built with -DFWA_*_AUX=...), each timed at C3
# in its own process right after the shipped policies in the same process order.
# Build the variants first (in the container, from fft_wgpu_amd/csrc after `make all lab`), e.g. for nt_ring_st:
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DFWA_LAB -DFWA_RING_ST_AUX=AUX_NT -c kernels_1m.hip -o /tmp/k1m_v.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/lab_nt_ring_st.so <every object of LAB_OBJS with kernels_1m.lab.o replaced by /tmp/k1m_v.o>
# (plain_ring_st: FWA_RING_ST_AUX=AUX_DEFAULT; sc1_ring_ld / nt_ring_ld: FWA_RING_LD_AUX; default_user_ld: FWA_USER_LD_AUX=AUX_DEFAULT;
#  plain_user_st / sc1_user_st: FWA_USER_ST_AUX)
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
