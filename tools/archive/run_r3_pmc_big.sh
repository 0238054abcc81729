#!/bin/bash
# counter summaries for the two-pass sizes left below 0.38 of the roofline at the 32-GiB footprint
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3pmc; mkdir -p $O
for spec in "21 256" "22 128" "23 64" "24 32"; do set -- $spec; : > $O/pmc_2p$1.txt; tools/run_pmc_counters.sh $1 $2 "streams=1" $O/pmc_2p$1.txt || exit 1; done
echo rc=$?
