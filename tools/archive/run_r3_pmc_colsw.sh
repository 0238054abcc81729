cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3pmc2
for spec in "18 1024" "19 512" "20 256"; do set -- $spec; : > gpurun_out/r3pmc2/pmc_2p$1.txt; tools/run_pmc_counters.sh $1 $2 "streams=1" gpurun_out/r3pmc2/pmc_2p$1.txt || exit 1; done
echo rc=$?
