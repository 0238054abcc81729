#!/bin/bash
# tools/run_sweep.sh OUT LG BATCH REPS [--lab] SETTING...  -- one interleaved A/B of plan settings through tools/sweep.py,
# the form every one-off run_rN_*.sh of rounds 2-4 had (those are kept under tools/archive/ for the record of what produced
# which file under profiles/).  Example (the round-3 group / chain sweep at C3):
#   tools/run_sweep.sh gpurun_out/sweep.jsonl 20 4096 5 "" "group=8" "group=32" "streams=1" "streams=4"
set -e
out=$1; lg=$2; batch=$3; reps=$4; shift 4
lab=""; if [ "$1" = "--lab" ]; then lab="--lab"; shift; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$(dirname "$out")"
args=()
for s in "$@"; do args+=(--set "$s"); done
timeout -k 10 900 python3 tools/sweep.py $lab --lg "$lg" --batch "$batch" --reps "$reps" "${args[@]}" > "$out" 2>&1
tail -n 20 "$out"
