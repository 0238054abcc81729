#!/bin/bash
# tools/run_round_profiles.sh -- the measurement set behind profiles/roundN/a_* and b_* (run on the GPU box through gpurun):
# bench line, rocprofv3 kernel stats (default and single-chain), PMC fetch / write passes, size sweeps at both
# footprints, C3 variants (product and laboratory), the four plan kinds, host link / pipeline, copy tuning.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
mkdir -p $O
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_default.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_streams1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > $O/prof_streams1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch_summary.txt
python3 tools/pmc_summary.py $O/pmc_write > $O/pmc_write_summary.txt
python3 tools/trace_summary.py $O/prof_default > $O/trace_default_summary.txt
python3 tools/trace_summary.py $O/prof_streams1 > $O/trace_streams1_summary.txt
find $O -name "*kernel_stats.csv" | head -4 > $O/stats_files.txt
cp "$(find $O/prof_default -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_default.csv
cp "$(find $O/prof_streams1 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_streams1_isolated.csv
rm -rf $O/prof_default $O/prof_streams1 $O/pmc_fetch $O/pmc_write
timeout -k 10 400 python3 tools/size_bench.py --lg-max 30 > $O/size_sweep_2GiB.jsonl 2>&1
timeout -k 10 400 python3 tools/size_bench.py --lg-min 9 --lg-max 24 --total-lg 32 --no-latency-shapes > $O/size_sweep_32GiB.jsonl 2>&1
timeout -k 10 300 python3 tools/sweep.py --lg 20 --batch 4096 --reps 7 --set "" --set "xcd_swizzle=0" --set "streams=1" --set "group=8" --set "group=32" --set "factors=9.11,colsw=1" --set "factors=9.11,colsw=1,tile_ring=0" --set "factors=8.12,colsw=1" > $O/c3_variants.jsonl 2>&1
timeout -k 10 300 python3 tools/sweep.py --lab --lg 20 --batch 4096 --reps 5 --set "" --set "tile_w=32" --set "path=5" --set "path=5,depth=4,ring_slots=8" > $O/c3_variants_lab.jsonl 2>&1
timeout -k 10 200 python3 tools/kinds_bench.py > $O/kinds_bench.jsonl 2>&1
timeout -k 10 200 python3 tools/reference_loop.py --iters 1000 > $O/reference_loop.jsonl 2>&1
timeout -k 10 100 python3 tools/link_probe.py > $O/host_link.jsonl 2>&1
echo done
