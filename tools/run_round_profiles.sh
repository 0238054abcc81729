#!/bin/bash
# tools/run_round_profiles.sh ROUND [profiles-only] -- the measurement set behind profiles/roundN/a_* and b_* (run on the GPU box through
# gpurun; results land in gpurun_out/roundN/, copy what is to be judged into profiles/roundN/).
# Order: PMC passes and kernel statistics FIRST, then profiles/bench_reference.json is refreshed
# from them, and only then the bench line is taken -- so that the line and the summaries of one job agree.
set -e
R=${1:?round number}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round$R
mkdir -p $O
# the profiled runs leave out the legs that launch the calibration kernels at other sizes (spread, batch-1 configurations): the
# per-dispatch means of k_copy / k_scale / k_small32<10> then are the 8 / 16 / 32-GiB passes the FETCH x 2 rule is checked on
B="python3 bench.py --no-cpu-baseline --spread 0 --config-execs 0"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- $B --steps 5 --warmup 2 > $O/prof_default.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_streams1 -- $B --steps 5 --warmup 2 --streams 1 > $O/prof_streams1.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch > $O/a_pmc_fetch_summary.txt
python3 tools/pmc_summary.py $O/pmc_write > $O/a_pmc_write_summary.txt
python3 tools/trace_summary.py $O/prof_default > $O/a_trace_default_summary.txt
python3 tools/trace_summary.py $O/prof_streams1 > $O/a_trace_streams1_summary.txt
cp "$(find $O/prof_default -name '*kernel_stats.csv' | head -1)" $O/a_kernel_stats_default.csv
cp "$(find $O/prof_streams1 -name '*kernel_stats.csv' | head -1)" $O/a_kernel_stats_streams1_isolated.csv
rm -rf $O/prof_default $O/prof_streams1 $O/pmc_fetch $O/pmc_write
python3 tools/make_bench_reference.py $O $R "${FWA_COMMIT:-}" > $O/bench_reference.log 2>&1   # FWA_COMMIT=$(git rev-parse --short=12 HEAD) in the gpurun command line: the box has no .git
cp profiles/bench_reference.json $O/bench_reference.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/a_bench_default.json 2> $O/bench_default.err
echo "profiles done"
if [ "${2:-}" = profiles-only ]; then exit 0; fi
timeout -k 10 400 python3 tools/size_bench.py --lg-max 24 > $O/b_size_sweep_2GiB.jsonl 2>&1
timeout -k 10 400 python3 tools/size_bench.py --lg-min 1 --lg-max 24 --total-lg 32 --no-latency-shapes > $O/b_size_sweep_32GiB.jsonl 2>&1
echo "sweeps done"
timeout -k 10 200 python3 tools/latency_shapes.py --label head > $O/latency_shapes.jsonl 2>&1
timeout -k 10 200 python3 tools/kinds_bench.py > $O/kinds_bench.jsonl 2>&1
sleep 3   # a process started right after one that freed tens of GiB runs its host copies serialised for ~0.5 s: profiles/round4/probe_pipe_slow_after_large_free.txt
timeout -k 10 200 python3 tools/reference_loop.py --iters 1000 > $O/d_reference_loop_pcie_inclusive.jsonl 2>&1
timeout -k 10 100 python3 tools/link_probe.py > $O/host_link.jsonl 2>&1
g++ -O2 -std=c++17 -Iinclude tools/example_basic_pipeline.cpp -Lfft_wgpu_amd -lfft_wgpu_amd -pthread -o /tmp/example_basic_pipeline
LD_LIBRARY_PATH=fft_wgpu_amd timeout -k 10 200 /tmp/example_basic_pipeline 300 3 > $O/host_pipeline_cpp.txt 2>&1
echo done
