#!/bin/bash
# round 4, job A: the new multi-GPU / robustness tests, then the whole GPU suite, then the bench line
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_sharding.py -x -q -m gpu > $O/pytest_sharding.log 2>&1 || { tail -60 $O/pytest_sharding.log; exit 1; }
tail -3 $O/pytest_sharding.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_lab.py -x -q -m gpu -k "failed_launch" > $O/pytest_lab_fail.log 2>&1 || { tail -60 $O/pytest_lab_fail.log; exit 1; }
tail -3 $O/pytest_lab_fail.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cpp_host_pipeline or graph or calib or copy" > $O/pytest_misc.log 2>&1 || { tail -60 $O/pytest_misc.log; exit 1; }
tail -3 $O/pytest_misc.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
cat $O/bench.json
echo done
