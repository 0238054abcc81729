#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/round4
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/round4/pytest_gpu_full.log 2>&1 || { tail -80 gpurun_out/round4/pytest_gpu_full.log; exit 1; }
tail -25 gpurun_out/round4/pytest_gpu_full.log
python3 -c "import __graft_entry__ as g; g.smoke()"
