#!/bin/bash
# tools/bisect_pipe_slow.sh -- the slow mode of the host pipeline (2070 iterations/s = H2D and D2H copies serialised): which
# PREDECESSOR process puts the next process into it, and does it go away with shader copies (HSA_ENABLE_SDMA=0)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 100 python3 tools/reference_loop.py --iters 300 2>&1 | grep "2 slots" | sed 's/.*iters_per_s": \([0-9.]*\).*/\1/'; }
pred() { timeout -k 10 100 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --spread 0 > /dev/null 2>&1; }
echo "fresh box: $(run)"
pred; echo "after bench.py: $(run)"
pred; echo "after bench.py, HSA_ENABLE_SDMA=0: $(HSA_ENABLE_SDMA=0 run)"
pred; echo "after bench.py, link probe:"; timeout -k 10 100 python3 tools/link_probe.py | head -2
pred; echo "after bench.py: $(run)"
echo "after that run: $(run)"
pred; sleep 3; echo "after bench.py + 3 s pause: $(run)"
pred; sleep 10; echo "after bench.py + 10 s pause: $(run)"
pred; timeout -k 10 100 python3 tools/pipe_long_probe.py 6000 2 250 | cut -c1-400
