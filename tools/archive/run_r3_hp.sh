cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
g++ -O2 -std=c++17 -Iinclude tools/example_basic_pipeline.cpp -Lfft_wgpu_amd -lfft_wgpu_amd -o /tmp/ebp && export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/fft_wgpu_amd:$LD_LIBRARY_PATH
for s in 2 3 4 6; do /tmp/ebp 40 $s | tail -1; done
timeout 200 python3 tools/reference_loop.py --iters 500 | cut -c1-200
for s in 3 3; do /tmp/ebp 40 $s | tail -1; done
