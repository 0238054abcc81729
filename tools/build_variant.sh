#!/bin/bash
# Build the product library with compile-time defaults overridden, for A/B measurements in one process (tools/ab_libs.py):
#   tools/build_variant.sh NAME "-DFWA_PF_TILE=1 -DFWA_PF_ROWS32=4"   ->  build/variants/NAME/libfft_wgpu_amd.so
# (build/ is git-ignored but travels to the GPU box with gpurun).  The sources are copied, so the tree's own objects stay.
set -euo pipefail
name=$1; extra=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/build/variants/$name
mkdir -p "$dst/csrc" "$root/build/variants/include"
cp "$root"/include/*.h "$root/build/variants/include/"       # csrc includes ../../include/fft_wgpu_amd.h
for f in "$root"/fft_wgpu_amd/csrc/*; do
    case "$f" in *.o|*.so) ;; *) cp -p "$f" "$dst/csrc/";; esac
done
make -s -C "$dst/csrc" -j8 all EXTRA="$extra" 2>&1 | grep -E "error|warning" || true
ls -la "$dst/libfft_wgpu_amd.so"
