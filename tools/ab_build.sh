#!/bin/bash
# Diagnostic builds for same-box A/B runs (tools/ab_cards.py reads tools/ab/*.so):
#   a_block256.so   the product build
#   b_block512.so   512-slot play workgroups: two play waves per SIMD on half the CUs at 65,536 games
#   c_block1024.so  1024-slot play workgroups: four play waves per SIMD on a quarter of the CUs
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab
for v in "a_block256:256" "b_block512:512" "c_block1024:1024"; do
  name=${v%%:*}; blk=${v##*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -I include -DTK_BLOCK=$blk -o tools/ab/$name.so tarok_amd/csrc/tarok_env.hip
done
ls -la tools/ab
