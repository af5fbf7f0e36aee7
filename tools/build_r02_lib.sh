#!/bin/bash
# tools/ab/r02.so: the library as round 2 ended (commit dbe64d9's sources), for same-box before/after runs.
set -e
cd "$(dirname "$0")/.."
REV=${1:-dbe64d9}
T=$(mktemp -d)
mkdir -p $T/tarok_amd/csrc $T/include tools/ab
for f in tarok_amd/csrc/tarok_env.hip tarok_amd/csrc/tarok_device.h tarok_amd/csrc/deal_network.inc include/tarok_env.h; do git show $REV:$f > $T/$f; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I $T/include -o tools/ab/r02.so $T/tarok_amd/csrc/tarok_env.hip
rm -rf $T
ls -la tools/ab/r02.so
