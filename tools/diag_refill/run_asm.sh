#!/bin/bash
# harness_run.py over the assembly-patched libraries (tools/diag_refill/asm_variants.py); output gpurun_out/asm_matrix.txt
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/asm_matrix.txt
: > $OUT
for v in "$@"; do
  echo "== $v" >> $OUT
  TAROK_LIB=tools/ab/$v.so timeout -k 10 200 python tools/diag_refill/harness_run.py --reps 10 --combos "0,4,0;0,14,0" 2>&1 | grep -v amdgpu.ids | cut -c1-400 | grep -v "by lane" >> $OUT || echo "exit $?" >> $OUT
done
cat $OUT
