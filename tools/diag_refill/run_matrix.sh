#!/bin/bash
# Round-4 diagnostic matrix (GPU box): the failing refill-role form with one change each; see gen_variants.py.
# Every run is its own process; output to gpurun_out/diag_matrix.txt.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/diag_matrix.txt
: > $OUT
run() {   # lib, extra args...
  local lib=$1; shift
  if [ "$lib" = product ]; then
    timeout -k 10 300 python tools/diag_refill/diag_soak.py "$@" >> $OUT 2>&1 || echo "[$lib $*] exit $?" >> $OUT
  else
    TAROK_LIB=tools/ab/diag_$lib.so timeout -k 10 300 python tools/diag_refill/diag_soak.py "$@" >> $OUT 2>&1 || echo "[$lib $*] exit $?" >> $OUT
  fi
  tail -n 3 $OUT
}
run product
run fail0 --repeat 2
run fail0 --eager --tag eager
run fail0 --sync --tag sync
run fail0 --nokrog --tag nokrog
run keep
run verify
run verify_keep
run tick
run norestrict
run fail0 --lazy 0 --target 3000 --tag lazy0
run keep --lazy 0 --target 3000 --tag lazy0
run norestrict --lazy 0 --target 3000 --tag lazy0
echo done >> $OUT
