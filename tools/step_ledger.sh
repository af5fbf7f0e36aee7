#!/bin/bash
# The step API's HBM traffic, array by array (GPU box): rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
# processes, the program directly after `--`) over tools/step_ledger.py with one output / load class switched off at
# a time, and over the round-2 library (tools/ab/r02.so, built by tools/build_r02_lib.sh) for the before/after.
#   usage: bash tools/step_ledger.sh [games=4194304] [tag=ledger]     -> gpurun_out/<tag>/step_ledger.txt
set -o pipefail
N=${1:-4194304}
TAG=${2:-ledger}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name lib mode - done reward
  for C in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/$1_$C
    if [ -n "$2" ]; then export TAROK_LIB=$2; else unset TAROK_LIB; fi
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $d -- python3 tools/step_ledger.py $N $3 $4 $5 $6 96 > $OUT/$1_$C.log 2>&1 || { echo "$1 $C FAILED"; tail -5 $OUT/$1_$C.log; exit 1; }
  done
  echo "$1 done"
}
run two_base "" two d 0 1
run two_done "" two d 1 1
run two_noreward "" two d 1 0
run random_base "" random d 0 1
if [ -f tools/ab/r02.so ]; then
  run two_r02 tools/ab/r02.so two d 1 1
  run random_r02 tools/ab/r02.so random d 1 1
fi
unset TAROK_LIB
python3 tools/step_ledger_summary.py $OUT $N $OUT/step_ledger.json | tee $OUT/step_ledger.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
