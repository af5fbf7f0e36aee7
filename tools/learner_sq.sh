#!/bin/bash
# SQ counters of the learner's kernels and the rollout kernel (GPU box): two rocprofv3 --pmc passes over tools/update_prof.py,
# summarised per kernel (means per dispatch over the second half of the dispatches).   usage: bash tools/learner_sq.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r04}; OUT=gpurun_out/${TAG}_lsq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -- python3 tools/update_prof.py > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p2 -- python3 tools/update_prof.py > $OUT/p2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/p3 -- python3 tools/update_prof.py > $OUT/p3.log 2>&1 || exit 1
python3 tools/learner_sq_summary.py $OUT > $OUT/learner_sq_counters.txt
cat $OUT/learner_sq_counters.txt
