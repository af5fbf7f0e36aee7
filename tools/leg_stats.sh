#!/bin/bash
# Per-leg rocprofv3 kernel statistics of the step-API legs at 65,536 games (GPU box): one process per leg.
#   usage: bash tools/leg_stats.sh <tag>    -> gpurun_out/<tag>/leg_<name>_kernel_stats.csv + leg_times.txt
TAG=${1:-legs}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
: > $OUT/leg_times.txt
for leg in "two:0" "one_card:1" "one_trick:4"; do
  name=${leg%%:*}; cards=${leg##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg_$name -- python3 tools/leg_run.py 65536 $cards 7680 2> $OUT/leg_$name.err | grep games >> $OUT/leg_times.txt
  find $OUT/leg_$name -name "*kernel_stats.csv" -exec cp {} $OUT/leg_${name}_kernel_stats.csv \;
  rm -rf $OUT/leg_$name
done
# the same legs without the profiler (its per-dispatch overhead stretches short launches)
for leg in "two:0" "one_card:1" "one_trick:4"; do
  python3 tools/leg_run.py 65536 ${leg##*:} 7680 2>/dev/null | grep games | sed 's/^/un-profiled: /' >> $OUT/leg_times.txt
done
cat $OUT/leg_times.txt
