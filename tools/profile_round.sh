#!/bin/bash
# Round profile pass on the GPU box: bench lines, rocprofv3 kernel stats, PMC passes (separate), N sweep, SQ counters.
# (Second half — step-API ledger, learner profiles, soak parity — in tools/profile_round_extra.sh: one gpurun call each.)
# usage: bash tools/profile_round.sh <tag>      (writes gpurun_out/<tag>/; copy the summaries into profiles/)
set -o pipefail
TAG=${1:-r03}
CARDS=${2:-128}          # the bench's default cards per launch
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
echo "bench done"; cut -c1-300 $OUT/bench_line.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_command.json 2>> $OUT/bench.err || exit 1
echo "driver-command bench done"; cut -c1-300 $OUT/bench_line_driver_command.json
# headline mode only: every k_play dispatch in this trace is a launch of the timed mode
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline --no-extras > $OUT/stats_bench.json 2> $OUT/stats.err || exit 1
# all modes (one trick / one card per launch, two-kernel path, rollout, self-play side measurements)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_all -- python3 bench.py --no-cpu-baseline > $OUT/stats_all_bench.json 2> $OUT/stats_all.err || exit 1
echo "stats done"
# HBM traffic: separate passes per counter; headline only, then with the side legs (the step API's kernels)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_headline_$C -- python3 bench.py --steps 20 --warmup 4 --repeats 2 --no-cpu-baseline --no-extras > $OUT/pmc_headline_$C.json 2> $OUT/pmc_headline_$C.err || exit 1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_all_$C -- python3 bench.py --steps 20 --warmup 4 --repeats 2 --no-cpu-baseline > $OUT/pmc_all_$C.json 2> $OUT/pmc_all_$C.err || exit 1
done
echo "pmc done"
python3 tools/pmc_summary.py $OUT/pmc_headline_FETCH_SIZE $OUT/pmc_headline_WRITE_SIZE $CARDS 65536 $OUT/pmc_fetch_write_65536.json
python3 tools/pmc_summary.py $OUT/pmc_all_FETCH_SIZE $OUT/pmc_all_WRITE_SIZE $CARDS 65536 $OUT/pmc_fetch_write_all_modes_65536.json
for N in 65536 1048576 4194304 16777216; do
  python3 bench.py --games $N --steps 8 --warmup 4 --no-cpu-baseline --no-extras > $OUT/bench_N$N.json 2>> $OUT/nsweep.err || exit 1
  cut -c1-200 $OUT/bench_N$N.json
done
bash tools/sq_counters.sh 65536 $TAG/sq64k > /dev/null 2>&1 && cp gpurun_out/$TAG/sq64k/sq_counters.json $OUT/sq_counters.json
bash tools/sq_counters.sh 4194304 $TAG/sq4m > /dev/null 2>&1 && cp gpurun_out/$TAG/sq4m/sq_counters.json $OUT/sq_counters_4194304.json
bash tools/play_only_counters.sh 65536 $TAG/play_only > /dev/null 2>&1 && cp gpurun_out/$TAG/play_only/play_only_counters.json $OUT/play_only_counters.json
timeout -k 10 600 ./tools/valu_issue 1500 > $OUT/valu_issue_raw.json 2> $OUT/valu_issue.err
python3 tools/first_launches.py 65536 $CARDS > $OUT/first_launches.txt 2>&1
python3 tools/krog_stamps.py 65536 $CARDS > $OUT/wave_stamps_65536.txt 2>&1
python3 tools/card_probe.py 65536 $CARDS > $OUT/card_probe_65536.txt 2>&1
python3 tools/mlp_time.py 65536 > $OUT/policy_mlp_times.txt 2>&1
python3 tools/observe_ref_time.py > $OUT/observe_ref_times.txt 2>&1
# keep only the summaries of the rocprof directories (the raw traces are large)
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
find $OUT/stats_all -name "*kernel_stats.csv" -exec cp {} $OUT/bench_all_modes_kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
bash tools/leg_stats.sh $TAG/legs > /dev/null 2>&1; cp gpurun_out/$TAG/legs/leg_*_kernel_stats.csv gpurun_out/$TAG/legs/leg_times.txt $OUT/ 2>/dev/null
echo ALL DONE
