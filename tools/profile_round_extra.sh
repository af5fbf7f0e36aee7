#!/bin/bash
# Second half of a round's profile pass (GPU box): the step API (N sweep, per-card durations, traffic ledger at 4 M
# games), the fused learner (per-kernel times of an update, phase stamps of k_learn_chain), soak parity of the
# final sources.   usage: bash tools/profile_round_extra.sh <tag>      (writes gpurun_out/<tag>x/)
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/${TAG}x
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/step_api_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/step_api_n_sweep.txt; echo "sweep done"
for m in random two; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/dur_$m -- python3 tools/step_ledger.py 4194304 $m d 0 1 96 > $OUT/dur_$m.log 2>&1
  python3 tools/step_durations.py $OUT/dur_$m 4194304 >> $OUT/step_durations_final.txt
  rm -rf $OUT/dur_$m
done
# 65,536 games: the one-card step deals in bulk every sixteenth launch there (period 16: launch number mod 16, 0 = the bulk launch)
rocprofv3 --kernel-trace --output-format csv -d $OUT/dur_small -- python3 tools/step_ledger.py 65536 two d 0 1 224 > $OUT/dur_small.log 2>&1
echo "65,536 games, by launch number mod 32:" >> $OUT/step_durations_final.txt
python3 tools/step_durations.py $OUT/dur_small 65536 32 >> $OUT/step_durations_final.txt
rm -rf $OUT/dur_small
echo "durations done"
for N in 4194304 65536; do bash tools/step_sq.sh $N two ${TAG}x/sq > /dev/null 2>&1; done
cat $OUT/sq/step_sq_4194304_two.txt $OUT/sq/step_sq_65536_two.txt > $OUT/step_sq.txt 2>/dev/null; echo "sq done"
timeout -k 10 400 bash tools/step_ledger.sh 4194304 ${TAG}x/ledger > $OUT/ledger.log 2>&1; cp $OUT/ledger/step_ledger.txt $OUT/step_ledger.txt; cp $OUT/ledger/step_ledger.json $OUT/step_ledger.json; echo "ledger done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/upd -- python3 tools/update_prof.py > $OUT/update_prof.txt 2>&1
find $OUT/upd -name "*kernel_stats.csv" -exec cp {} $OUT/update_kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -delete
python3 tools/update_prof.py 2>&1 | grep -v amdgpu.ids > $OUT/update_times.txt
python3 tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/chain_stamps.txt; echo "learner done"
timeout -k 10 500 python3 tools/soak_parity.py > $OUT/soak_parity.txt 2>&1; echo "soak rc $?"
timeout -k 10 300 python3 tools/soak_paths.py >> $OUT/soak_parity.txt 2>&1; echo "paths rc $?"
timeout -k 10 400 python3 tools/soak_mixed.py 2>&1 | grep -v amdgpu.ids >> $OUT/soak_parity.txt; echo "mixed rc $?"
echo ALL DONE
