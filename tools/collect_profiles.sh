#!/bin/bash
# Copy the summaries of a round's two profile passes (gpurun_out/<tag>, gpurun_out/<tag>x: tools/profile_round.sh,
# profile_round_extra.sh) into profiles/<tag>_*.   usage (here, after the gpurun calls): bash tools/collect_profiles.sh r04
set -e
TAG=${1:-r04}; S=gpurun_out/$TAG; X=gpurun_out/${TAG}x; P=profiles
cp $S/bench_kernel_stats.csv $P/${TAG}_bench_kernel_stats.csv
cp $S/bench_all_modes_kernel_stats.csv $P/${TAG}_bench_all_modes_kernel_stats.csv
for l in two one_card one_trick; do cp $S/leg_${l}_kernel_stats.csv $P/${TAG}_leg_${l}_kernel_stats.csv; done
cp $S/leg_times.txt $P/${TAG}_leg_times.txt
cp $S/pmc_fetch_write_65536.json $P/${TAG}_pmc_fetch_write_65536.json
cp $S/pmc_fetch_write_all_modes_65536.json $P/${TAG}_pmc_fetch_write_all_modes_65536.json
cp $S/sq_counters.json $P/${TAG}_sq_counters.json
cp $S/sq_counters_4194304.json $P/${TAG}_sq_counters_4194304.json
cp $S/play_only_counters.json $P/${TAG}_play_only_counters.json
cp $S/valu_issue_raw.json $P/${TAG}_valu_issue_raw.json
python3 tools/valu_issue_summary.py $S/valu_issue_raw.json $P/${TAG}_valu_issue.json > /dev/null
python3 tools/n_sweep_summary.py $S $P/${TAG}_n_sweep.json
for f in first_launches.txt wave_stamps_65536.txt card_probe_65536.txt policy_mlp_times.txt observe_ref_times.txt; do grep -v amdgpu.ids $S/$f > $P/${TAG}_$f; done
grep -v amdgpu.ids $X/step_api_n_sweep.txt > $P/${TAG}_step_api_n_sweep.txt
cp $X/step_durations_final.txt $P/${TAG}_step_durations.txt
cp $X/step_sq.txt $P/${TAG}_step_sq.txt
cp $X/step_ledger.txt $P/${TAG}_step_ledger.txt
cp $X/step_ledger.json $P/${TAG}_step_ledger.json
cp $X/update_kernel_stats.csv $P/${TAG}_update_kernel_stats.csv
grep -v amdgpu.ids $X/update_times.txt > $P/${TAG}_update_times.txt
grep -v amdgpu.ids $X/chain_stamps.txt > $P/${TAG}_chain_stamps.txt
python3 - <<PY
import json, glob
shas = {f: json.load(open(f)).get("kernel_src_sha") for f in glob.glob("$P/${TAG}_*.json") if "bench_line" not in f and "2rank" not in f}
print("kernel_src_sha of the collected JSON files:", sorted(set(str(v) for v in shas.values())))
PY
