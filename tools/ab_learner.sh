#!/bin/bash
# Same-box A/B of learner libraries (GPU box): un-profiled update / rollout wall times of tools/update_prof.py and the
# per-kernel averages of one rocprofv3 pass, per library under tools/ab/.   usage: bash tools/ab_learner.sh lib1.so lib2.so ...
export TMPDIR=/tmp
mkdir -p gpurun_out/abl
for lib in "$@"; do
  echo "== $lib"
  for rep in 1 2; do TAROK_LIB=tools/ab/$lib python3 tools/update_prof.py 2>/dev/null | tail -n 2 | tr '\n' ' '; echo; done
  rm -rf gpurun_out/abl/$lib
  TAROK_LIB=tools/ab/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl/$lib -- python3 tools/update_prof.py > /dev/null 2>&1
  f=$(find gpurun_out/abl/$lib -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("   " + " | ".join("%s %.1f" % (r["Name"].split("(")[0][:16], float(r["AverageNs"]) / 1e3) for r in rows if r["Name"].startswith(("k_learn", "_Z13k_policy_step", "k_returns"))))
PY
  rm -rf gpurun_out/abl/$lib
done
