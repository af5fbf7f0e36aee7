#!/bin/bash
# SQ counters of the play role alone (GPU box): launches whose refill workgroups have nothing to do
# (tools/play_only.py).  usage: bash tools/play_only_counters.sh <games> <tag>
N=${1:-65536}; TAG=${2:-play_only}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- python3 tools/play_only.py $N > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/play_only.py $N > $OUT/p2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/p3 -- python3 tools/play_only.py $N > $OUT/p3.log 2>&1 || echo "(third pass not available)"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_INSTS_SENDMSG SQ_ITEMS SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p4 -- python3 tools/play_only.py $N > $OUT/p4.log 2>&1 || echo "(fourth pass not available)"
python3 - <<PY
import csv, glob, collections, json
res = {}
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace(" ", "")
            if "k_play_wide" in k:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 2:]
            res[k] = {"launches": len(v), "mean": sum(v) / len(v)}
import sys
sys.path.insert(0, ".")
import bench
res["games"] = $N; res["cards_per_launch"] = 128; res["kernel_src_sha"] = bench.kernel_src_sha()
res["source"] = "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> -- python3 tools/play_only.py $N; means over the second half of the k_play_wide launches (each follows a reset: no refill work)"
json.dump(res, open("$OUT/play_only_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*kernel_trace.csv" -delete
