#!/bin/bash
# SQ instruction / cycle counters of the headline kernel (GPU box).  usage: bash tools/sq_counters.sh <games> <tag>
N=${1:-65536}; TAG=${2:-sq}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--games $N --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/p1.json 2> $OUT/p1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > $OUT/p2.json 2> $OUT/p2.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_VMEM SQ_WAVES_EQ_64 --output-format csv -d $OUT/p3 -- python3 bench.py $ARGS > $OUT/p3.json 2> $OUT/p3.err || echo "(third counter pass not available)"
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
import bench
res = {}
for p in ("p1", "p2", "p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace(" ", "")
            if "k_play_wide" in k:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 2:]          # steady state: second half of the launches
            res[k] = {"launches": len(v), "mean": sum(v) / len(v)}
res["games"] = $N; res["cards_per_launch"] = json.load(open("$OUT/p1.json"))["config"]["cards_per_launch"]; res["kernel_src_sha"] = bench.kernel_src_sha()
res["source"] = "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> -- python3 bench.py $ARGS; means over the second half of the k_play_wide launches"
json.dump(res, open("$OUT/sq_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*kernel_trace.csv" -delete
