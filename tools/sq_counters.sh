#!/bin/bash
# SQ instruction / cycle counters of the headline kernel (GPU box).  usage: bash tools/sq_counters.sh <games> <tag>
N=${1:-4194304}; TAG=${2:-sq}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- python3 bench.py --games $N --steps 96 --warmup 48 --repeats 1 --no-cpu-baseline --no-extras > $OUT/p1.json 2> $OUT/p1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 bench.py --games $N --steps 96 --warmup 48 --repeats 1 --no-cpu-baseline --no-extras > $OUT/p2.json 2> $OUT/p2.err || exit 1
python3 - <<PY
import csv, glob, collections, json
res = {}
for p in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_play<true>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 2:]          # steady state: second half of the launches
            res[k] = {"launches": len(v), "mean": sum(v) / len(v)}
res["games"] = $N; res["cards_per_launch"] = 48
json.dump(res, open("$OUT/sq_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*kernel_trace.csv" -delete
