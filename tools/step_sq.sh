#!/bin/bash
# SQ instruction / cycle counters of the one-card step kernel by the card of the trick (GPU box).
#   usage: bash tools/step_sq.sh <games> <mode: two|random> <tag>      -> gpurun_out/<tag>/step_sq_<games>_<mode>.txt
N=${1:-4194304}; M=${2:-two}; TAG=${3:-stepsq}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/q1_${N}_$M -- python3 tools/step_ledger.py $N $M d 0 1 96 > $OUT/q1.log 2>&1 || { tail -5 $OUT/q1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/q2_${N}_$M -- python3 tools/step_ledger.py $N $M d 0 1 96 > $OUT/q2.log 2>&1 || { tail -5 $OUT/q2.log; exit 1; }
python3 - <<PY > $OUT/step_sq_${N}_$M.txt
import csv, glob, collections
n = $N
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("q1", "q2"):
    for f in glob.glob("$OUT/%s_${N}_$M/**/*counter_collection.csv" % p, recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = r["Kernel_Name"].split("(")[0]
            if "k_step" in k or "k_policy" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("SQ counters per launch by the card of the trick, %d games, mode $M (median over the tricks after the second; per game where it says so)" % n)
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        v = v[8:]
        by = [sorted(v[j::4]) for j in range(4)]
        med = [b[len(b) // 2] if b else float("nan") for b in by]
        per_game = c.startswith("SQ_INSTS") 
        scale = (64.0 / n) if per_game else 1.0
        print("  %-22s %s%s" % (c, "  ".join("%12.1f" % (m * scale) for m in med), "   per game (x64 / games)" if per_game else ""))
PY
cat $OUT/step_sq_${N}_$M.txt
find $OUT -name "*kernel_trace.csv" -delete
