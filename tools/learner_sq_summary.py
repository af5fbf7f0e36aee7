#!/usr/bin/env python3
"""Summary of tools/learner_sq.sh: per kernel, the mean per dispatch of every counter over the second half of its dispatches."""
import csv, glob, sys, collections
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "k_policy_step" if "k_policy_step" in k else k.split("(")[0]
        if name.startswith(("k_learn_chain", "k_learn_dw", "k_policy_step")):
            vals[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for name in ("k_policy_step", "k_learn_chain", "k_learn_dw"):
    print(name)
    m = {}
    for c, lst in sorted(vals[name].items()):
        lst.sort()
        half = [v for _, v in lst[len(lst) // 2:]]
        m[c] = sum(half) / max(1, len(half))
    print("   " + " | ".join("%s %.4g" % (c, v) for c, v in sorted(m.items())))
    if "SQ_WAVE_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"] * 4
        print("   wave cycles (x4) %.4g | MFMA busy cycles / wave cycles %.3f | waiting %.3f | waiting at issue %.3f | VALU instructions per MFMA %.1f"
              % (wc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / wc, m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
                 m.get("SQ_INSTS_VALU", 0) / max(1.0, m.get("SQ_INSTS_MFMA", 1))))
