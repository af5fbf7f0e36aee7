#!/usr/bin/env python3
"""Diagnostic: per-dispatch durations of the step kernels from a rocprofv3 --kernel-trace of tools/step_ledger.py,
grouped by the card of the trick the launch plays (dispatch j of the step kernel plays card j mod 4).
    python3 tools/step_durations.py <trace dir> <games>"""
import csv, glob, os, sys, collections
d, n = sys.argv[1], int(sys.argv[2])
period = int(sys.argv[3]) if len(sys.argv) > 3 else 4
per = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            per[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for name, v in sorted(per.items()):
    if not ("k_step" in name or "k_play" in name or "k_policy" in name):
        continue
    v = [x for _, x in sorted(v)][2 * period:]
    by = [[x for j, x in enumerate(v) if j % period == c] for c in range(period)]
    med = lambda xs: sorted(xs)[len(xs) // 2] / 1e3 if xs else float("nan")
    print("%-40s n=%d  median us by card of the trick: %s   all: %.2f us = %.1f ps per game" %
          (name, len(v), "  ".join("%.2f" % med(b) for b in by), med(v), med(v) * 1e6 / n))
