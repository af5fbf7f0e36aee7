#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of tools/fetch_calib's kernels against the bytes they are known to touch.
    python3 tools/fetch_calib_summary.py <fetch_dir> <write_dir>"""
import csv, glob, os, sys, collections
def collect(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return per
n = 1 << 20
def chosen(i): return ((i * 2654435761) & 0xFFFFFFFF) >> 7
cnt = sum(1 for i in range(n) if chosen(i) % 10 == 0)
known = {"k_stream16": ("R", n * 16), "k_stream8": ("R", n * 8), "k_stream1": ("R", n * 1), "k_sparse64": ("R", cnt * 64),
         "k_sparse32": ("R", cnt * 32), "k_wsparse32": ("W", cnt * 20), "k_wstream16": ("W", n * 16)}
fe, wr = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
print("lanes %d, chosen (one in ten by a hash) %d" % (n, cnt))
print("%-12s %12s %14s %10s" % ("kernel", "bytes known", "counter bytes", "ratio"))
for k, (rw, b) in known.items():
    v = (fe if rw == "R" else wr).get(k, [])
    if not v: continue
    c = sorted(v)[len(v) // 2] * 1024.0
    print("%-12s %12d %14.0f %10.3f   (%s_SIZE as reported, KB x 1024)" % (k, b, c, c / b, "FETCH" if rw == "R" else "WRITE"))
