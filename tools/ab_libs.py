#!/usr/bin/env python3
"""Diagnostic (GPU box): same-box A/B of libraries under tools/ab/ (each built by hand with its -D flags):
bench.py --no-extras --no-cpu-baseline per library and batch size, medians of the repeats.

    python tools/ab_libs.py base.so,prio3.so 65536,1048576 [steps]"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
libs = sys.argv[1].split(",")
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [65536]
steps = sys.argv[3] if len(sys.argv) > 3 else "40"
for n in sizes:
    row = []
    for lib in libs:
        env = dict(os.environ, TAROK_LIB=os.path.join(HERE, "ab", lib))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline", "--games", str(n), "--steps", steps],
                             env=env, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print(lib, n, "FAILED", out.stderr[-400:]); sys.exit(1)
        d = json.loads(out.stdout.strip().split("\n")[-1])
        row.append("%s %.1f (%.1f-%.1f)" % (lib, d["repeats"]["median"] / 1e9, d["repeats"]["min"] / 1e9, d["repeats"]["max"] / 1e9))
    print("%9d games: " % n + "   ".join(row), flush=True)
