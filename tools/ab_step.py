#!/usr/bin/env python3
"""Diagnostic (GPU box): same-box A/B of the one-card step (two-kernel external-policy path and tarok_step_random)
over libraries under tools/ab/ at streaming batch sizes.

    python3 tools/ab_step.py lib1.so,lib2.so [sizes=1048576,4194304]      ('' = the product library)"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = r'''
import sys, os, time, json
sys.path.insert(0, %r)
import torch
from tarok_amd import TarokVecEnv, karte as K
n = int(sys.argv[1])
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
out = {}
for cards, name in ((0, "two"), (1, "random")):
    env.reset()
    steps = int(os.environ.get("AB_STEPS", 960 if n <= (1 << 20) else 384))
    env.run_random(192, cards_per_launch=cards, graph_chunk=192, auto_reset=True)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=192, auto_reset=True)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    out[name] = best
print(json.dumps(out))
''' % os.path.dirname(HERE)
libs = sys.argv[1].split(",")
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1048576,4194304").split(",")]
for n in sizes:
    for lib in libs:
        for rep in range(2):
            env = dict(os.environ)
            if lib:
                env["TAROK_LIB"] = os.path.join(HERE, "ab", lib)
            r = subprocess.run([sys.executable, "-c", WORKER, str(n)], env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                print(lib, n, "FAILED", r.stderr[-300:]); continue
            d = json.loads(r.stdout.strip().split("\n")[-1])
            print("%9d games  %-12s:  policy+step %7.2f us (%.3f of 8 TB/s by 54 B)   step_random %7.2f us (%.3f)"
                  % (n, lib or "product", d["two"], 54 * n / d["two"] / 8e6, d["random"], 54 * n / d["random"] / 8e6), flush=True)
