#!/usr/bin/env python3
"""Diagnostic (GPU box): same-box A/B of two library builds (tools/ab/*.so), interleaved."""
import sys, os, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
libs = sorted(f for f in os.listdir(os.path.join(HERE, "ab")) if f.endswith(".so"))
n = sys.argv[1] if len(sys.argv) > 1 else "65536"
for rep in range(3):
    for lib in libs:
        for fan in ("1",) if int(n) < (1 << 18) else ("8",):
            env = dict(os.environ, TAROK_LIB=os.path.join(HERE, "ab", lib), TAROK_REFILL_FAN=fan)
            r = subprocess.run([sys.executable, os.path.join(HERE, "fan_sweep.py"), "child", n], env=env, capture_output=True, text=True)
            print(lib, "fan", fan, r.stdout.strip().split("\n")[-1], flush=True)
