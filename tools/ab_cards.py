#!/usr/bin/env python3
"""Diagnostic (GPU box): same-box A/B of library builds (tools/ab/*.so), interleaved:
us/step of tarok_run_random for (mix, cards per launch) pairs."""
import sys, os, subprocess, json, time
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("all", 48), ("klop", 48)]
if os.environ.get("AB_CASES"):          # e.g. AB_CASES=all:48,all:64
    CASES = [(c.split(":")[0], int(c.split(":")[1])) for c in os.environ["AB_CASES"].split(",")]

def child(n):
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    from tarok_amd import TarokVecEnv, karte as K
    out = {}
    for mixname, cards in CASES:
        env = TarokVecEnv(n, seed=0, mix={"all": K.MIX_ALL, "klop": 16, "berac": 23}[mixname])
        chunk = cards * 4
        best = 1e9
        for rep in range(3):
            env.reset()
            env.run_random(max(1, 960 // chunk) * chunk, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
            torch.cuda.synchronize()
            steps = max(1, 9600 // chunk) * chunk
            t0 = time.perf_counter()
            env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e6)
        out["%s/%d" % (mixname, cards)] = round(best, 3)
        env.close()
    print(json.dumps(out))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(int(sys.argv[2]))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "65536"
    libs = sorted(f for f in os.listdir(os.path.join(HERE, "ab")) if f.endswith(".so"))
    for rep in range(2):
        for lib in libs:
            env = dict(os.environ, TAROK_LIB=os.path.join(HERE, "ab", lib))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", n], env=env, capture_output=True, text=True)
            print(lib.ljust(16), r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
