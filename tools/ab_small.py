#!/usr/bin/env python3
"""Diagnostic (GPU box): same-box A/B of library builds (tools/ab/*.so) on the small-launch modes:
us per lock-step of tarok_run_random with 1 and 4 cards per launch (timing only: parity is not checked)."""
import sys, os, subprocess, json, time
HERE = os.path.dirname(os.path.abspath(__file__))

def child(n):
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    from tarok_amd import TarokVecEnv, karte as K
    out = {}
    env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
    for cards in (0, 1, 4, 48):
        env.reset()
        env.run_random(192 * max(1, cards // 4), cards_per_launch=cards, graph_chunk=192, auto_reset=True)
        torch.cuda.synchronize()
        steps = 3840
        t0 = time.perf_counter()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=192, auto_reset=True)
        torch.cuda.synchronize()
        out["cards%d" % cards] = round((time.perf_counter() - t0) / steps * 1e6, 3)
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
            print(lib.ljust(18), r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
