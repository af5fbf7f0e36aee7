#!/usr/bin/env python3
"""Diagnostic (GPU box): us/step vs (cards per launch, refill fan), mixed contracts."""
import sys, os, subprocess, json, time
HERE = os.path.dirname(os.path.abspath(__file__))
def child(n):
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    from tarok_amd import TarokVecEnv, karte as K
    out = {}
    env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
    for cards in (12, 16, 24, 32, 48):
        chunk = 192 if 192 % (2 * cards) == 0 else 2 * cards * (96 // cards or 1)
        best = 1e9
        for rep in range(3):
            env.reset()
            env.run_random(960 // chunk * chunk, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
            torch.cuda.synchronize()
            steps = 9600 // chunk * chunk
            t0 = time.perf_counter()
            env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e6)
        out["c%d" % cards] = round(best, 3)
    print(json.dumps(out))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(int(sys.argv[2]))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "65536"
    for fan in ("1", "2", "4", "8"):
        env = dict(os.environ, TAROK_REFILL_FAN=fan)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", n], env=env, capture_output=True, text=True)
        print("fan", fan, r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
