#!/usr/bin/env python3
"""Diagnostic (GPU box): us/step of the one-trick-per-launch mode vs refill fan (env TAROK_REFILL_FAN)."""
import sys, os, time, json, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from tarok_amd import TarokVecEnv, karte as K
    n = int(sys.argv[2])
    env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
    best = 1e9
    for rep in range(3):
        env.reset()
        env.run_random(960, cards_per_launch=4, graph_chunk=192, auto_reset=True, prefetch_every=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.run_random(4800, cards_per_launch=4, graph_chunk=192, auto_reset=True, prefetch_every=0)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 4800 * 1e6)
    print(best)
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "65536"
    for fan in (1, 2, 3, 4, 6, 8):
        env = dict(os.environ, TAROK_REFILL_FAN=str(fan))
        r = subprocess.run([sys.executable, __file__, "child", n], env=env, capture_output=True, text=True)
        print("fan", fan, r.stdout.strip().split("\n")[-1])
