#!/usr/bin/env python3
"""Diagnostic (GPU box): us/step of tarok_run_random vs prefetch period."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
out = {}
for fused in (True, False):
    for pf in (2, 4, 6, 8, 12, 16, 24, 48):
        best = 1e9
        for rep in range(3):
            env.reset()
            env.run_random(960, fused=fused, graph_chunk=48, auto_reset=True, prefetch_every=pf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            env.run_random(4800, fused=fused, graph_chunk=48, auto_reset=True, prefetch_every=pf)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 4800 * 1e6)
        out["fused=%d pf=%d" % (fused, pf)] = round(best, 3)
for cards, pf in [(2, 0), (4, 0), (4, 8), (8, 0), (12, 0), (16, 0), (24, 0), (48, 0)]:
    best = 1e9
    chunk = 192 if 192 % cards == 0 else 48 * cards
    for rep in range(3):
        env.reset()
        env.run_random(960 // chunk * chunk, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True, prefetch_every=pf)
        torch.cuda.synchronize()
        steps = 4800 // chunk * chunk
        t0 = time.perf_counter()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True, prefetch_every=pf)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    out["cards=%d pf=%d chunk=%d" % (cards, pf, chunk)] = round(best, 3)
print(json.dumps(out, indent=1))
