#!/usr/bin/env python3
"""Diagnostic (GPU box): us per lock-step of tarok_run_random vs hipGraph size (lock-steps per graph), for the
one-card-per-launch modes (cards 1 = tarok_step_random, 0 = tarok_policy_random + tarok_step) and one trick."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
for cards in (1, 0, 4):
    row = {}
    for chunk in (0, 48, 192, 768, 1536):
        unit = max(1, cards)
        steps = 3072
        env.reset()
        env.run_random(max(chunk, 192), cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
        torch.cuda.synchronize()
        row[chunk] = round((time.perf_counter() - t0) / steps * 1e6, 3)
    print(json.dumps({"cards": cards, "us_per_lock_step_by_graph_chunk": row}), flush=True)
env.close()
