#!/usr/bin/env python3
"""Diagnostic (GPU box): the step API (tarok_policy_random + tarok_step, one card per launch, state through HBM
on every card: the path SURVEY 8d's 54 algorithmic B/step describe) at growing batch sizes: us per lock-step,
env steps/s and the algorithmic bandwidth 54 B x N / time against the 8 TB/s HBM peak; also tarok_step_random
(policy in-kernel, one launch per lock-step)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K
rows = []
for n in (65536, 1 << 20, 1 << 22, 1 << 24):
    env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
    row = {"games": n}
    for cards, name in ((0, "policy_plus_step"), (1, "step_random")):
        env.reset()
        steps = 960 if n <= (1 << 20) else 192
        env.run_random(192, cards_per_launch=cards, graph_chunk=192, auto_reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=192, auto_reset=True)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / steps * 1e6
        row[name] = {"us_per_lock_step": round(us, 3), "G_env_steps_per_s": round(n / us / 1e3, 2),
                     "algorithmic_GBps_54B": round(54 * n / us / 1e3, 1), "frac_of_8TBps": round(54 * n / us / 1e3 / 8000, 4)}
    rows.append(row)
    print(json.dumps(row), flush=True)
    env.close()
