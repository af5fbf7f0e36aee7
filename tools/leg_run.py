#!/usr/bin/env python3
"""Diagnostic (GPU box): ONE side leg of bench.py on its own — same plan, warm-up and timed region as bench.py's leg() —
so that a rocprofv3 --stats pass over this process holds the kernels of that leg only (round 3's all-modes CSV mixed
the 65,536- and 4,096-game legs and could not be read against the timed lock-step).

    python3 tools/leg_run.py <games> <cards: 0 = policy + step, 1 = step_random, >= 2 = krog> [lock_steps=7680]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tarok_amd import TarokVecEnv, karte as K

n, cards = int(sys.argv[1]), int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 7680
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
plan = bench.plan_region(max(1, passes // max(1, cards)) if cards >= 2 else passes, cards, 1536)
kw = {"done_rows": False} if cards <= 1 else {}
run = lambda p: env.run_random(p["lock_steps"], cards_per_launch=cards, graph_chunk=p["graph_chunk"], auto_reset=True, **kw)
env.reset(episode=0)
run(bench.plan_region(min(plan["launches"], 2 * plan["launches_per_graph"]), cards, 1536, plan["launches_per_graph"]))
best = None
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(plan)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print("%d games, cards %d: %d launches per region (%s), best of 3 regions: %.3f us per lock-step, %.3f us per launch"
      % (n, cards, plan["launches"] * (2 if cards == 0 else 1), bench.describe_mode(plan), best / plan["lock_steps"] * 1e6,
         best / (plan["launches"] * (2 if cards == 0 else 1)) * 1e6))
env.close()
