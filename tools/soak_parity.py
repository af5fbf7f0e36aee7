#!/usr/bin/env python3
"""Diagnostic (GPU box): the whole default bench run (960 warm-up + 9,600 timed lock-steps, 48 cards per
launch, 65,536 games, auto-reset) replayed and compared with the CPU oracle at the end: episode numbers,
score sums, canonical state, observation words (~45 s of oracle time on one core)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K
from oracle import oracle as O
n, steps = 65536, 10560
t0 = time.time()
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
env.reset()
env.run_random(960, cards_per_launch=48, graph_chunk=192, auto_reset=True)
env.run_random(9600, cards_per_launch=48, graph_chunk=192, auto_reset=True)
ep, ss = env.counters(); st = env.state(); ob = env.obs_words.cpu().numpy().view(np.uint64)
print("gpu done", time.time() - t0, "episodes", int(ep.sum()), flush=True)
ref = O.run_autoreset(0, 0, n, K.MIX_ALL, steps)
print("oracle done", time.time() - t0, flush=True)
print("episodes equal", bool((ep == ref["episode"]).all()), "scores equal", bool((ss == ref["score_sum"]).all()),
      "state equal", bool((st == ref["lanes"]).all()), "obs equal", bool((ob == ref["obs"]).all()))
