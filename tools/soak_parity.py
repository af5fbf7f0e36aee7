#!/usr/bin/env python3
"""Diagnostic (GPU box): a run of the size and shape of the default bench run (49,152 warm-up + 5 x 12,800 timed
lock-steps, 128 cards per launch, graphs of 25 launches, 65,536 games, auto-reset) compared with the CPU oracle
at the end: episode numbers, score sums, canonical state, observation words (the oracle replays the 113,152
lock-steps of all 65,536 slots on the host cores)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K
from oracle import oracle as O
n, steps = 65536, 49152 + 5 * 12800
mix = int(sys.argv[1]) if len(sys.argv) > 1 else K.MIX_ALL       # (2 = contracts from the on-device bidding, 16 + c = contract c only)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
env = TarokVecEnv(n, seed=seed, mix=mix)
env.reset()
env.run_random(49152, cards_per_launch=128, graph_chunk=3200, auto_reset=True)
for _ in range(5):
    env.run_random(12800, cards_per_launch=128, graph_chunk=3200, auto_reset=True)
ep, ss = env.counters(); st = env.state(); ob = env.obs_words.cpu().numpy().view(np.uint64)
print("mix", mix, "seed", seed, "gpu done", time.time() - t0, "episodes", int(ep.sum()), flush=True)
from concurrent.futures import ThreadPoolExecutor           # (the oracle call releases the GIL: slots in parallel)
parts = 64
with ThreadPoolExecutor(16) as ex:
    res = list(ex.map(lambda k: O.run_autoreset(seed, k * (n // parts), n // parts, mix, steps), range(parts)))
ref = {"episode": np.concatenate([r["episode"] for r in res]), "score_sum": np.concatenate([r["score_sum"] for r in res]),
       "lanes": np.concatenate([r["lanes"] for r in res], axis=1), "obs": np.concatenate([r["obs"] for r in res])}
print("oracle done", time.time() - t0, "steps", sum(r["total_steps"] for r in res), flush=True)
print("episodes equal", bool((ep == ref["episode"]).all()), "scores equal", bool((ss == ref["score_sum"]).all()),
      "state equal", bool((st == ref["lanes"]).all()), "obs equal", bool((ob == ref["obs"]).all()))
