#!/usr/bin/env python3
"""Diagnostic (GPU box): 9,600 auto-reset lock-steps of 65,536 games through the other launch modes — the
two-kernel external-policy path (0), one card per launch (1), one trick per launch (4) and an unaligned 5
cards per launch — each compared with the CPU oracle at the end (one oracle run, ~45 s)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K
from oracle import oracle as O
n, steps = 65536, 9600
from concurrent.futures import ThreadPoolExecutor
parts = 64
with ThreadPoolExecutor(16) as ex:
    res = list(ex.map(lambda k: O.run_autoreset(3, 17 + k * (n // parts), n // parts, K.MIX_ALL, steps), range(parts)))
ref = {"episode": np.concatenate([r["episode"] for r in res]), "score_sum": np.concatenate([r["score_sum"] for r in res]),
       "lanes": np.concatenate([r["lanes"] for r in res], axis=1), "obs": np.concatenate([r["obs"] for r in res])}
print("oracle done", flush=True)
for cards, chunk in ((0, 192), (1, 192), (4, 192), (5, 0), (48, 192), (64, 640), (160, 1600)):
    env = TarokVecEnv(n, seed=3, mix=K.MIX_ALL, game_offset=17)
    env.reset()
    st = steps // max(1, cards) * max(1, cards)
    env.run_random(st, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
    if st != steps:
        env.run_random(steps - st, cards_per_launch=1, graph_chunk=0, auto_reset=True)
    ep, ss = env.counters()
    ok = bool((ep == ref["episode"]).all() and (ss == ref["score_sum"]).all() and (env.state() == ref["lanes"]).all()
              and (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all())
    print("cards", cards, "chunk", chunk, "equal to the oracle after", steps, "steps:", ok, flush=True)
    env.close()
