#!/usr/bin/env python3
"""Diagnostic (GPU box): the launch kinds MIXED on one env — stretches of one-card launches (tarok_step_random and the
two-kernel path, graph-replayed and eager: with the bulk deals of small batches their emptied lines wait on the stretch
lists), cut by launches of the multi-card kernel (4 .. 128 cards: they drop the stretch lists, so slots come back to stale
lines, deal in place and list all fourteen) — 65,536 and 1,048,576 games, auto-reset throughout, seeded segment lengths;
compared with the CPU oracle at the end."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from tarok_amd import TarokVecEnv, karte as K
from oracle import oracle as O

def oracle(seed, n, steps, parts=64):
    if n % parts:
        return O.run_autoreset(seed, 0, n, K.MIX_ALL, steps, threads=16)
    with ThreadPoolExecutor(16) as ex:
        res = list(ex.map(lambda k: O.run_autoreset(seed, k * (n // parts), n // parts, K.MIX_ALL, steps), range(parts)))
    return {"episode": np.concatenate([r["episode"] for r in res]), "score_sum": np.concatenate([r["score_sum"] for r in res]),
            "lanes": np.concatenate([r["lanes"] for r in res], axis=1), "obs": np.concatenate([r["obs"] for r in res])}

CONFIGS = ((65536, 6000, None), (65536, 3000, 0), (1 << 20, 1200, None), (1 << 20, 1200, 1))
if os.environ.get("SOAK_MORE"):                  # (other sizes — fan 4 — and other segment sequences)
    CONFIGS = ((1 << 18, 4000, None), (1 << 19, 2500, None), (1 << 18, 2500, 0), (65536, 6001, None), (65536, 6002, None), (65536, 6003, 0))
if os.environ.get("SOAK_DEFAULT_ONLY"):          # (libraries without TAROK_OPT_LAZY_REFILL: tools/ab/*.so)
    CONFIGS = tuple(c for c in CONFIGS if c[2] is None)
if os.environ.get("SOAK_MORE2"):                 # (longer runs, other sequences again)
    CONFIGS = ((65536, 12007, None), (65536, 12011, 0), (1 << 20, 2503, None), (1 << 20, 2509, 1), (1 << 17, 9001, None), (20000, 9007, None))
for n, target, lazy in CONFIGS:
    rnd = np.random.RandomState(n % 1000 + target)
    env = TarokVecEnv(n, seed=11, mix=K.MIX_ALL, lazy_refill=lazy)
    env.reset()
    steps, log = 0, []
    while steps < target:
        kind = rnd.choice(["random", "two", "eager", "krog"], p=[0.35, 0.3, 0.15, 0.2])
        if kind == "random":
            k = int(rnd.choice([16, 48, 80, 112])); env.run_random(k, cards_per_launch=1, graph_chunk=16, auto_reset=True)
        elif kind == "two":
            k = int(rnd.choice([20, 60, 100])); env.run_random(k, cards_per_launch=0, graph_chunk=20, auto_reset=True)
        elif kind == "eager":
            k = int(rnd.randint(1, 23))
            for _ in range(k):
                env.step_random(auto_reset=True)
        else:
            k = int(rnd.choice([4, 8, 28, 48, 128])); env.krog_random(k, auto_reset=True)
        steps += k; log.append((kind, k))
    ref = oracle(11, n, steps)
    ep, ss = env.counters()
    ok = bool((ep == ref["episode"]).all() and (ss == ref["score_sum"]).all() and (env.state() == ref["lanes"]).all()
              and (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all())
    print("%8d games, lazy %s: %d segments (%d one-card stretches, %d multi-card launches), %d lock-steps, %d games finished: equal to the oracle %s"
          % (n, lazy, len(log), sum(1 for a, _ in log if a != "krog"), sum(1 for a, _ in log if a == "krog"), steps, int(ep.sum()), ok), flush=True)
    env.close()
