#!/usr/bin/env python3
"""Diagnostic (GPU box): us/step of tarok_run_random vs cards per launch, for several contract mixes
(mix 16 = all Klop: every game is 48 cards, so no slot ever finishes twice inside a launch)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
out = {}
for mixname, mix in (("all", K.MIX_ALL), ("klop", 16), ("berac", 16 + 7)):
    env = TarokVecEnv(n, seed=0, mix=mix)
    for cards in (1, 4, 8, 12, 16, 24, 48):
        best = 1e9
        chunk = 192 if 192 % (2 * cards) == 0 else 2 * cards * (96 // cards or 1)
        for rep in range(3):
            env.reset()
            env.run_random(960 // chunk * chunk, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
            torch.cuda.synchronize()
            steps = 4800 // chunk * chunk
            t0 = time.perf_counter()
            env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e6)
        out["%s cards=%d chunk=%d" % (mixname, cards, chunk)] = round(best, 3)
    env.close()
print(json.dumps(out, indent=1))
