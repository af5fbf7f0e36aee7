#!/usr/bin/env python3
"""Diagnostic (GPU box): us/step of tarok_run_random vs hipGraph chunk length."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
out = {}
for fused in (True, False):
    for chunk in (24, 48, 96, 192, 384, 768):
        best = 1e9
        steps = 768 * 8
        for rep in range(3):
            env.reset()
            env.run_random(768, fused=fused, graph_chunk=chunk, auto_reset=True, prefetch_every=8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            env.run_random(steps, fused=fused, graph_chunk=chunk, auto_reset=True, prefetch_every=8)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e6)
        out["fused=%d chunk=%d" % (fused, chunk)] = round(best, 3)
print(json.dumps(out, indent=1))
