#!/usr/bin/env python3
"""Diagnostic (GPU box): three self-play iterations (rollout + update) — run under rocprofv3 --kernel-trace --stats for
the per-kernel times of the update (profiles/r03_update_kernel_stats.csv), or plainly for the wall times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K, selfplay
env = TarokVecEnv(65536, seed=0, mix=K.MIX_ALL)
sp = selfplay.SelfPlay(env, hidden=256, seed=0)
for it in range(4):
    st = sp.iterate(T=48, epochs=1, minibatches=8)
    print(it, "rollout %.2f ms  update %.2f ms  loss %.4f" % (st["rollout_s"] * 1e3, st["update_s"] * 1e3, st["loss"]), flush=True)
