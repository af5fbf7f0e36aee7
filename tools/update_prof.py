#!/usr/bin/env python3
"""Diagnostic (GPU box): two self-play iterations (rollout + PPO update) for rocprofv3 --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K, selfplay
env = TarokVecEnv(65536, seed=0, mix=K.MIX_ALL)
sp = selfplay.SelfPlay(env, hidden=256, seed=0)
for it in range(3):
    st = sp.iterate(T=48, epochs=1, minibatches=8)
    print(it, "rollout %.2f ms  update %.2f ms" % (st["rollout_s"] * 1e3, st["update_s"] * 1e3), flush=True)
