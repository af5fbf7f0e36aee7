#!/usr/bin/env python3
"""Driver for the counter passes of tools/play_only_counters.sh (GPU box): 128-card launches that each follow a
reset, so that their refill workgroups find empty lists and the SQ counters of the launch are the play role's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cards = int(sys.argv[2]) if len(sys.argv) > 2 else 128
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
for it in range(24):
    env.reset(episode=it * 100)
    env.krog_random(cards, auto_reset=True)
    torch.cuda.synchronize()
