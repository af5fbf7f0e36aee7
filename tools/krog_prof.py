import sys, os
sys.path.insert(0, "/root/repo")
import torch
from tarok_amd import TarokVecEnv, karte as K
env = TarokVecEnv(65536, seed=0, mix=K.MIX_ALL)
env.reset()
for cards in (4, 8):
    env.run_random(1920, cards_per_launch=cards, graph_chunk=192, auto_reset=True, prefetch_every=8)
torch.cuda.synchronize()
