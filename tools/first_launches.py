#!/usr/bin/env python3
"""Diagnostic (GPU box): play-wave cycles of the first launches after a reset.  Launch 0 finds empty refill
lists (every line was dealt by the prefetch), so its play waves have their SIMDs to themselves; from launch 1
on a refill wave shares each SIMD for part of the launch.  The difference is what sharing costs a play wave."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cards = int(sys.argv[2]) if len(sys.argv) > 2 else 128
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
nw = (n + 63) // 64
st = torch.zeros((nw, 3), dtype=torch.int64, device="cuda")
for trial in range(3):
    env.reset(episode=trial * 1000)
    env.krog_random(4, auto_reset=True)          # (clocks up; consumes next to nothing)
    env.reset(episode=trial * 1000)
    torch.cuda.synchronize()
    _native.check(env.L.tarok_debug_stamps(env._h, C.c_void_p(st.data_ptr())))
    row = []
    for it in range(5):
        env.krog_random(cards, auto_reset=True)
        torch.cuda.synchronize()
        a = st.cpu().numpy()
        play = a[:, 2] & 0xFFFFFFFF
        span = int((a[:, 1].max() - a[:, 0].min()) * 10)
        row.append("launch %d: play median %d max %d cycles, span %.1f us" % (it, np.median(play), play.max(), span / 1e3))
    _native.check(env.L.tarok_debug_stamps(env._h, None))
    print("trial %d\n  " % trial + "\n  ".join(row), flush=True)
