#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wave timing of k_krog (play part vs in-launch refill)."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cards = int(sys.argv[2]) if len(sys.argv) > 2 else 4
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
env.reset()
env.run_random(cards * 8, cards_per_launch=cards, graph_chunk=cards * 4, auto_reset=True)
nw = (n + 63) // 64
st = torch.zeros((nw, 3), dtype=torch.int64, device="cuda")
_native.check(env.L.tarok_debug_stamps(env._h, C.c_void_p(st.data_ptr())))
res = []
for it in range(10):
    env.krog_random(cards, auto_reset=True)
    torch.cuda.synchronize()
    a = st.cpu().numpy()
    total = (a[:, 2] >> 32) & 0xFFFFFFFF
    play = a[:, 2] & 0xFFFFFFFF
    res.append(dict(span_ns=int((a[:, 1].max() - a[:, 0].min()) * 10), play_cyc_med=float(np.median(play)), play_cyc_max=int(play.max()),
                    total_cyc_med=float(np.median(total)), total_cyc_p90=float(np.percentile(total, 90)), total_cyc_max=int(total.max()),
                    waves_refilling=int((total - play > 2000).sum())))
_native.check(env.L.tarok_debug_stamps(env._h, None))
for r in res: print(r)
