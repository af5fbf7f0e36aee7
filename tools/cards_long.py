#!/usr/bin/env python3
"""Diagnostic (GPU box): us per lock-step of tarok_run_random vs cards per launch beyond one game's length
(48 .. 192), graph of 4 launches, mixed contracts, 65,536 games; per-wave play cycles (median / slowest)."""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for cards in (48, 64, 72, 80):
    env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
    chunk = cards * 4
    best = 1e9
    for rep in range(3):
        env.reset()
        env.run_random(chunk * 2, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
        torch.cuda.synchronize()
        steps = chunk * (9600 // chunk)
        t0 = time.perf_counter()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    nw = n // 64
    st = torch.zeros((nw, 3), dtype=torch.int64, device="cuda")
    _native.check(env.L.tarok_debug_stamps(env._h, C.c_void_p(st.data_ptr())))
    env.krog_random(cards, auto_reset=True); env.krog_random(cards, auto_reset=True)
    torch.cuda.synchronize()
    a = st.cpu().numpy().view(np.uint64)
    play = (a[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    _native.check(env.L.tarok_debug_stamps(env._h, None))
    print(json.dumps({"cards": cards, "us_per_lock_step": round(best, 4), "G_steps_per_s": round(n / best / 1e3, 2),
                      "play_cycles_median": int(np.median(play)), "play_cycles_max": int(play.max()),
                      "per_card_median": round(float(np.median(play)) / cards, 1)}), flush=True)
    env.close()
