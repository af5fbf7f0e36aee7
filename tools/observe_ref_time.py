#!/usr/bin/env python3
"""Diagnostic (GPU box): time of tarok_observe_ref (the reference-layout observation, 12,544 B per game written)
and of the exchange / hands encoders at 65,536 and 1 M games, mid-game; GB/s written against the HBM peak."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K
for n in (65536, 1 << 20):
    env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL, history=True)
    env.reset()
    env.krog_random(30, auto_reset=True)
    rec = torch.empty((n, K.REF_RECORD_BYTES), dtype=torch.uint8, device="cuda")
    meta = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    for name, fn, nbytes in (("observe_ref", lambda: env.observe_ref(rec, meta), n * K.REF_RECORD_BYTES),
                             ("observe (build-owned 256 features, bf16)", lambda: env.observe(), n * 512)):
        fn(); torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        print(json.dumps({"games": n, "kernel": name, "us": round(us, 1), "GB_written": round(nbytes / 1e9, 3),
                          "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 3)}), flush=True)
    env.close()
