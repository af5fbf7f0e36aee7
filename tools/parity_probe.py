#!/usr/bin/env python3
"""Diagnostic (GPU box): headline-mode parity vs the oracle for a few (n, fan, offset, cards) settings."""
import sys, os, subprocess, json
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
def child(n, off, cards, steps, chunk):
    import numpy as np, torch
    from tarok_amd import TarokVecEnv, karte as K
    from oracle import oracle as O
    ref = O.run_autoreset(5, off, n, K.MIX_ALL, steps)
    env = TarokVecEnv(n, seed=5, mix=K.MIX_ALL, game_offset=off)
    env.reset()
    env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True)
    ep, ss = env.counters()
    bad = np.nonzero(ep != ref["episode"])[0]
    st = (env.state() == ref["lanes"]).all(axis=0)
    print(json.dumps(dict(n=n, off=off, cards=cards, steps=steps, chunk=chunk, fan=os.environ.get("TAROK_REFILL_FAN"),
                          bad_ep=int(bad.size), first=bad[:8].tolist(), bad_state=int((~st).sum()),
                          ep_gpu=ep[bad[:4]].tolist(), ep_ref=ref["episode"][bad[:4]].tolist())))
if sys.argv[1] == "child":
    child(*[int(x) for x in sys.argv[2:7]])
else:
    cases = [(1 << 20, 987654321, 24, 96, 48, None), (1 << 20, 0, 24, 96, 48, None), (1 << 20, 0, 24, 96, 48, "1"),
             (1 << 18, 0, 24, 96, 48, None), (1 << 17, 0, 24, 96, 48, None), (1 << 17, 0, 24, 96, 48, "8"),
             (1 << 20, 0, 4, 96, 48, None), (1 << 20, 0, 24, 96, 0, None), (1 << 20, 0, 24, 48, 0, None),
             # one card per launch, the two-kernel external-policy path, ragged N
             (1 << 20, 0, 1, 96, 48, None), (1 << 20, 0, 0, 96, 48, None), (1 << 20, 5, 0, 64, 0, None),
             (1000003, 77, 24, 96, 48, None), (1000003, 77, 1, 64, 0, None), (1 << 22, 0, 24, 48, 48, None),
             (1 << 20, 11, 48, 192, 96, None), (1000003, 77, 48, 96, 96, None), (1 << 22, 0, 48, 96, 96, None), (1 << 20, 0, 48, 96, 0, "1")]
    for n, off, cards, steps, chunk, fan in cases:
        env = dict(os.environ)
        if fan: env["TAROK_REFILL_FAN"] = fan
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), str(off), str(cards), str(steps), str(chunk)], env=env, capture_output=True, text=True)
        print(r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
