#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wave timing of the step kernel from in-kernel stamps."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, _native

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
    env.reset()
    env.run_random(960, fused=True, graph_chunk=48, auto_reset=True, prefetch_every=4)   # reach steady state
    nw = (n + 63) // 64
    st = torch.zeros((nw, 3), dtype=torch.int64, device="cuda")
    _native.check(env.L.tarok_debug_stamps(env._h, C.c_void_p(st.data_ptr())))
    res = []
    for it in range(24):
        if it % 4 == 0:
            env.prefetch()
        env.step_random(auto_reset=True)
        torch.cuda.synchronize()
        a = st.cpu().numpy()
        span = (a[:, 1].max() - a[:, 0].min()) * 10      # ns (100 MHz)
        dur = (a[:, 1] - a[:, 0]) * 10
        cyc = a[:, 2]
        res.append(dict(span_ns=int(span), wave_ns_med=float(np.median(dur)), wave_ns_p99=float(np.percentile(dur, 99)),
                        wave_ns_max=int(dur.max()), cyc_med=float(np.median(cyc)), cyc_max=int(cyc.max()),
                        start_spread_ns=int((a[:, 0].max() - a[:, 0].min()) * 10)))
    _native.check(env.L.tarok_debug_stamps(env._h, None))
    print(json.dumps(res, indent=0))
    env.close()

if __name__ == "__main__":
    main()
