#!/usr/bin/env python3
"""Diagnostic (GPU box): un-profiled per-launch slot time of each kernel when replayed
back-to-back from a captured graph, at several N.  Not part of the product or bench."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K


def replay_time(fn, per_graph, reps):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (per_graph * reps)   # us per launch


def main():
    out = {}
    for n in [int(x) for x in (sys.argv[1:] or ["65536"])]:
        env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
        env.reset()
        r = {}
        r["k_policy"] = replay_time(lambda: env.policy_random(), 48, 50)
        r["k_legal"] = replay_time(lambda: env.legal_actions(), 48, 50)
        r["k_prefetch_noop"] = replay_time(lambda: env.prefetch(), 48, 50)
        # real stepping, fused, no auto reset: games end and idle -> mostly cheap path
        env.reset()
        r["k_step_true_noreset"] = replay_time(lambda: env.step_random(auto_reset=False), 48, 1)
        env.reset()
        def pair():
            env.policy_random(); env.step(env.action, auto_reset=False)
        r["policy+step_noreset"] = replay_time(pair, 48, 1)
        for fused in (False, True):
            for pf in (2, 4, 8):
                env.reset()
                env.run_random(480, fused=fused, graph_chunk=48, auto_reset=True, prefetch_every=pf)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                env.run_random(4800, fused=fused, graph_chunk=48, auto_reset=True, prefetch_every=pf)
                torch.cuda.synchronize()
                r["run fused=%d pf=%d us/step" % (fused, pf)] = (time.perf_counter() - t0) / 4800 * 1e6
        out[n] = r
        env.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
