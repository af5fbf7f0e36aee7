#!/usr/bin/env python3
"""Round-4 diagnostic (GPU box): tools/soak_mixed.py's failing sequence on ONE library (TAROK_LIB), with switches that
separate host causes from device causes, and a dump of the diagnostic records of the tools/diag_refill builds.

    python tools/diag_refill/diag_soak.py [--n 65536] [--target 6000] [--lazy -1|0|1] [--eager] [--sync] [--repeat R] [--tag NAME]

    --eager   no graphs at all: every segment as eager launches (graph_chunk = 0)
    --sync    torch.cuda.synchronize() after every segment
    --nokrog  no multi-card launches in the sequence (their share goes to the one-card kinds)
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from tarok_amd import TarokVecEnv, karte as K, _native
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--target", type=int, default=6000)
ap.add_argument("--lazy", type=int, default=-1)
ap.add_argument("--eager", action="store_true")
ap.add_argument("--sync", action="store_true")
ap.add_argument("--nokrog", action="store_true")
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--tag", default="")
args = ap.parse_args()


def oracle(seed, n, steps, parts=64):
    if n % parts:
        return O.run_autoreset(seed, 0, n, K.MIX_ALL, steps, threads=16)
    with ThreadPoolExecutor(16) as ex:
        res = list(ex.map(lambda k: O.run_autoreset(seed, k * (n // parts), n // parts, K.MIX_ALL, steps), range(parts)))
    return {"episode": np.concatenate([r["episode"] for r in res]), "score_sum": np.concatenate([r["score_sum"] for r in res]),
            "lanes": np.concatenate([r["lanes"] for r in res], axis=1), "obs": np.concatenate([r["obs"] for r in res])}


def diag_records():
    L = _native.lib()
    if not hasattr(L, "tarok_diag_read"):
        return None, None
    L.tarok_diag_read.restype = C.c_int
    L.tarok_diag_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    buf = np.zeros((4096, 16), np.uint64)
    tick = C.c_uint32(0)
    n = L.tarok_diag_read(buf.ctypes.data, 4096, C.byref(tick))
    return n, buf[:max(0, min(n, 4096))], int(tick.value)


lib_name = os.path.basename(os.environ.get("TAROK_LIB", "product"))
n, target = args.n, args.target
lazy = None if args.lazy < 0 else args.lazy
seen_diag = 0
for rep in range(args.repeat):
    rnd = np.random.RandomState(n % 1000 + target)
    env = TarokVecEnv(n, seed=11, mix=K.MIX_ALL, lazy_refill=lazy)
    env.reset()
    steps, log = 0, []
    t0 = time.time()
    while steps < target:
        kind = rnd.choice(["random", "two", "eager", "krog"], p=[0.35, 0.3, 0.15, 0.2])
        if kind == "random":
            k = int(rnd.choice([16, 48, 80, 112])); env.run_random(k, cards_per_launch=1, graph_chunk=0 if args.eager else 16, auto_reset=True)
        elif kind == "two":
            k = int(rnd.choice([20, 60, 100])); env.run_random(k, cards_per_launch=0, graph_chunk=0 if args.eager else 20, auto_reset=True)
        elif kind == "eager":
            k = int(rnd.randint(1, 23))
            for _ in range(k):
                env.step_random(auto_reset=True)
        else:
            k = int(rnd.choice([4, 8, 28, 48, 128]))
            if args.nokrog:
                for _ in range(k):
                    env.step_random(auto_reset=True)
            else:
                env.krog_random(k, auto_reset=True)
        if args.sync:
            torch.cuda.synchronize()
        steps += k; log.append((kind, k))
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    ref = oracle(11, n, steps)
    ep, ss = env.counters()
    st = env.state()
    bad_ep = np.flatnonzero(ep != ref["episode"])
    bad_ss = np.flatnonzero((ss != ref["score_sum"]).any(axis=1))
    bad_st = np.flatnonzero((st != ref["lanes"]).any(axis=0))
    bad_ob = np.flatnonzero(env.obs_words.cpu().numpy().view(np.uint64) != ref["obs"])
    ok = not (len(bad_ep) or len(bad_ss) or len(bad_st) or len(bad_ob))
    print("[%s%s] %s n=%d lazy=%s eager=%d sync=%d nokrog=%d rep=%d: %d segments, %d lock-steps, %d games finished, %.2f s: %s"
          % (lib_name, (" " + args.tag) if args.tag else "", "PASS" if ok else "FAIL", n, lazy, args.eager, args.sync, args.nokrog, rep, len(log), steps,
             int(ep.sum()), t_gpu, "equal to the oracle" if ok else
             "bad slots: episode %d, score_sum %d, state %d, obs %d; first %s" % (len(bad_ep), len(bad_ss), len(bad_st), len(bad_ob), bad_st[:8].tolist())), flush=True)
    if not ok:
        for s in bad_st[:6]:
            print("    slot %d: episode gpu %d ref %d; lanes gpu %s" % (s, ep[s], ref["episode"][s], " ".join("%016x" % int(x) for x in st[:, s])))
            print("    %s             ref %s" % (" " * len(str(s)), " ".join("%016x" % int(x) for x in ref["lanes"][:, s])))
    d = diag_records()
    if d[0] is not None:
        cnt, recs, tick = d
        new = recs[seen_diag:]
        print("    diag: %d records in all (%d new), k_tick count %d" % (cnt, max(0, cnt - seen_diag), tick), flush=True)
        for r in new[:24]:
            kind = int(r[0]) & 0xFFFFFFFF
            if kind == 2:
                print("    PHASE where=%d count=%d block=%d tid=%d phase=%d tick=%d" % (int(r[3]), int(r[0]) >> 32, int(r[1]) >> 40, int(r[1]) & 0xFFFFFF, int(r[2]) & 0xFFFFFFFF, int(r[2]) >> 32))
            else:
                print("    LINE kind=%d count=%d slot=%d block=%d wanted_ep=%d tag_read=%d tick=%d nep_reread=%d" %
                      (kind, int(r[0]) >> 32, int(r[1]) & ((1 << 40) - 1), int(r[1]) >> 40, int(r[2]) & 0xFFFFFFFF, int(r[2]) >> 32, int(r[15]) >> 32, int(r[15]) & 0xFFFFFFFF))
                print("         read     %s" % " ".join("%016x" % int(x) for x in r[3:8]))
                print("         expected %s" % " ".join("%016x" % int(x) for x in r[8:13]))
                rr = [int(x) for x in r[3:7]]; ee = [int(x) for x in r[8:12]]
                M = (1 << 64) - 1
                f = lambda a, b: a ^ ((b * 0x9E3779B97F4A7C15) & M)
                print("         second read: play pair %s, seat pair %s" % (
                    "= expected" if int(r[13]) == f(ee[0], ee[1]) else ("= first read" if int(r[13]) == f(rr[0], rr[1]) else "OTHER"),
                    "= expected" if int(r[14]) == f(ee[2], ee[3]) else ("= first read" if int(r[14]) == f(rr[2], rr[3]) else "OTHER")))
        if len(new):
            out = os.path.join("gpurun_out", "diag_%s_%s_%d.json" % (lib_name, args.tag or "run", rep))
            os.makedirs("gpurun_out", exist_ok=True)
            json.dump({"lib": lib_name, "args": vars(args), "records": [[int(x) for x in r] for r in new]}, open(out, "w"))
        seen_diag = max(seen_diag, min(cnt, 4096))
    env.close()
