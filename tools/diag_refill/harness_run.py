#!/usr/bin/env python3
"""Round-4 diagnostic (GPU box): the refill role of the real step kernels (TAROK_LIB = a tools/diag_refill harness build)
fed with hand-built refill lists — every slot of every group, `per_slot` episodes each — and every line it writes compared
with a straight re-deal by a separate kernel.  Turns the mixed-launch corruption into a unit test of the deal code as the
step kernels compile it.

    TAROK_LIB=tools/ab/diag_harness.so python tools/diag_refill/harness_run.py [--n 65536] [--reps 20]
"""
import argparse, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, _native

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--combos", default="")
args = ap.parse_args()

L = _native.lib()
L.tarok_diag_read.restype = C.c_int
L.tarok_diag_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
L.tarok_diag_harness.restype = C.c_int
L.tarok_diag_harness.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int]
KIND = {0: "k_step<random>", 1: "k_step<action>", 2: "k_play_wide(4 cards)"}

env = TarokVecEnv(args.n, seed=11, mix=K.MIX_ALL)
env.reset()
torch.cuda.synchronize()
seen = 0
buf = np.zeros((4096, 16), np.uint64)
combos = [(0, 1, 0), (0, 2, 0), (0, 4, 0), (0, 14, 0), (0, 14, 1), (0, 14, 2), (1, 14, 0), (2, 1, 0), (2, 4, 0), (2, 14, 0), (2, 14, 2)]
if args.combos:
    combos = [tuple(int(x) for x in c.split(",")) for c in args.combos.split(";")]
for kind, per_slot, order in combos:
    ep0 = 100 + 20 * per_slot
    rc = L.tarok_diag_harness(env._h, kind, per_slot, ep0, order, args.reps)
    tick = C.c_uint32(0)
    cnt = L.tarok_diag_read(buf.ctypes.data, 4096, C.byref(tick))
    new = buf[seen:min(cnt, 4096)].copy()
    deals = args.reps * args.n * per_slot
    print("%-22s per_slot=%2d order=%d reps=%d rc=%d: %d wrong lines of %d deals (%.2e)%s"
          % (KIND[kind], per_slot, order, args.reps, rc, cnt - seen, deals, (cnt - seen) / deals, " [record buffer full]" if cnt >= 4096 else ""), flush=True)
    if len(new):
        slot = (new[:, 1] & np.uint64((1 << 40) - 1)).astype(np.int64)
        ep = (new[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        tag = (new[:, 2] >> np.uint64(32)).astype(np.int64)
        rep = (new[:, 0] >> np.uint64(32)).astype(np.int64)
        k = ep - ep0 - 1
        t = slot % 256
        if order == 1: pos = 255 - t
        elif order == 2: pos = np.array([[(x * 37 + 11) % 256 for x in range(256)].index(int(v)) for v in t])
        else: pos = t
        print("    upper half of the groups: %d of %d; tag wrong: %d; by iteration k: %s" % (int((slot >= args.n // 2).sum()), len(new), int((tag != ep).sum()), dict(sorted(collections.Counter(k.tolist()).items()))))
        print("    by wave of the refill workgroup (list position / 64): %s; by rep: %s" % (dict(sorted(collections.Counter((pos // 64).tolist()).items())), dict(sorted(collections.Counter(rep.tolist()).items()))))
        print("    by lane: %s" % dict(sorted(collections.Counter((pos % 64).tolist()).items())))
        words = collections.Counter()
        for r in new:
            words[tuple(int(r[3 + w] != r[8 + w]) for w in range(5))] += 1
        print("    words differing (x0 x1 y0 y1 key): %s" % dict(words))
        again = collections.Counter(zip(slot.tolist(), ep.tolist()))
        print("    (slot, episode) pairs wrong in more than one rep: %d of %d distinct" % (sum(1 for v in again.values() if v > 1), len(again)))
        for r in new[:4]:
            print("      slot %d ep %d rep %d: read %s" % (int(r[1]) & ((1 << 40) - 1), int(r[2]) & 0xFFFFFFFF, int(r[0]) >> 32, " ".join("%016x" % int(x) for x in r[3:8])))
            print("      %s expected %s" % (" " * 22, " ".join("%016x" % int(x) for x in r[8:13])))
    seen = min(cnt, 4096)
    if seen >= 4096:
        print("record buffer full: stopping"); break
env.close()
