#!/usr/bin/env python3
"""Diagnostic (GPU box): what makes a play wave slow?  Builds the library with -DTK_EVENT_STAMPS (per-wave
counts of in-place deals, late line fetches and finishing events next to the cycle stamps), plays 48-card
launches at 65,536 games and correlates the counts with the waves' cycle counts."""
import sys, os, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("TAROK_LIB"):
    lib = os.path.join(ROOT, "gpurun_out", "libtarokenv_events.so")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DTK_EVENT_STAMPS",
                           "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(ROOT, "tarok_amd", "csrc", "tarok_env.hip")])
    os.environ["TAROK_LIB"] = lib
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, _native
n, cards = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 48
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
env.reset()
env.run_random(cards * 8, cards_per_launch=cards, graph_chunk=cards * 4, auto_reset=True)
nw = n // 64
st = torch.zeros((nw, 3), dtype=torch.int64, device="cuda")
_native.check(env.L.tarok_debug_stamps(env._h, C.c_void_p(st.data_ptr())))
rows = []
for it in range(8):
    env.krog_random(cards, auto_reset=True)
    torch.cuda.synchronize()
    a = st.cpu().numpy().view(np.uint64)
    ev = a[:, 0]
    total = (a[:, 2] >> np.uint64(32)).astype(np.int64); play = (a[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    deal = (ev >> np.uint64(48)).astype(np.int64); lazy = ((ev >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    renew = ((ev >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64); cons = (ev & np.uint64(0xFFFF)).astype(np.int64)
    rows.append((play, deal, lazy, renew, cons))
play = np.concatenate([r[0] for r in rows]); deal = np.concatenate([r[1] for r in rows]); lazy = np.concatenate([r[2] for r in rows])
renew = np.concatenate([r[3] for r in rows]); cons = np.concatenate([r[4] for r in rows])
print("play cycles: median %d p90 %d max %d" % (np.median(play), np.percentile(play, 90), play.max()))
print("waves with deal_here lanes: %.3f  (mean lanes %.3f)   fetches on the spot per wave: mean %.2f   renew events: mean %.2f   early top-ups per wave: mean %.2f" % ((deal > 0).mean(), deal.mean(), lazy.mean(), renew.mean(), cons.mean()))
for name, x in (("deal", deal), ("lazy", lazy), ("renew", renew), ("early", cons)):
    print(name, "corr with play cycles %.3f" % np.corrcoef(x, play)[0, 1], " by value:", {int(v): int(np.median(play[x == v])) for v in np.unique(x)[:8]})
