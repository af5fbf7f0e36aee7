import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, _native
n, cards = 65536, 24
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
env.reset()
env.run_random(960, cards_per_launch=cards, graph_chunk=192, auto_reset=True)
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
print("waves with deal_here lanes: %.3f  (mean lanes %.3f)   lazy fetch events per wave: mean %.2f   renew events: mean %.2f   lanes consuming: mean %.1f" % ((deal > 0).mean(), deal.mean(), lazy.mean(), renew.mean(), cons.mean()))
for name, x in (("deal", deal), ("lazy", lazy), ("renew", renew), ("cons", cons)):
    print(name, "corr with play cycles %.3f" % np.corrcoef(x, play)[0, 1], " by value:", {int(v): int(np.median(play[x == v])) for v in np.unique(x)[:8]})
