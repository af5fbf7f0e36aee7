#!/usr/bin/env python3
"""Diagnostic (GPU box): where do a play wave's cycles go within a trick?  Builds the library with
-DTK_CARD_STAMPS (s_memtime around cards 0-2 and around the trick's 4th card in the trick-aligned
loop) and reports, for 48-card launches at 65,536 games, the cycles per trick of each part."""
import sys, os, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("TAROK_LIB"):
    lib = os.path.join(ROOT, "gpurun_out", "libtarokenv_cards.so")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DTK_CARD_STAMPS",
                           "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(ROOT, "tarok_amd", "csrc", "tarok_env.hip")])
    os.environ["TAROK_LIB"] = lib
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cards = int(sys.argv[2]) if len(sys.argv) > 2 else 48
for mixname, mix in (("all", K.MIX_ALL), ("klop", 16)):
    env = TarokVecEnv(n, seed=0, mix=mix)
    env.reset()
    env.run_random(cards * 8, cards_per_launch=cards, graph_chunk=cards * 4, auto_reset=True)
    nw = n // 64
    st = torch.zeros((nw, 3), dtype=torch.int64, device="cuda")
    _native.check(env.L.tarok_debug_stamps(env._h, C.c_void_p(st.data_ptr())))
    a012, a3, tot, s0, s1, s2 = [], [], [], [], [], []
    for it in range(8):
        env.krog_random(cards, auto_reset=True)
        torch.cuda.synchronize()
        a = st.cpu().numpy().view(np.uint64)
        a012.append((a[:, 0] >> np.uint64(32)).astype(np.int64)); a3.append((a[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64))
        tot.append((a[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64))
        s0.append((a[:, 1] >> np.uint64(32)).astype(np.int64)); s1.append((a[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64))
        s2.append((a[:, 2] >> np.uint64(32)).astype(np.int64))
    a012, a3, tot = np.concatenate(a012), np.concatenate(a3), np.concatenate(tot)
    s0, s1, s2 = np.concatenate(s0), np.concatenate(s1), np.concatenate(s2)
    tr = cards // 4
    print("%s: per trick, median over waves: cards 0-2 %.0f cycles (%.0f per card), 4th card %.0f cycles; play part of the launch %.0f cycles "
          "= %.0f per trick (stamps included)" % (mixname, np.median(a012) / tr, np.median(a012) / tr / 3, np.median(a3) / tr, np.median(tot), np.median(tot) / tr))
    print("    the 4th card: rules + action store %.0f | scoring queue (push, drain) %.0f | renewal (swap-in) %.0f | legal mask, observation, stores %.0f"
          "   (each with one s_memtime)" % (np.median(s0) / tr, np.median(s1) / tr, np.median(s2) / tr, np.median(a3 - s0 - s1 - s2) / tr))
    _native.check(env.L.tarok_debug_stamps(env._h, None))
    env.close()
