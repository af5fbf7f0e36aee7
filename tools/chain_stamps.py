#!/usr/bin/env python3
"""Diagnostic (GPU box): where a workgroup of k_learn_chain spends its cycles (in-kernel s_memtime stamps through
tarok_debug_stamps): gather + expansion | layer 1 | layer 2 | layer 3 + loss | dH2 | dH1, medians over the workgroups
of one minibatch of 393,216 samples (two workgroups share a CU: a workgroup's cycles include the other's turns)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tarok_amd import TarokVecEnv, karte as K, selfplay, _native
env = TarokVecEnv(65536, seed=0, mix=K.MIX_ALL)
sp = selfplay.SelfPlay(env, hidden=256, seed=0)
sp.iterate(T=48, epochs=1, minibatches=8)
blocks = (393216 + 95) // 96
st = torch.zeros((blocks, 8), dtype=torch.int64, device="cuda")
_native.check(env.L.tarok_debug_stamps_sized(env._h, C.c_void_p(st.data_ptr()), st.numel()))
buf = sp.collect(48)
sp.update_fused(buf, epochs=1, minibatches=8)
torch.cuda.synchronize()
_native.check(env.L.tarok_debug_stamps(env._h, None))
s = st.cpu().numpy().astype(np.int64)
if len(sys.argv) > 1 and sys.argv[1] == "fine":      # a library built with -DLN_FINE_STAMPS (TAROK_LIB)
    d = np.diff(s[:, :8], axis=1)
    print("k_learn_chain, layers 1-2 in detail (wave 0's clock; two workgroups share the CU), shader cycles (median | p10 | p90):")
    for k, nm in enumerate(["gather+expand", "layer 1 GEMM", "layer 1 barrier + epilogue + barrier", "layer 1 tile store (issue)", "layer 2 GEMM",
                            "layer 2 barrier + epilogue + barrier", "layer 2 tile store (issue)"]):
        print("  %-38s %8d | %8d | %8d" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
    sys.exit(0)
d = np.diff(s[:, :7], axis=1)
names = ["gather+expand", "layer 1", "layer 2", "layer 3 + loss", "dH2", "dH1"]
print("k_learn_chain, last minibatch of an update (4,096 workgroups of 96 samples), shader cycles per phase (median | p10 | p90):")
for k, nm in enumerate(names):
    print("  %-16s %8d | %8d | %8d" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
tot = s[:, 6] - s[:, 0]
print("  %-16s %8d | %8d | %8d" % ("whole workgroup", np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
