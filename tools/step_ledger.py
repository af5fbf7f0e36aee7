#!/usr/bin/env python3
"""Diagnostic (GPU box): one variant of the step-API traffic ledger.  Runs `steps` lock-steps of the external-policy
path (tarok_policy_random + tarok_step) or of tarok_step_random on `games` games, eagerly, one launch per
kernel and step, so that a rocprofv3 --pmc pass over this process sees every dispatch (dispatch j of the step
kernel plays card j mod 4 of its trick: all slots stay trick-aligned under auto-reset).

    python3 tools/step_ledger.py <games> <mode: two|random> <spec: 0|1|d> <done: 0|1> <reward: 0|1> [steps]

Driven by tools/step_ledger.sh (one process per variant and counter); summarised by tools/step_ledger_summary.py."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, _native, karte as K

n, mode, spec, want_done, want_reward = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4] == "1", sys.argv[5] == "1"
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 96
L = _native.lib()
kw = {}          # (<spec> selected the finish-path load mode while both existed: profiles/r03_step_ledger.txt; ignored now)
env = TarokVecEnv(n, seed=0, mix=int(os.environ.get("TAROK_LEDGER_MIX", K.MIX_ALL)), **kw)     # (16 = all Klop: games end together, every 48 cards)
env.reset()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)
rw = env.reward if want_reward else None
dn = env.done if want_done else None
for t in range(steps):
    if mode == "two":
        _native.check(L.tarok_policy_random(env._h, P(env.obs_words), P(env.action), s))
        _native.check(L.tarok_step(env._h, P(env.action), P(rw), P(dn), None, P(env.obs_words), K.AUTO_RESET, s))
    else:
        _native.check(L.tarok_step_random(env._h, P(env.action), P(rw), P(dn), None, P(env.obs_words), K.AUTO_RESET, s))
torch.cuda.synchronize()
ep, _ = env.counters()
print("games %d mode %s spec %s done %d reward %d steps %d: %d games finished" % (n, mode, spec, want_done, want_reward, steps, int(ep.sum())))
env.close()
