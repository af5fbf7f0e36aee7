#!/usr/bin/env python3
"""Diagnostic (GPU box): time tarok_policy_mlp alone (graph replay), with and without feature output."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tarok_amd import TarokVecEnv, karte as K, selfplay as SP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = TarokVecEnv(n, seed=0, mix=K.MIX_ALL)
obs = env.reset()
for t in range(9):
    obs, _, _ = env.step(env.policy_random(obs), auto_reset=True)
words = obs.words.clone()
net = SP.PolicyNet(256).cuda()
w = [env.mfma_weight_order(net.fc1.weight), net.fc1.bias.detach().float().contiguous(), env.mfma_weight_order(net.fc2.weight),
     net.fc2.bias.detach().float().contiguous(), env.mfma_weight_order(net.head.weight), net.head.bias.detach().float().contiguous()]
feat = torch.empty((n, 256), dtype=torch.bfloat16, device="cuda")
a = torch.empty(n, dtype=torch.uint8, device="cuda"); lp = torch.empty(n, device="cuda"); v = torch.empty(n, device="cuda")
def bench(fn, reps=20, per=10):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(per): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * per) * 1e6
print("policy_mlp with features_out: %.1f us" % bench(lambda: env.policy_mlp(w, words, a, lp, v, features_out=feat)))
print("policy_mlp no features_out:   %.1f us" % bench(lambda: env.policy_mlp(w, words, a, lp, v)))
fw = torch.empty((n, 4), dtype=torch.int64, device="cuda")
w2 = torch.empty(n, dtype=torch.int64, device="cuda"); rw = torch.zeros((n, 4), dtype=torch.int16, device="cuda"); dn = torch.zeros(n, dtype=torch.uint8, device="cuda")
print("policy_mlp + feature words:   %.1f us" % bench(lambda: env.policy_mlp(w, words, a, lp, v, feature_words_out=fw)))
def two():
    env.policy_mlp(w, words, a, lp, v, feature_words_out=fw)
    env.step(a, auto_reset=True, obs_out=w2, reward_out=rw, done_out=dn)
print("policy_mlp then step:         %.1f us" % bench(two))
print("policy_step (one launch):     %.1f us" % bench(lambda: env.policy_step(w, words, w2, a, lp, v, feature_words_out=fw, reward_out=rw, done_out=dn)))
print("observe:                      %.1f us" % bench(lambda: env.observe(feat)))
lg = torch.randn((n, 64), device="cuda").to(torch.bfloat16)
print("sample_policy:                %.1f us" % bench(lambda: env.sample_policy(lg, words, a, lp)))
x = feat
w0 = net.fc1.weight.detach().to(torch.bfloat16)
print("one torch linear 256x256:     %.1f us" % bench(lambda: torch.nn.functional.linear(x, w0)))
# per-phase shader cycles of a workgroup (stamps: diagnostics build path)
import ctypes as C, numpy as np
from tarok_amd import _native
nb = (n + 127) // 128
st = torch.zeros((max(nb * 8, (n + 63) // 64 * 3), ), dtype=torch.int64, device="cuda")
_native.check(env.L.tarok_debug_stamps_sized(env._h, C.c_void_p(st.data_ptr()), st.numel()))
env.policy_mlp(w, words, a, lp, v)
torch.cuda.synchronize()
_native.check(env.L.tarok_debug_stamps(env._h, None))
t = st[: nb * 8].view(nb, 8).cpu().numpy()
d = np.diff(t[:, :6], axis=1)
print("phase cycles (median over workgroups): features %d, layer1 %d, layer2 %d, layer3 %d, sample %d" % tuple(np.median(d, axis=0)))
print("features split: state words %d, expansion %d" % (np.median(t[:, 6] - t[:, 0]), np.median(t[:, 1] - t[:, 6])))
print("total median %d  max %d" % (np.median(t[:, 5] - t[:, 0]), (t[:, 5] - t[:, 0]).max()))
