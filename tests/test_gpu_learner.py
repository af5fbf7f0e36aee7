"""GPU tests of the fused learner (tarok_learn_*: returns, forward + loss + backward chain on MFMA, split-K
weight gradients through transposed LDS reads, clip + Adam on flat vectors) against float32 torch — autograd
for every gradient.  Build-owned code (the reference has no policy-gradient learner): these are numerics
tests of kernels, with bf16 tolerances written where they apply.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import tarok_amd
    tarok_amd.build()
    return tarok_amd


def frag_order(w, ksteps):
    """[rows, 16 * ksteps] -> the MFMA fragment order the kernels read: [rows / 32][k-step][half][32 rows][8]."""
    import torch
    rows = w.shape[0]
    return w.to(torch.bfloat16).view(rows // 32, 32, ksteps, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def test_learn_adam_vs_torch(T):
    """tarok_learn_adam: three steps of clip_grad_norm_ + torch.optim.Adam on the same flat vector; the bf16
    fragment-order copies of W1, W2, W3 and of the transposes W3^T, W2^T are those of the updated weights."""
    import torch
    K = T.karte
    env = T.TarokVecEnv(256, seed=1)
    torch.manual_seed(0)
    P = K.MLP_PARAMS
    flat = (torch.randn(P, device="cuda") * 0.05).contiguous()
    ref = flat.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    m, v = torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    gn = torch.zeros(1, device="cuda")
    bf = lambda k: torch.empty(k, dtype=torch.bfloat16, device="cuda")
    wf = dict(w1=bf(65536), w2=bf(65536), w3=bf(16384), w3t=bf(16384), w2t=bf(65536))
    for it, gscale in enumerate((0.01, 1.0, 1e-4)):           # the middle step clips (norm ~385 > 1)
        g = torch.randn(P, device="cuda") * gscale
        env.learn_adam(flat, g, m, v, step, wf, lr=1e-3, max_norm=1.0, gnorm=gn)
        ref.grad = g.clone()
        n = torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        assert abs(gn.item() - n.item()) <= 1e-4 * n.item()
        assert torch.allclose(flat, ref.detach(), rtol=2e-5, atol=2e-7), it
    assert step.item() == 3
    W1 = flat[K.MLP_W1:K.MLP_B1].view(256, 256); W2 = flat[K.MLP_W2:K.MLP_B2].view(256, 256); W3 = flat[K.MLP_W3:K.MLP_B3].view(64, 256)
    assert torch.equal(wf["w1"], frag_order(W1, 16)) and torch.equal(wf["w2"], frag_order(W2, 16)) and torch.equal(wf["w3"], frag_order(W3, 16))
    assert torch.equal(wf["w1"].view(256, 256), env.mfma_weight_order(W1))                 # (= the rollout kernels' order)
    assert torch.equal(wf["w2t"], frag_order(W2.t().contiguous(), 16))
    assert torch.equal(wf["w3t"], frag_order(W3.t().contiguous(), 4))
    # apply = 0: only the copies
    flat.mul_(0.5)
    env.learn_adam(flat, None, None, None, None, wf, apply=False)
    assert torch.equal(wf["w2t"], frag_order(flat[K.MLP_W2:K.MLP_B2].view(256, 256).t().contiguous(), 16))
    env.close()


def test_learn_returns_vs_torch(T):
    """tarok_learn_returns vs selfplay.assign_returns (the plain torch statement) and torch's mean / std."""
    import torch
    from tarok_amd import selfplay as SP
    n, Tn = 1000, 37
    env = T.TarokVecEnv(n, seed=1)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    done = (torch.rand((Tn, n), device="cuda", generator=g) < 0.08).to(torch.uint8)
    reward = torch.randint(-90, 91, (Tn, n, 4), device="cuda", generator=g, dtype=torch.int16)
    seat = torch.randint(0, 4, (Tn, n), device="cuda", generator=g)
    words = (seat << T.karte.OBS_SEAT_SHIFT) | torch.randint(0, 1 << 50, (Tn, n), device="cuda", generator=g)
    logp = -torch.rand((Tn, n), device="cuda", generator=g)
    val = torch.randn((Tn, n), device="cuda", generator=g)
    act = torch.randint(0, 54, (Tn, n), device="cuda", generator=g, dtype=torch.uint8)
    rec = torch.empty((Tn, n, 4), device="cuda"); stats = torch.empty(4, device="cuda")
    scratch = torch.empty(((n + 255) // 256, 4), device="cuda")
    env.learn_returns(Tn, done, reward, words, logp, val, act, 1.0 / 70.0, rec, stats, scratch)
    ret, known = SP.assign_returns(done.bool(), reward, seat)
    ret = ret / 70.0
    assert torch.equal(rec[..., 0], logp) and torch.equal(rec[..., 2], val)
    assert torch.allclose(rec[..., 1], ret, rtol=1e-6, atol=1e-7)
    bits = rec[..., 3].contiguous().view(torch.int32)
    assert torch.equal((bits & 255).to(torch.uint8), act) and torch.equal(((bits >> 8) & 1).bool(), known)
    adv = (ret - val)[known]
    mean, std = adv.mean().item(), adv.std(unbiased=False).item()
    assert abs(stats[0].item() - mean) < 1e-4 and abs(stats[1].item() - 1.0 / std) < 1e-3 / std
    assert abs(stats[2].item() - known.float().mean().item()) < 1e-6
    env.close()


def _rollout_words(T, env, net_w, steps):
    """feature words [steps * n, 4] and observation words of real positions (policy_mlp on a stepped env)"""
    import torch
    obs = env.reset()
    fws, ows = [], []
    for t in range(steps):
        for _ in range(3):
            obs, _, _ = env.step(env.policy_random(obs), auto_reset=True)
        fw = torch.zeros((env.n, 4), dtype=torch.int64, device="cuda")
        env.policy_mlp(net_w, obs.words, feature_words_out=fw)
        fws.append(fw); ows.append(obs.words.clone())
    return torch.cat(fws), torch.cat(ows)


@pytest.mark.parametrize("B", [5000, 128, 333])
def test_learn_chain_and_dw_vs_torch_autograd(T, B):
    """tarok_learn_chain + tarok_learn_dw on a ragged minibatch (gathered through an index) vs float32 torch
    autograd of the same network on the same bf16 weights, with the kernel's bf16 roundings of H1 / H2 written
    into the reference (a dtype cast is transparent to autograd): activations, loss terms, d loss / d output,
    the hidden gradients and every weight and bias gradient."""
    import torch
    import torch.nn.functional as F
    from tarok_amd import selfplay as SP
    K = T.karte
    n = 2048
    env = T.TarokVecEnv(n, seed=5, mix=K.MIX_ALL)
    torch.manual_seed(1)
    net = SP.PolicyNet(256).cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    ps = [net.fc1.weight, net.fc1.bias, net.fc2.weight, net.fc2.bias, net.head.weight, net.head.bias]
    flat = torch.cat([p.detach().reshape(-1) for p in ps]).contiguous()
    bf = lambda k: torch.empty(k, dtype=torch.bfloat16, device="cuda")
    wf = dict(w1=bf(65536), w2=bf(65536), w3=bf(16384), w3t=bf(16384), w2t=bf(65536))
    env.learn_adam(flat, None, None, None, None, wf, apply=False)
    bias = (flat[K.MLP_B1:K.MLP_B1 + 256], flat[K.MLP_B2:K.MLP_B2 + 256], flat[K.MLP_B3:K.MLP_B3 + 64])
    roll_w = [wf["w1"].view(256, 256), bias[0], wf["w2"].view(256, 256), bias[1], wf["w3"].view(64, 256), bias[2]]
    words, obs_words = _rollout_words(T, env, roll_w, 3)
    M = words.shape[0]
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    idx = torch.randperm(M, device="cuda", generator=g)[:B].contiguous()
    legal_all = SP.legal_matrix(obs_words & K.OBS_MASK)
    assert torch.equal(words[:, 1] & K.OBS_MASK, obs_words & K.OBS_MASK)             # feature word 1 = the legal cards
    act_all = torch.multinomial(legal_all.float(), 1, generator=g).squeeze(1)
    Wq = [p.detach().to(torch.bfloat16).float().requires_grad_(True) if p.dim() == 2 else p.detach().clone().requires_grad_(True) for p in ps]

    def forward(x):
        z1 = x @ Wq[0].T + Wq[1]
        z1.retain_grad()                                       # (the kernel's dH is the gradient at the PRE-activation)
        h1 = torch.relu(z1).to(torch.bfloat16).float()
        z2 = h1 @ Wq[2].T + Wq[3]
        z2.retain_grad()
        h2 = torch.relu(z2).to(torch.bfloat16).float()
        out = h2 @ Wq[4].T + Wq[5]
        out.retain_grad()
        return h1, h2, out, z1, z2
    x = env.expand_feature_words(words[idx], torch.float32)
    h1, h2, out, z1, z2 = forward(x)
    legal = legal_all[idx]
    with torch.no_grad():
        lp_now = F.log_softmax(out[:, :54].masked_fill(~legal, float("-inf")), -1).gather(1, act_all[idx][:, None]).squeeze(1)
    rec = torch.zeros((M, 4), device="cuda")
    rec[idx, 0] = lp_now + 0.4 * torch.randn(B, device="cuda", generator=g)       # ratios on both sides of the clip range
    rec[:, 1] = torch.randn(M, device="cuda", generator=g)
    rec[:, 2] = 0.5 * torch.randn(M, device="cuda", generator=g)
    known = torch.rand(M, device="cuda", generator=g) < 0.8
    rec[:, 3] = (act_all.to(torch.int32) | (known.to(torch.int32) << 8)).view(torch.float32)
    stats = torch.tensor([0.1, 0.9, 0.8, 0.0], device="cuda")
    clip, vf, ent_c = 0.2, 0.5, 0.01
    # ---- reference loss
    r = rec[idx]
    adv = (r[:, 1] - r[:, 2] - stats[0]) * stats[1]
    w = known[idx].float()
    wsum = w.sum().clamp(min=1)
    lg = out[:, :54].masked_fill(~legal, float("-inf"))
    logp_all = F.log_softmax(lg, dim=-1)
    logp = logp_all.gather(-1, act_all[idx][:, None]).squeeze(-1)
    ratio = (logp - r[:, 0]).exp()
    pi = -(torch.min(ratio * adv, ratio.clamp(1 - clip, 1 + clip) * adv) * w).sum() / wsum
    vl = (((out[:, 54] - r[:, 1]) ** 2) * w).sum() / wsum
    p = logp_all.exp()
    H = (-(p * torch.where(legal, logp_all, torch.zeros_like(logp_all))).sum(-1) * w).sum() / wsum
    (pi + vf * vl - ent_c * H).backward()
    # ---- kernels
    act_t = lambda k: torch.zeros((B + K.LEARN_PAD, k), dtype=torch.bfloat16, device="cuda")
    H1, H2, dH2, dH1, dOut = act_t(256), act_t(256), act_t(256), act_t(256), act_t(64)
    scratch = torch.empty(((B + 95) // 96, 4), device="cuda")
    terms = torch.empty(4, device="cuda"); running = torch.zeros(4, device="cuda")
    Xw = torch.zeros((B + K.LEARN_PAD, 4), dtype=torch.int64, device="cuda")
    env.learn_chain(B, words, idx, rec, stats, clip, vf, ent_c, wf, bias, Xw, H1, H2, dOut, dH2, dH1, scratch, terms, running)
    assert torch.equal(Xw[:B], words[idx])                       # the feature words in minibatch order
    # activations: equal to the reference up to a bf16 ulp where the f32 sums round differently
    for got, want, name in ((H1, h1, "H1"), (H2, h2, "H2")):
        d = (got[:B].float() - want.detach()).abs()
        assert d.max().item() <= 0.02 * (1 + want.abs().max().item()), name
        assert (d > 0.004 * (1 + want.detach().abs())).float().mean().item() < 0.01, name
    ref_terms = torch.stack([pi, vl, H]).detach()
    assert torch.allclose(terms[:3], ref_terms, rtol=5e-3, atol=5e-4), (terms, ref_terms)
    assert abs(terms[3].item() - 1.0 / wsum.item()) < 1e-9 and running[3].item() == 1.0
    # gradients w.r.t. the outputs and the hidden activations (kernel: unscaled, i.e. times the weight sum)
    for got, want, name, tol in ((dOut, out.grad, "dOut", 0.01), (dH2, z2.grad, "dH2", 0.02), (dH1, z1.grad, "dH1", 0.03)):
        want = want * wsum
        err = (got[:B].float() - want).abs()
        scale = want.abs().max().item()
        assert err.max().item() < tol * scale + 1e-9, (name, err.max().item(), scale)
    assert (dOut[:B, 55:] == 0).all().item() and (dOut[:B, :54][~legal] == 0).all().item()
    Xw[B:] = -1
    for t_ in (H1, H2, dH2, dH1, dOut):
        t_[B:].uniform_(-3, 3)                                # (the padding rows may hold anything: tarok_learn_dw ignores them)
    # weight and bias gradients
    work = torch.empty(env.learn_workspace_bytes(), dtype=torch.uint8, device="cuda")
    grad = torch.zeros(K.MLP_PARAMS, device="cuda")
    env.learn_dw(B, Xw, H1, H2, dOut, dH2, dH1, terms, work, grad)
    off = 0
    for q, name in zip(Wq, ("W1", "b1", "W2", "b2", "W3", "b3")):
        gk = grad[off:off + q.numel()].view_as(q)
        off += q.numel()
        rel = (gk - q.grad).norm().item() / (q.grad.norm().item() + 1e-12)
        assert rel < 0.02, (name, rel)
        assert (gk - q.grad).abs().max().item() < 0.03 * q.grad.abs().max().item() + 1e-9, name
    env.close()


def test_selfplay_fused_learner_matches_the_torch_update(T):
    """SelfPlay with the fused learner: an iteration runs (graph-captured rollout on the learner's own weight
    copies, fused update), statistics finite, parameters and the rollout's weight copies move together; and ONE
    Adam step from the same rollout, same minibatch, gives the parameters of the torch update (tarok_ppo_loss +
    autograd + clip_grad_norm_ + torch.optim.Adam) up to the bf16 noise in the gradient."""
    import torch
    from tarok_amd import selfplay as SP
    K = T.karte
    n = 4096
    envs = [T.TarokVecEnv(n, seed=9, mix=K.MIX_ALL) for _ in range(2)]
    a = SP.SelfPlay(envs[0], hidden=256, seed=0, fused_learner=True)
    b = SP.SelfPlay(envs[1], hidden=256, seed=0, fused_learner=False)
    assert torch.equal(a.flat, b.flat)
    buf = a.collect(24)
    bufb = b.collect(24)
    for k in ("act", "words", "done", "reward"):
        assert torch.equal(buf[k], bufb[k]), k                                         # same weights, same games, same draws
    p0 = a.flat.clone()
    sa = a.update_fused(buf, epochs=1, minibatches=1)
    sb = b.update(bufb, epochs=1, minibatches=1)
    for k in ("loss", "pi_loss", "v_loss", "entropy"):
        assert np.isfinite(sa[k]) and abs(sa[k] - sb[k]) < 2e-2 * (1 + abs(sb[k])), (k, sa[k], sb[k])
    assert abs(sa["known_frac"] - sb["known_frac"]) < 1e-6
    da, db = a.flat - p0, b.flat - p0
    assert da.abs().max().item() > 0
    # Adam's first step moves every parameter by lr * g / (|g| + eps): the same sign wherever the gradient is not noise
    big = db.abs() > 0.5 * a.lr
    assert big.float().mean().item() > 0.5
    assert (torch.sign(da[big]) == torch.sign(db[big])).float().mean().item() > 0.995
    assert torch.equal(a._wf["w1"].view(256, 256), envs[0].mfma_weight_order(a.flat[:65536].view(256, 256)))
    st = a.iterate(T=24, epochs=2, minibatches=4)                                      # the next rollout reads the updated copies
    assert st["env_errors"] == 0 and np.isfinite(st["loss"]) and st["iteration_steps_per_s"] > 0
    for e in envs:
        e.close()


def test_fused_update_is_reproducible_bit_for_bit(T):
    """Two learners from the same seed on two envs: rollouts, every minibatch's gradient (split-K partial sums added in a
    fixed order, no atomics), Adam and the weight copies give the SAME bits after three iterations — the update has no
    run-to-run freedom (which is also why its tile-to-workgroup assignment is static: profiles/r04_learner_steps.txt (6))."""
    import torch
    from tarok_amd import selfplay as SP
    K = T.karte
    n = 8192
    envs = [T.TarokVecEnv(n, seed=4, mix=K.MIX_ALL) for _ in range(2)]
    sps = [SP.SelfPlay(e, hidden=256, seed=2, fused_learner=True) for e in envs]
    for _ in range(3):
        stats = [sp.iterate(T=24, epochs=1, minibatches=4) for sp in sps]
        assert stats[0]["loss"] == stats[1]["loss"]
    assert torch.equal(sps[0].flat, sps[1].flat) and torch.equal(sps[0].adam_m, sps[1].adam_m) and torch.equal(sps[0].adam_v, sps[1].adam_v)
    for k in ("w1", "w2", "w3", "w3t", "w2t"):
        assert torch.equal(sps[0]._wf[k], sps[1]._wf[k]), k
    for e in envs:
        e.close()
