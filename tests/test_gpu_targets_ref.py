"""GPU: the reference agent's transition targets (tarok_targets_ref, SURVEY §8 f row 3: Nevronski_igralec.rezultat_stiha +
rezultat_igre, Igralec.py:387-446) through the C ABI against the line-cited restatement in oracle/encoder_spec.py, on the
games RECORDED FROM THE REFERENCE (tests/golden/traces_v1.npz): the legal masks, the cards, every trick's value and
winner (what the reference's engines told rezultat_stiha) and the final scores are the reference's own.

Parity unpinned for the assembly itself (the reference holds no fixture of these targets and Igralec.py cannot be
imported); pinned for its inputs."""
import os

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


def test_targets_ref_on_reference_games(T, golden_dir):
    import torch
    from oracle import encoder_spec as E
    from test_gpu_observe_ref import reset_from_traces, sample_games
    tr = dict(np.load(os.path.join(golden_dir, "traces_v1.npz")))
    idx = sample_games(tr)
    n = len(idx)
    env = T.TarokVecEnv(n, seed=0)
    obs = reset_from_traces(env, tr, idx)
    Tn = 48
    ob = torch.zeros((Tn, n), dtype=torch.int64, device="cuda"); act = torch.full((Tn, n), 255, dtype=torch.uint8, device="cuda")
    trick = torch.zeros((Tn, n), dtype=torch.int16, device="cuda"); done = torch.zeros((Tn, n), dtype=torch.uint8, device="cuda")
    reward = torch.zeros((Tn, n, 4), dtype=torch.int16, device="cuda")
    for t in range(Tn):
        ob[t] = obs.words
        a = torch.from_numpy(np.where(tr["nsteps"][idx] > t, tr["actions"][idx, t], 255).astype(np.uint8)).cuda()
        act[t] = a
        obs, rw, dn = env.step(a, tricks=True, reward_ref=True)
        trick[t] = env.trick; done[t] = dn
        reward[t] = torch.where(dn.bool()[:, None], rw, torch.zeros_like(rw))
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    next_q = torch.randn((Tn, n), device="cuda", generator=g)
    factor = 0.1
    for nq in (next_q, None):
        dy, meta = env.targets_ref(ob, act, trick, done, reward, nq, factor)
        dy, meta = dy.cpu().numpy(), meta.cpu().numpy()
        nqh = None if nq is None else nq.cpu().numpy()
        checked = 0
        for j, i in enumerate(idx):
            c = int(tr["contract"][i])
            tip = E.TIP_IZBIRE[c]
            ntr = int(tr["nsteps"][i]) // 4
            for s in range(4):
                trans, steps = [], []
                for b in range(ntr):
                    k = [t for t in range(4 * b, 4 * b + 4) if int(tr["seats"][i, t]) == s][0]
                    mozne = np.array([(int(tr["masks"][i, k]) >> cc) & 1 for cc in range(54)])
                    d0 = E.rezultat_stiha_dy(mozne, int(tr["actions"][i, k]), int(tr["trick_value"][i, b]), int(tr["trick_winner"][i, b]) == s)
                    steps.append(k)
                    trans.append([d0, int(tr["actions"][i, k]), None])
                for b in range(ntr - 1):                       # next_Q_max = the agent's value at its next decision (:351,417-418)
                    trans[b][2] = 0.0 if nqh is None else float(nqh[steps[b + 1], j])
                st = E.rezultat_igre_st_tock(int(tr["scores"][i, s]), tip, s == int(tr["declarer"][i]), 12 - ntr)
                want = E.rezultat_igre_dy(trans, st, factor)
                for b in range(12):
                    if b < ntr:
                        assert meta[b, j, s] == (1 | (2 if b == ntr - 1 else 0)), (i, s, b, meta[b, j, s])
                        assert np.allclose(dy[b, j, s], want[b], rtol=1e-6, atol=1e-5), (i, s, b, c)
                        checked += 1
                    else:
                        assert meta[b, j, s] == 0 and not dy[b, j, s].any(), (i, s, b)
        assert checked > 4000
    env.close()


def test_targets_ref_through_auto_resets_and_at_the_rollout_end(T):
    """A rollout of the in-kernel policy with auto-reset (several games per slot): every completed trick gives four
    valid rows; the rows of a game's last trick carry the final reward; the last trick block of the rollout has no
    next decision to bootstrap from unless its game ended there."""
    import torch
    K = T.karte
    n, Tn = 5000, 24
    env = T.TarokVecEnv(n, seed=4, mix=K.MIX_ALL)
    obs = env.reset()
    ob = torch.zeros((Tn, n), dtype=torch.int64, device="cuda"); act = torch.zeros((Tn, n), dtype=torch.uint8, device="cuda")
    trick = torch.zeros((Tn, n), dtype=torch.int16, device="cuda"); done = torch.zeros((Tn, n), dtype=torch.uint8, device="cuda")
    reward = torch.zeros((Tn, n, 4), dtype=torch.int16, device="cuda")
    for t in range(Tn):
        ob[t] = obs.words
        obs, rw, dn = env.step_random(auto_reset=True, tricks=True, reward_ref=True)
        act[t] = env.action; trick[t] = env.trick; done[t] = dn
        reward[t] = torch.where(dn.bool()[:, None], rw, torch.zeros_like(rw))
    dy, meta = env.targets_ref(ob, act, trick, done, reward, None, 0.5)
    assert ((meta & 1) == 1).all().item()                                       # auto-reset: every slot plays every trick
    last = (meta & 2) != 0
    d4 = done.view(Tn // 4, 4, n)[:, 3].bool()                                  # a game can only end on a trick's 4th card
    assert torch.equal(last, d4[:, :, None].expand(-1, -1, 4))
    boot_missing = (meta & 4) != 0
    assert not boot_missing[:-1].any().item() and torch.equal(boot_missing[-1], ~last[-1])
    # the played card's entry: +/- trick value (+ 0.5 x final reward on a last trick); every other entry 0 or -70
    tv = ((trick.view(Tn // 4, 4, n)[:, 3].to(torch.int32) & 0x7FFF) >> 4).float()
    win = (trick.view(Tn // 4, 4, n)[:, 3].to(torch.int32) & 3)
    seats = ((ob >> K.OBS_SEAT_SHIFT) & 3).view(Tn // 4, 4, n)
    cards = act.view(Tn // 4, 4, n).long()
    rw4 = reward.view(Tn // 4, 4, n, 4)[:, 3].float()
    for k in range(4):
        s = seats[:, k]                                                          # [blocks, n] seat of the k-th card
        got = dy.gather(2, s[:, :, None, None].expand(-1, -1, 1, 54)).squeeze(2).gather(2, cards[:, k][:, :, None]).squeeze(2)
        want = torch.where(win == s, tv, -tv) + 0.5 * torch.where(d4, rw4.gather(2, s[:, :, None]).squeeze(2), torch.zeros_like(tv))
        assert torch.allclose(got, want)
    vals = torch.unique(dy)
    assert (-70.0 in vals.tolist()) and (0.0 in vals.tolist())
    env.close()
