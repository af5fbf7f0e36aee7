"""GPU: the reference-layout observation (tarok_observe_ref & co) and the rezultat_igre
reward (f3) through the C ABI, against the line-cited restatement oracle/encoder_spec.py on the games
RECORDED FROM THE REFERENCE (tests/golden/traces_v1.npz: deals, contracts, exchanges and every card).

Parity unpinned for the encoder itself (the reference holds no fixture for these tensors and Igralec.py
cannot be imported): what is pinned here is that the device equals the restatement, on inputs that are the
reference's own games.  Bit-exact (0/1 bytes, integers)."""
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


@pytest.fixture(scope="module")
def E():
    from oracle import encoder_spec
    return encoder_spec


@pytest.fixture(scope="module")
def traces(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "traces_v1.npz")))


def sample_games(tr, per_contract=24):
    idx = []
    for c in range(10):
        w = np.where(tr["contract"] == c)[0]
        idx += list(w[:: max(1, len(w) // per_contract)][:per_contract])
    return np.array(sorted(idx))


def reset_from_traces(env, tr, idx, **kw):
    king = np.where(tr["king"][idx] < 0, 0, tr["king"][idx]).astype(np.int8)
    choice = np.where(tr["choice"][idx] < 0, 0, tr["choice"][idx]).astype(np.int8)
    return env.reset(deals=tr["deals"][idx], contract=tr["contract"][idx], declarer=tr["declarer"][idx], king_suit=king,
                     talon_choice=choice, discards=tr["discards"][idx], **kw)


def expected_record(E, tr, i, t):
    """The restatement's record for game i of the traces when its card number t is about to be played."""
    c = int(tr["contract"][i])
    tip = E.TIP_IZBIRE[c]
    me = int(tr["seats"][i, t])
    decl = int(tr["declarer"][i])
    z = E.zgodovina_of(tr["deals"][i], c, tr["choice"][i], tr["seats"][i], tr["actions"][i], t)
    dealt = [int(x) for x in tr["deals"][i][12 * me:12 * me + 12]]
    zal = [int(x) for x in tr["discards"][i][:E.GROUP[c]]] if (c in E.GROUP and me == decl) else None
    king = int(tr["king"][i]) if 1 <= c <= 3 else None
    mozne = [b for b in range(54) if (int(tr["masks"][i, t]) >> b) & 1]
    lists = E.stanje_v_vektor(me, tip, z, dealt, zal, king, decl, mozne)
    rec, T_ = E.pack_record(me, tip, lists)
    return rec, T_, E.TIPI_NN[tip], me


def test_observe_ref_matches_the_restatement_at_every_step_of_reference_games(T, E, traces):
    tr = traces
    idx = sample_games(tr)
    n = len(idx)
    env = T.TarokVecEnv(n, seed=0, history=True)
    reset_from_traces(env, tr, idx)
    for t in range(48):
        rec, meta = env.observe_ref()
        rec, meta = rec.cpu().numpy(), meta.cpu().numpy()
        for j, i in enumerate(idx):
            if tr["nsteps"][i] > t:
                exp, T_, typ, me = expected_record(E, tr, i, t)
                assert meta[j].tolist() == [T_, typ, t, me], (i, t, int(tr["contract"][i]))
                bad = np.nonzero(rec[j] != exp)[0]
                assert bad.size == 0, ("record", i, t, int(tr["contract"][i]), bad[:8].tolist())
            else:
                assert meta[j, 0] == 0 and not rec[j].any(), ("finished game", i, t)
        env.step(tr["actions"][idx, t])
    # the views name the reference's tensors
    v = T.TarokVecEnv.ref_views(env.observe_ref()[0])
    assert tuple(v["input_layer_nasprotiki"].shape) == (n, 56, 3, 54) and tuple(v["talon_input"].shape) == (n, 6, 55)
    env.close()


def test_history_is_recorded_by_multi_card_launches_and_survives_auto_reset(T, E):
    """tarok_krog_random on a history env: the history (and so the reference observation) after k cards equals
    the one-card-per-launch path's, through auto-resets (a new game restarts at row 0)."""
    import torch
    n = 3000
    a = T.TarokVecEnv(n, seed=5, mix=T.karte.MIX_ALL, history=True)
    b = T.TarokVecEnv(n, seed=5, mix=T.karte.MIX_ALL, history=True)
    a.reset(episode=0)
    b.reset(episode=0)
    for cards in (4, 7, 48, 13, 24):
        a.krog_random(cards, auto_reset=True)
        for _ in range(cards):
            b.step_random(auto_reset=True)
        ra, ma = a.observe_ref()
        rb, mb = b.observe_ref()
        assert torch.equal(ma, mb) and torch.equal(ra, rb), cards
        plays = ma[:, 2].cpu().numpy()
        ha, hb = a.get_history().cpu().numpy(), b.get_history().cpu().numpy()
        for p in range(48):
            m = plays > p
            assert (ha[p, m] == hb[p, m]).all(), (cards, p)
    # checkpoint: canonical lanes + history restore the observation in a fresh env
    c = T.TarokVecEnv(n, seed=5, mix=T.karte.MIX_ALL, history=True)
    c.reset(episode=0)
    c.set_state(a.state())
    c.set_history(a.get_history())
    rc, mc = c.observe_ref()
    ra, ma = a.observe_ref()
    assert torch.equal(ra, rc) and torch.equal(ma, mc)
    for e in (a, b, c):
        e.close()


def test_observe_ref_needs_a_history_env(T):
    env = T.TarokVecEnv(64, seed=1)
    env.reset()
    with pytest.raises(T.TarokNativeError):
        env.observe_ref()
    with pytest.raises(T.TarokNativeError):
        env.get_history()
    env.close()


def test_exchange_and_bidding_inputs(T, E, traces):
    tr = traces
    idx = np.where((tr["contract"] >= 1) & (tr["contract"] <= 6))[0][:300]
    idx = np.concatenate([idx, np.where(tr["contract"] == 0)[0][:20], np.where(tr["contract"] == 8)[0][:20]])
    n = len(idx)
    env = T.TarokVecEnv(n, seed=0, history=True)
    reset_from_traces(env, tr, idx, defer_exchange=True)
    ex = env.observe_exchange_ref().cpu().numpy()
    hands = env.observe_hands_ref().cpu().numpy()
    for j, i in enumerate(idx):
        c, d = int(tr["contract"][i]), int(tr["declarer"][i])
        deal = [int(x) for x in tr["deals"][i]]
        for s in range(4):                                               # Igralec.py:278-281
            assert sorted(np.nonzero(hands[j, s])[0].tolist()) == sorted(deal[12 * s:12 * s + 12])
        if c in E.GROUP:
            gs = E.GROUP[c]
            kupcki = [deal[48 + k:48 + k + gs] for k in range(0, 6, gs)]
            roka, talon, igra = E.menjaj_talon_v_vektor(deal[12 * d:12 * d + 12], kupcki, c, tr["king"][i] if c <= 3 else None)
            exp = np.concatenate([roka.reshape(-1), talon.reshape(-1), igra.reshape(-1), np.zeros(7)]).astype(np.uint8)
            assert (ex[j] == exp).all(), (i, c)
        else:
            assert not ex[j].any()
    # while a game waits for the exchange its play observation is empty; afterwards it is the restatement's
    rec, meta = env.observe_ref()
    waiting = np.isin(tr["contract"][idx], list(E.GROUP))
    assert (meta.cpu().numpy()[waiting, 0] == 0).all() and not rec.cpu().numpy()[waiting].any()
    choice = np.where(tr["choice"][idx] < 0, 0, tr["choice"][idx]).astype(np.int8)
    env.exchange(talon_choice=choice, discards=tr["discards"][idx])
    rec, meta = env.observe_ref()
    rec, meta = rec.cpu().numpy(), meta.cpu().numpy()
    for j, i in enumerate(idx):
        exp, T_, typ, me = expected_record(E, tr, i, 0)
        assert meta[j].tolist() == [T_, typ, 0, me] and (rec[j] == exp).all(), i
    env.close()


def test_reward_ref_is_what_rezultat_igre_folds_in(T, E, traces):
    tr = traces
    idx = sample_games(tr, per_contract=40)
    n = len(idx)
    env = T.TarokVecEnv(n, seed=0)
    obs = reset_from_traces(env, tr, idx)
    seen = np.zeros(n, bool)
    for t in range(48):
        obs, reward, done = env.step(tr["actions"][idx, t], reward_ref=True)
        d = done.cpu().numpy().astype(bool)
        rw = reward.cpu().numpy()
        for j in np.nonzero(d)[0]:
            i = idx[j]
            c, decl = int(tr["contract"][i]), int(tr["declarer"][i])
            left = 12 - (t + 1) // 4                                     # cards left in every hand when the game ended
            exp = [E.rezultat_igre_st_tock(int(tr["scores"][i][s]), E.TIP_IZBIRE[c], s == decl, left) for s in range(4)]
            assert rw[j].tolist() == exp, (i, c, t)
            seen[j] = True
    assert seen.all()
    berac = np.isin(tr["contract"][idx], (7, 9))
    assert berac.any() and (tr["nsteps"][idx][berac] < 48).any() and (tr["nsteps"][idx][berac] == 48).any()
    ep, ss = env.counters()                                              # the score sums stay the plain scores
    assert (ss == tr["scores"][idx]).all()
    env.close()


def test_reward_ref_in_multi_card_launches_equals_the_one_card_path(T):
    """TAROK_REWARD_REF through the deferred scoring of tarok_krog_random (finished games are scored later, on
    dense lanes) against tarok_step_random's on-the-spot path: same reward rows, same done rows, same score sums,
    on an all-Berac batch (several finishes per slot and launch, ring drains inside the loop) and a mixed one."""
    import torch
    for mix, n in ((T.karte.MIX_FIXED + 7, 5000), (T.karte.MIX_FIXED + 9, 3000), (T.karte.MIX_ALL, 20000)):
        a = T.TarokVecEnv(n, seed=8, mix=mix)
        b = T.TarokVecEnv(n, seed=8, mix=mix)
        a.reset(episode=0)
        b.reset(episode=0)
        for cards in (48, 64, 20):
            kb = a.krog_random(cards, auto_reset=True, reward_ref=True)
            for c in range(cards):
                _, rw, dn = b.step_random(auto_reset=True, reward_ref=True)
                d = dn.bool()
                assert torch.equal(kb["done"][c], dn), (mix, cards, c)
                assert torch.equal(kb["reward"][c][d], rw[d]), (mix, cards, c)
        ea, sa = a.counters()
        eb, sb = b.counters()
        assert (ea == eb).all() and (sa == sb).all()
        if mix != T.karte.MIX_ALL:                                       # a Berac's defenders never see their 0
            fin = kb["done"].bool()
            vals = kb["reward"][fin].unique().tolist()
            assert set(vals) <= {-90, -70, 70, 90, -20, 20} and 0 not in vals
        a.close()
        b.close()


def test_observe_ref_at_65536_games_round_trips_its_own_fields(T):
    """Full size: the record of every game is consistent with the env's other outputs (legal mask = the
    observation word's, T's rule, one-hot rows, own-row count) — size-independent properties."""
    import torch
    n = 65536
    env = T.TarokVecEnv(n, seed=9, mix=T.karte.MIX_ALL, history=True)
    env.reset(episode=0)
    for cards in (22, 48):
        kb = env.krog_random(cards, auto_reset=True)
        rec, meta = env.observe_ref()
        v = T.TarokVecEnv.ref_views(rec)
        words = kb["obs"][cards - 1]
        legal_bits = ((words.unsqueeze(1) >> torch.arange(54, device=words.device)) & 1).to(torch.uint8)
        assert torch.equal(v["mozne_vec"], legal_bits)
        plays = meta[:, 2].long()
        assert torch.equal(plays, (words >> 56) & 63)
        assert torch.equal(meta[:, 3].long(), (words >> 54) & 3)
        opp = v["input_layer_nasprotiki"].sum(dim=(2, 3)).long()          # [N,56]: 1 for an opponent's play
        own_rows = (v["roka_input"].sum(dim=2) > 0).long()                # [N,56]: 1 for an own play
        rows = torch.arange(56, device=words.device).unsqueeze(0)
        used = (rows < plays.unsqueeze(1)).long()
        assert torch.equal(opp + own_rows, used)                          # every played row belongs to exactly one side
        T_ = meta[:, 0].long()
        assert ((T_ % 8 == 0) & (T_ > plays) & (T_ <= 56)).all()
        # own rows: the k-th own play shows a hand of (dealt 12) - k cards... at least monotone non-increasing
        cnt = v["roka_input"].sum(dim=2).long()
        big = torch.where(own_rows.bool(), cnt, torch.full_like(cnt, 99))
        run_min = torch.cummin(big, dim=1).values
        assert torch.equal(torch.where(own_rows.bool(), run_min, big), big)
    env.close()
