"""GPU parity: the HIP path (through the C ABI, via tarok_amd.TarokVecEnv) against
the CPU oracle and against the fixtures produced by running the reference.
Bit-exact: every legal mask, seat, action, score, pile and state lane.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import tarok_amd
    from tarok_amd import _native
    tarok_amd.build()
    assert os.path.exists(_native.LIB_PATH), "libtarokenv.so missing: the HIP path is the product, no fallback"
    return tarok_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def S():
    from oracle import tarok_spec
    return tarok_spec


@pytest.fixture(scope="module")
def traces(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "traces_v1.npz")))


def digest(p):
    h = hashlib.sha256()
    for name in ("nsteps", "seats", "masks", "actions", "scores"):
        h.update(np.ascontiguousarray(p[name]).tobytes())
    return h.hexdigest()


def oracle_games(O, S, tr, idx):
    games = []
    for i in idx:
        g = O.Game(tr["deals"][i], tr["contract"][i], tr["declarer"][i], tr["king"][i])
        if g.g.phase == 1:
            assert g.exchange(tr["choice"][i], tr["discards"][i][: S.N_DISCARD[int(tr["contract"][i])]]) == 0
        games.append(g)
    return games


def test_golden_traces_every_step(T, O, S, traces):
    """All 2880 reference traces replayed on the GPU: masks/seats per step, done,
    scores, final piles/hands vs the reference; full canonical state vs the oracle."""
    tr = traces
    n = len(tr["contract"])
    env = T.TarokVecEnv(n, seed=0, mix=T.karte.MIX_ALL)
    king = np.where(tr["king"] < 0, 0, tr["king"]).astype(np.int8)
    choice = np.where(tr["choice"] < 0, 0, tr["choice"]).astype(np.int8)
    obs = env.reset(deals=tr["deals"], contract=tr["contract"], declarer=tr["declarer"], king_suit=king,
                    talon_choice=choice, discards=tr["discards"])
    sub = list(range(0, n, 7))
    og = oracle_games(O, S, tr, sub)
    st = env.state()
    for j, i in enumerate(sub):
        assert (st[:, i] == og[j].lanes()).all(), ("state after reset", i, int(tr["contract"][i]))
    scores = np.zeros((n, 4), np.int16)
    finished = np.zeros(n, bool)
    for t in range(48):
        live = tr["nsteps"] > t
        m = obs.mask_numpy()
        assert (m[live] == tr["masks"][live, t]).all(), t
        assert (m[~live] == 0).all()
        assert (obs.seat.cpu().numpy()[live] == tr["seats"][live, t]).all(), t
        assert (obs.step.cpu().numpy()[live] == t).all()
        assert not obs.error.any().item()
        obs, reward, done = env.step(tr["actions"][:, t], tricks=True)
        d = done.cpu().numpy().astype(bool)
        assert (d == (tr["nsteps"] == t + 1)).all(), t
        ti = env.trick.cpu().numpy().view(np.uint16).astype(int)
        if t % 4 == 3:      # rezultat_stiha: winner seat and Roka.vrednost_stiha(stih) from the reference run
            k = t // 4
            exp = 0x8000 | (tr["trick_value"][:, k].astype(int) << 4) | tr["trick_winner"][:, k].astype(int)
            assert (ti[live] == exp[live]).all(), t
            assert (ti[~live] == 0).all()
        else:
            assert (ti == 0).all()
        scores[d] = reward.cpu().numpy()[d]
        finished |= d
        assert (obs.done.cpu().numpy() == finished).all()
        st = env.state()
        for j, i in enumerate(sub):
            if tr["nsteps"][i] > t:
                og[j].step(tr["actions"][i, t])
            assert (st[:, i] == og[j].lanes()).all(), ("state", i, t, int(tr["contract"][i]))
    assert finished.all()
    assert (scores == tr["scores"]).all()
    st = env.state()
    assert (st[0:4].T == tr["hands_end"]).all()
    assert (st[4:8].T == tr["piles"]).all()
    env.close()


def test_device_deal_and_setup_match_oracle(T, O, S):
    """reset() with no arrays: sorting-network deal + contract mix + Bot exchange."""
    for mix, n, seed, ep in [(S.MIX_ALL, 4096, 3, 0), (S.MIX_NAVADNA3, 1000, 4, 9), (S.MIX_FIXED + S.SOLO_DVE, 300, 5, 2),
                             (S.MIX_BOT, 4096, 6, 1)]:
        env = T.TarokVecEnv(n, seed=seed, mix=mix, game_offset=17)
        env.reset(episode=ep)
        st = env.state()
        for i in range(n):
            g = O.Game.synth(seed, 17 + i, ep, mix)
            assert (st[:, i] == g.lanes()).all(), (mix, i)
        env.close()


def test_deferred_exchange_flow(T, O, S, traces):
    """reset(defer) -> exchange(): the two-phase form of Navadna_igra.py:60-66."""
    tr = traces
    idx = np.where((tr["contract"] >= 1) & (tr["contract"] <= 6))[0][:512]
    n = len(idx)
    env = T.TarokVecEnv(n, seed=0)
    king = np.where(tr["king"][idx] < 0, 0, tr["king"][idx]).astype(np.int8)
    obs = env.reset(deals=tr["deals"][idx], contract=tr["contract"][idx], declarer=tr["declarer"][idx],
                    king_suit=king, defer_exchange=True)
    assert (obs.mask_numpy() == 0).all()
    st = env.state()
    for j, i in enumerate(idx[:64]):
        g = O.Game(tr["deals"][i], tr["contract"][i], tr["declarer"][i], tr["king"][i])
        assert g.g.phase == 1 and (st[:, j] == g.lanes()).all()
    obs = env.exchange(talon_choice=tr["choice"][idx], discards=tr["discards"][idx])
    assert (obs.mask_numpy() == tr["masks"][idx, 0]).all()
    # bad exchange: discard a card that is not in the hand -> error bit, still waiting
    env.reset(deals=tr["deals"][idx], contract=tr["contract"][idx], declarer=tr["declarer"][idx],
              king_suit=king, defer_exchange=True)
    bad = tr["discards"][idx].copy()
    other = (tr["declarer"][idx].astype(int) + 1) % 4
    bad[:, 0] = tr["deals"][idx][np.arange(n), other * 12]     # a card of another seat's hand
    obs = env.exchange(talon_choice=np.zeros(n, np.int8), discards=bad)
    assert obs.error.all().item() and (obs.mask_numpy() == 0).all()
    env.close()


def test_illegal_action_rejected(T, S):
    n = 512
    env = T.TarokVecEnv(n, seed=9, mix=S.MIX_ALL)
    obs = env.reset()
    before = env.state()
    m = obs.mask_numpy()
    bad = np.array([[i for i in range(54) if not (int(x) >> i) & 1][0] for x in m], np.uint8)
    bad[::3] = 200                                  # out of range ids too
    obs, reward, done = env.step(bad)
    assert obs.error.all().item()
    assert (obs.mask_numpy() == m).all() and not done.any().item()
    after = env.state()
    assert (before[:9] == after[:9]).all()
    assert ((after[9] >> np.uint64(54)) & np.uint64(1)).all()
    assert ((after[9] & ~(np.uint64(1) << np.uint64(54))) == before[9]).all()
    env.close()


@pytest.mark.parametrize("n", [1, 63, 100, 257])
def test_ragged_sizes(T, O, S, n):
    env = T.TarokVecEnv(n, seed=21, mix=S.MIX_ALL)
    out = env.rollout_random(episode=1, trace=True)
    ref = O.rollout(21, 0, n, 1, S.MIX_ALL)
    assert (out["scores"].cpu().numpy() == ref["scores"]).all()
    assert (out["nsteps"].cpu().numpy() == ref["nsteps"]).all()
    obs = env.reset(episode=1)
    for t in range(48):
        a = env.policy_random(obs)
        obs, _, _ = env.step(a, auto_reset=True)
    ar = O.run_autoreset(21, 0, n, S.MIX_ALL, 48, episode0=1)
    assert (env.state() == ar["lanes"]).all()
    env.close()


def test_policy_kernel_forms_agree_with_the_oracle(T, O, S):
    """tarok_policy_random's two kernels — four games per thread from 2^21 games on when the caller's arrays allow
    16-byte loads, one game per thread otherwise (an observation array 8 bytes off a 16-byte boundary, an odd action
    address) — and the partial last workgroup of the first: the cards of the two agree on every slot, and equal the
    oracle's policy_action on the slot's key, step and legal mask (the last three workgroups and one slot in 61)."""
    import torch
    from tarok_amd import _native
    n = (1 << 21) + 5 * 1024 - 120                   # whole workgroups of k_policy_x4 (1,024 games each) and a partial one
    env = T.TarokVecEnv(n, seed=5, mix=S.MIX_ALL)
    obs = env.reset(episode=0)
    for t in range(7):
        obs, _, _ = env.step(env.policy_random(obs))
    words = obs.words.clone()
    w = words.cpu().numpy().view(np.uint64)
    idx = np.unique(np.concatenate([np.arange(0, n, 61), np.arange(n - 3 * 1024, n), np.arange(0, 2048)]))
    want = np.array([O.policy_action(O.game_key(5, i, 0), (int(w[i]) >> T.karte.OBS_STEP_SHIFT) & 63, int(w[i]) & T.karte.OBS_MASK)
                     if int(w[i]) & T.karte.OBS_MASK else 255 for i in idx], np.uint8)           # (a Berac that is over: no card)
    assert (want == 255).any() and (want != 255).sum() > len(idx) // 2
    assert words.data_ptr() % 16 == 0 and env.action.data_ptr() % 2 == 0
    got = env.policy_random(T.Obs(words)).cpu().numpy().copy()          # k_policy_x4
    assert (got[idx] == want).all()
    shifted = torch.zeros(n + 1, dtype=words.dtype, device=words.device)
    shifted[1:] = words
    abuf = torch.zeros(n + 1, dtype=torch.uint8, device=words.device)
    for wv, av in ((shifted[1:], abuf[:n]), (words, abuf[1:]), (shifted[1:], abuf[1:])):     # k_policy
        assert (wv.data_ptr() % 16 != 0) or (av.data_ptr() % 2 != 0)
        abuf.zero_()
        _native.check(env.L.tarok_policy_random(env._h, env._p(wv), env._p(av), env._stream()))
        assert (av.cpu().numpy() == got).all()
    env.close()


def gpu_rollout_arrays(env, episode):
    out = env.rollout_random(episode=episode, trace=True)
    return dict(nsteps=out["nsteps"].cpu().numpy(), scores=out["scores"].cpu().numpy(),
                seats=np.ascontiguousarray(out["seats"].cpu().numpy().T),
                masks=np.ascontiguousarray(out["masks"].cpu().numpy().T).view(np.uint64),
                actions=np.ascontiguousarray(out["actions"].cpu().numpy().T))


def test_fused_rollout_digests_match_reference(T, golden_dir):
    """BASELINE configs 2 and 3 at full size: the fused rollout kernel's
    masks/seats/actions/scores hash to the digest of the REFERENCE engine's run."""
    with open(os.path.join(golden_dir, "digests_v1.json")) as f:
        runs = json.load(f)["synthetic"]
    for r in runs:
        env = T.TarokVecEnv(r["n"], seed=r["seed"], mix=r["mix"])
        got = gpu_rollout_arrays(env, r["episode"])
        assert int(got["nsteps"].sum()) == r["total_steps"], r["name"]
        assert digest(got) == r["sha256"], r["name"]
        env.close()


def test_step_api_full_size_vs_oracle(T, O, S):
    """65,536 mixed games through reset / policy kernel / step kernel, every step
    compared with the oracle's trace (which is digest-pinned to the reference)."""
    n, seed = 65536, 0
    ref = O.rollout(seed, 0, n, 0, S.MIX_ALL, threads=8)
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL)
    obs = env.reset()
    scores = np.zeros((n, 4), np.int16)
    nsteps = np.zeros(n, np.int16)
    for t in range(48):
        live = ref["nsteps"] > t
        assert (obs.mask_numpy() == ref["masks"][:, t]).all(), t
        assert (obs.seat.cpu().numpy()[live] == ref["seats"][live, t]).all(), t
        a = env.policy_random(obs)
        assert (a.cpu().numpy() == ref["actions"][:, t]).all(), t
        obs, reward, done = env.step(a)
        d = done.cpu().numpy().astype(bool)
        scores[d] = reward.cpu().numpy()[d]
        nsteps[d] = t + 1
    assert (nsteps == ref["nsteps"]).all() and (scores == ref["scores"]).all()
    ep, ss = env.counters()
    assert (ss == ref["scores"]).all() and (ep == 0).all()
    # size-independent properties of the scores (Klop.py:36-45, Berac.py:16-18, Navadna_igra.py:100-108)
    st = env.state()
    contract = ((st[9] >> np.uint64(33)) & np.uint64(15)).astype(int)
    sc = scores.astype(int)
    klop = sc[contract == 0]
    assert ((klop <= 0).all(axis=1)).all() and (klop >= -35).all()
    assert set(np.unique(np.abs(sc[contract == 7]))) <= {0, 70} and set(np.unique(np.abs(sc[contract == 9]))) <= {0, 90}
    nav = sc[(contract >= 1) & (contract <= 6) | (contract == 8)]
    assert (nav % 5 == 0).all() and ((nav != 0).sum(axis=1) <= 2).all()
    env.close()


def test_fused_step_equals_two_kernel_path(T, S):
    n = 8192
    a = T.TarokVecEnv(n, seed=5, mix=S.MIX_ALL)
    b = T.TarokVecEnv(n, seed=5, mix=S.MIX_ALL)
    oa, ob = a.reset(), b.reset()
    for t in range(60):
        act = a.policy_random(oa).clone()
        oa, ra, da = a.step(act, auto_reset=True)
        ob, rb, db = b.step_random(auto_reset=True)
        assert (b.action == act).all().item(), t
        assert (oa.words == ob.words).all().item(), t
        assert (da == db).all().item()
    assert (a.state() == b.state()).all()
    a.close(); b.close()


def test_auto_reset_run_vs_oracle_full_size(T, O, S):
    """tarok_run_random with auto-reset (wave-cooperative re-deal inside the step
    kernel), graph-replayed and eager, at 65,536 games x 192 steps vs the oracle."""
    n, seed, steps = 65536, 2, 192
    ref = O.run_autoreset(seed, 0, n, S.MIX_ALL, steps, threads=8)
    # prefetch 4: every finished game is a 32-byte swap from the prefetched buffer;
    # 0: every one is dealt by the wave inside the step kernel; 16: a mixture.
    # cards per launch: 0 = policy + step kernels, 1 = fused one-card kernel, >= 2 = tarok_krog_random
    # fan: refill lists per refill workgroup (tarok_set_option; None = the default for the size, 1)
    # (the bench's own launch shape, 128 cards x 65,536 games, is test_bench_launch_shape_vs_oracle)
    # lazy: the one-card step's emptied lines dealt in bulk every thirty-second launch (tarok_set_option; the default below 2^20
    # games) or in the launch after (0)
    for cards, chunk, pf, fan, *lazy in [(0, 48, 4, None), (1, 0, 0, None), (1, 64, 16, None), (0, 0, 2, None),
                                  (0, 48, 0, 8), (1, 64, 0, 3), (1, 0, 16, 8), (0, 96, 0, 4),
                                  (0, 48, 0, 8, 0), (1, 64, 0, 3, 0), (1, 0, 16, None, 0), (0, 96, 4, 4, 0), (1, 0, 0, 2, 0),
                                  (4, 48, 8, None), (4, 0, 4, None), (8, 96, 8, None), (16, 192, 16, None), (3, 48, 12, None), (48, 192, 48, None),
                                  (4, 64, 0, None), (8, 64, 0, 5), (12, 48, 0, None),
                                  (24, 192, 0, None), (24, 96, 0, None), (32, 192, 0, None), (48, 192, 0, None), (48, 96, 0, None)]:
        env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL, refill_fan=fan, lazy_refill=lazy[0] if lazy else None)
        env.reset()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True, prefetch_every=pf)
        ep, ss = env.counters()
        cfg = (cards, chunk, pf, fan, lazy)
        assert (ep == ref["episode"]).all(), cfg
        assert (ss == ref["score_sum"]).all(), cfg
        assert (env.state() == ref["lanes"]).all(), cfg
        assert (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all(), cfg
        env.close()


def test_step_api_streaming_size_vs_oracle(T, O, S):
    """The one-card step at a batch size that streams (2^20 games; eight refill lists per refill workgroup, spread
    among the play workgroups): the two-kernel external-policy path and tarok_step_random,
    graph-replayed, 96 lock-steps with auto-reset vs the oracle — every slot's episode number, score sums,
    canonical state and observation word."""
    n, seed, steps = 1 << 20, 9, 96
    ref = O.run_autoreset(seed, 0, n, S.MIX_ALL, steps, threads=16)
    for cards, lazy in ((0, None), (1, None), (1, 1)):         # (lazy: bulk deals every thirty-second launch — off by default at this size)
        env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL, lazy_refill=lazy)
        env.reset()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=48, auto_reset=True)
        ep, ss = env.counters()
        assert (ep == ref["episode"]).all(), (cards, lazy)
        assert (ss == ref["score_sum"]).all(), (cards, lazy)
        assert (env.state() == ref["lanes"]).all(), (cards, lazy)
        assert (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all(), (cards, lazy)
        env.close()


def test_bench_launch_shape_vs_oracle(T, O, S):
    """The launch shape bench.py times, exactly: 65,536 games (seed 0, MIX_ALL), 128 cards per launch, ONE
    hipGraph of 20 launches through tarok_run_random (action / done / obs rows written, no trick rows: the
    trick-aligned STD card loop of k_play_wide), captured + replayed, then replayed again (the bench's timed
    region replays the graph its warm-up captured): episode numbers, score sums, canonical state and
    observation words of all 65,536 slots equal the oracle's after 2,560 and after 5,120 lock-steps."""
    n, seed, cards, launches = 65536, 0, 128, 20
    steps = cards * launches
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL)
    env.reset()
    for k in (1, 2):
        env.run_random(steps, cards_per_launch=cards, graph_chunk=steps, auto_reset=True)
        ref = O.run_autoreset(seed, 0, n, S.MIX_ALL, k * steps, threads=16)
        ep, ss = env.counters()
        assert (ep == ref["episode"]).all(), k
        assert (ss == ref["score_sum"]).all(), k
        assert (env.state() == ref["lanes"]).all(), k
        assert (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all(), k
    env.close()


@pytest.mark.parametrize("cards,tricks", [(128, False), (192, False), (128, True)])
def test_long_launch_rows_vs_oracle_trace(T, O, S, cards, tricks):
    """EVERY per-card output row of long launches (128 = the bench's, 192 = the longest) at 65,536 games: two
    consecutive tarok_krog_random launches after a lead-in launch; for a 1-in-64 sample of the slots the
    oracle replays the slot from the env state at the first launch's start, through the auto-resets (next
    game = episode + 1 of the slot), and row c of action / obs / done / reward (/ trick) must be what the
    oracle's c-th card gives: the Bot policy's card on the spec RNG, the next observation word, the done
    flag, the scores of a game that ends, what rezultat_stiha is told.  tricks = False is the set of outputs
    tarok_run_random asks for (the card loop the bench times)."""
    n, seed = 65536, 3
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL)
    env.reset()
    env.run_random(cards, cards_per_launch=cards, graph_chunk=0, auto_reset=True)      # lead-in: slots mid-run, lines consumed
    lanes0 = env.state()
    ep0, ss0 = env.counters()
    rows = dict(action=[], obs=[], done=[], reward=[], trick=[])
    for launch in range(2):
        kb = env.krog_random(cards, auto_reset=True, tricks=tricks)
        for k in rows:
            rows[k].append(kb[k].cpu().numpy().copy())
    act = np.concatenate(rows["action"]); obs = np.concatenate(rows["obs"]).view(np.uint64)
    done = np.concatenate(rows["done"]).astype(bool); reward = np.concatenate(rows["reward"])
    trick = np.concatenate(rows["trick"]).view(np.uint16)
    ep1, ss1 = env.counters()
    lanes1 = env.state()
    for i in range(0, n, 64):
        g = O.Game.from_lanes(lanes0[:, i])
        ep = int(ep0[i])
        key = O.game_key(seed, i, ep)
        ssum = ss0[i].astype(np.int64).copy()
        for t in range(2 * cards):
            a = O.policy_action(key, g.g.trick_no * 4 + g.g.n_in_trick, g.legal())
            assert int(act[t, i]) == a, (i, t)
            assert g.step(a) >= 0, (i, t)
            fin = g.done
            assert bool(done[t, i]) == fin, (i, t)
            if tricks:
                assert int(trick[t, i]) == int(g.g.last_trick), (i, t)
            if fin:
                assert reward[t, i].tolist() == g.scores, (i, t)
                ssum += np.array(g.scores)
                ep += 1
                g = O.Game.synth(seed, i, ep, S.MIX_ALL)
                key = O.game_key(seed, i, ep)
            assert g.obs_word(fin) == int(obs[t, i]), (i, t)
        assert ep == int(ep1[i]) and (ssum == ss1[i]).all(), i
        assert (g.lanes() == lanes1[:, i]).all(), i
    env.close()


def test_long_launches_vs_oracle(T, O, S):
    """The bench's launch length and beyond: 128 / 160 / 192 cards per launch (a slot finishes up to a dozen games
    within one launch: early line top-ups, the fast renewal of the trick-aligned loop, finishing lanes WITHOUT a
    line — dealt on the spot into the line registers — and, at 192, slots that run out of dealt-ahead lines), with
    an extra prefetch after every launch and without (the lines then come from the refill workgroups alone), 16,384 games x 1,920 lock-steps vs the
    oracle: episode numbers, score sums, canonical state, observation words."""
    n, seed, steps = 16384, 5, 1920
    ref = O.run_autoreset(seed, 0, n, S.MIX_ALL, steps, threads=8)
    for cards, chunk, pf in [(128, 640, 128), (128, 0, 0), (160, 960, 0), (192, 1920, 0), (64, 640, 64)]:
        env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL)
        env.reset()
        env.run_random(steps, cards_per_launch=cards, graph_chunk=chunk, auto_reset=True, prefetch_every=pf)
        ep, ss = env.counters()
        cfg = (cards, chunk, pf)
        assert (ep == ref["episode"]).all(), cfg
        assert (ss == ref["score_sum"]).all(), cfg
        assert (env.state() == ref["lanes"]).all(), cfg
        assert (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all(), cfg
        env.close()


def test_krog_kernel_writes_what_four_single_steps_write(T, S):
    """tarok_krog_random(cards): row c of every output equals what the c-th tarok_step_random
    call writes (actions, observation words, done, per-trick info, scores), with and without
    auto-reset, from aligned and mid-trick starts."""
    # (the last case: more workgroups than fit on the chip at once and an N that puts workgroup
    #  boundaries inside cache lines of the byte-sized output rows)
    for auto, cards, lead_in, n in [(True, 4, 0, 16384), (True, 4, 2, 16384), (False, 4, 0, 16384), (True, 8, 1, 16384),
                                    (True, 5, 0, 16384), (True, 24, 0, 16384), (True, 24, 3, 16384), (True, 48, 0, 16384), (True, 48, 1, 300007)]:
        a = T.TarokVecEnv(n, seed=19, mix=S.MIX_ALL)
        b = T.TarokVecEnv(n, seed=19, mix=S.MIX_ALL)
        a.reset(); b.reset()
        for t in range(lead_in):
            a.step_random(auto_reset=auto); b.step_random(auto_reset=auto)
        for rounds in range(14 if cards < 24 else (6 if cards < 48 else 3)):
            kb = a.krog_random(cards, auto_reset=auto)
            for c in range(cards):
                ob, rw, dn = b.step_random(auto_reset=auto, tricks=True)
                assert (kb["action"][c] == b.action).all().item(), (auto, cards, rounds, c)
                assert (kb["obs"][c] == ob.words).all().item(), (auto, cards, rounds, c)
                assert (kb["done"][c] == dn).all().item()
                assert (kb["trick"][c] == b.trick).all().item()
                d = dn.bool()
                assert (kb["reward"][c][d] == rw[d]).all().item()
        assert (a.state() == b.state()).all()
        ea, sa = a.counters(); eb, sb = b.counters()
        assert (ea == eb).all() and (sa == sb).all()
        a.close(); b.close()


def test_auto_reset_revives_games_finished_earlier(T, O, S):
    """Games finished WITHOUT auto-reset are replaced by the first auto-reset step."""
    n, seed = 2048, 31
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_FIXED + S.BERAC)
    obs = env.reset()
    for t in range(48):
        obs, _, _ = env.step(env.policy_random(obs))
    assert obs.done.all().item()
    obs, _, done = env.step(env.policy_random(obs), auto_reset=True)    # no card played, new games swapped in
    assert not done.any().item() and not obs.done.any().item()
    st = env.state()
    for i in range(0, n, 37):
        assert (st[:, i] == O.Game.synth(seed, i, 1, S.MIX_FIXED + S.BERAC).lanes()).all()
    ep, _ = env.counters()
    assert (ep == 1).all()
    env.close()


def test_sharded_envs_play_the_same_games(T, O, S):
    """Two half-size envs with game_offset = what two ranks would hold."""
    n, seed = 4096, 8
    whole = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL)
    w = whole.rollout_random(episode=0)
    parts = []
    for r in range(2):
        e = T.TarokVecEnv(n // 2, seed=seed, mix=S.MIX_ALL, game_offset=r * n // 2)
        parts.append(e.rollout_random(episode=0)["scores"].cpu().numpy())
        e.close()
    assert (np.concatenate(parts) == w["scores"].cpu().numpy()).all()
    whole.close()


class _SpecPlayer:
    """Duck-typed reference-shaped player (NOT an Igralec subclass): plays the spec's
    deterministic policy keyed by game id — the same one gen_golden.py's ParalelCtl drove
    the reference's Tarok.paralel_start with."""

    def __init__(self, ime, S, seed, mix, shared):
        self.ime, self.S, self.seed, self.mix, self.shared = str(ime), S, seed, mix, shared
        self.roka, self.kupcek, self.seat = {}, {}, {}

    def _key(self, g):
        return self.S.game_key(self.seed, g, 0)

    def _setup(self, g):
        S = self.S
        c, d, k = S.sample_setup(self._key(g), self.mix)
        if not (d == 0 or c >= S.DVE) or c == S.KLOP:
            d = 0
        return c, d, k

    def nova_igra(self, roka, igralci, g):
        self.roka[g], self.kupcek[g] = roka, []
        self.seat[g] = [i for i, p in enumerate(igralci) if p is self][0]
        self.shared.setdefault("steps", {})[g] = 0

    def licitiram(self, min_igra, g, obvezno=None, prednost=False):
        from tarok_amd import licitacija as L
        c, d, k = self._setup(g)
        want = c * 10 if (self.seat[g] == d and c != self.S.KLOP) else L.NAPREJ
        return L.base_filter(want, min_igra, obvezno, prednost)

    def izberi_barvo_kralja(self, g):
        return self._setup(g)[2]

    def konec_licitiranja(self, *a):
        pass

    def menjaj_iz_talona(self, kupcki, st_kart, g):
        from tarok_amd import igralec as I, karte as K
        self.roka[g].dodaj_karte(kupcki[0])
        hand = K.ids_to_mask(k.v_id() for k in self.roka[g])
        for cid in self.S.bot_discards(self._key(g), hand, st_kart):
            k = I.Karta.iz_id(cid)
            self.kupcek[g].append(k)
            self.roka[g].igraj_karto(k)
        return 0

    def pripravi_igraj_karto(self, stih, mozne, zgodovina, g):
        pass

    def igraj_karto(self, stih, mozne, zgodovina, g):
        from tarok_amd import karte as K
        t = self.shared["steps"][g]
        self.shared["steps"][g] = t + 1
        cid = self.S.policy_action(self._key(g), t, K.ids_to_mask(k.v_id() for k in mozne))
        karta = [k for k in mozne if k.v_id() == cid][0]
        self.roka[g].igraj_karto(karta)
        return karta

    def rezultat_stiha(self, stih, sem_pobral, g):
        if sem_pobral:
            self.shared.setdefault("tricks", {}).setdefault(g, []).append(len(stih))

    def rezultat_igre(self, pts, zgodovina, g):
        self.shared.setdefault("scores", {}).setdefault(g, {})[self.ime] = int(pts)


# the player protocol's callbacks (Igralec.py:32-122) and where each carries the game id (oracle/gen_golden.py logs
# the reference's run with the same table)
_CALLBACKS = ["nova_igra", "pripavi_licitiram", "predict_licitiram", "licitiram", "izberi_barvo_kralja", "konec_licitiranja",
              "pripravi_izbral_iz_talona", "predict_izberi_iz_talona", "menjaj_iz_talona", "izbral_iz_talona",
              "poglej_karte_odprtega_beraca", "pripravi_igraj_karto", "predict_igraj_karto", "igraj_karto", "rezultat_stiha",
              "rezultat_igre"]
_ID_ARG = [2, 0, -1, 1, 0, 2, 2, -1, 2, 2, 1, 3, -1, 3, 2, 2]


def _logged(cls, log):
    """cls with every protocol callback logging (name, game id or -1, player) before it runs; callbacks the class
    does not define are added as no-ops, so the log holds every call the scheduler makes."""
    def wrap(name, k):
        inner = getattr(cls, name, None)

        def f(self, *a, **kw):
            log.append((name, -1 if _ID_ARG[k] < 0 else int(a[_ID_ARG[k]]), int(self.ime)))
            return inner(self, *a, **kw) if inner is not None else None
        return f
    return type("Logged" + cls.__name__, (cls,), {name: wrap(name, k) for k, name in enumerate(_CALLBACKS)})


def _per_game(log, g):
    """The calls that concern game g — its own callbacks as (name, seat-independent player) and the batch barriers as
    (name, -1), one entry per barrier (the four players' predict_* calls collapsed) — up to its rezultat_igre."""
    seq = []
    for name, gid, p in log:
        if gid == g:
            seq.append((name, p))
        elif gid == -1 and (not seq or seq[-1] != (name, -1)):
            seq.append((name, -1))
    last = max(k for k, x in enumerate(seq) if x[0] == "rezultat_igre")
    return seq[:last + 1]


def test_paralel_start_adapter_matches_reference_run(T, S, golden_dir):
    """tarok_amd.igralec.Tarok.paralel_start (callbacks served from the GPU env) against the
    reference's own Tarok.paralel_start on the same deals, bids and card choices."""
    from tarok_amd import igralec as I
    ref = dict(np.load(os.path.join(golden_dir, "paralel_v1.npz")))
    n, seed, mix = len(ref["deals"]), int(ref["seed"]), int(ref["mix"])
    shared = {}
    log = []
    players = [_logged(_SpecPlayer, log)(i, S, seed, mix, shared) for i in range(4)]
    t = I.Tarok(players, n, seed=seed)
    rez = t.paralel_start()
    assert [rez[p] for p in players] == [int(x) for x in ref["totals"]]
    for g in range(n):
        assert [shared["scores"][g][str(i)] for i in range(4)] == [int(x) for x in ref["per_game_scores"][g]], g
        d, c, _ = t.zadnje_igre[g]
        assert (c // 10, d) == (int(ref["setup"][g][0]), int(ref["setup"][g][1])), g
        if c == 0:      # Klop: tricks 1-6 carry the talon card (Klop.py:67-71)
            assert shared["tricks"][g] == [5] * 6 + [4] * 6
    # ---- the ORDER of the callbacks, against the log recorded from the reference's own scheduler and
    # engines (gen_golden.py: every protocol callback of Tarok.paralel_start's run, in call order)
    names = [str(x) for x in ref["call_names"]]
    assert names == _CALLBACKS
    ref_log = [(names[k], int(g), int(p)) for k, g, p in ref["calls"]]
    assert len(log) > 10000 and sorted(ref_log) == sorted(log)          # the same calls were made ...
    solo_brez = [g for g in range(n) if int(ref["setup"][g][0]) == 8]
    assert solo_brez, "the fixture holds Solo_brez games"
    for g in range(n):                                                   # ... and, game by game, in the same order
        mine, theirs = _per_game(log, g), _per_game(ref_log, g)
        if g in solo_brez:
            # the one expected difference (Navadna_igra.py:48,67-68: no 'Pripravljen menjat' yield, the game runs
            # one next() ahead of Tarok.paralel_start): same callbacks in the same order, one barrier early
            strip = lambda seq: [x for x in seq if x[1] >= 0]
            assert strip(mine) == strip(theirs), g
            first = lambda seq: next(k for k, x in enumerate(seq) if x[0] == "pripravi_igraj_karto")
            barrier = ("predict_izberi_iz_talona", -1)
            assert barrier in theirs[first(theirs):] and barrier not in mine[first(mine):] and barrier in mine[:first(mine)]
        else:
            assert mine == theirs, (g, int(ref["setup"][g][0]), next((k, a, b) for k, (a, b) in enumerate(zip(mine, theirs)) if a != b) if len(mine) == len(theirs) else (len(mine), len(theirs)))


def test_paralel_start_with_bot_players(T):
    """Reference-shaped random players through the adapter: runs, scores are well-formed."""
    import random
    from tarok_amd import igralec as I
    bots = [I.Bot_igralec(i, rng=random.Random(100 + i)) for i in range(4)]
    t = I.Tarok(bots, 64, seed=3)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rez = t.paralel_start()
    assert len(t.zadnje_igre) == 64
    for d, c, sc in t.zadnje_igre:
        assert c in (0, 10, 20, 30)             # the Bot only bids Tri/Dve/Ena (Igralec.py:151)
        assert all(len(b.roka[g]) == 0 for g, b in [(0, bots[0])])
    assert sum(rez.values()) == sum(sum(sc) for _, _, sc in t.zadnje_igre)


def test_observe_features_match_python_restatement(T, S):
    """tarok_observe's [N,256] bf16 rows against a plain restatement from the canonical state."""
    n = 3000
    env = T.TarokVecEnv(n, seed=12, mix=S.MIX_ALL)
    obs = env.reset()
    for t in range(13):
        obs, _, _ = env.step(env.policy_random(obs), auto_reset=(t % 2 == 0))
    feat = env.observe().float().cpu().numpy()
    assert feat.shape == (n, 256) and set(np.unique(feat)) <= {0.0, 1.0}
    st = env.state()
    mask = obs.mask_numpy()
    for i in range(0, n, 11):
        meta = int(st[9, i])
        trick = [(meta >> (6 * j)) & 63 for j in range(4)]
        nt, leader, trick_no = (meta >> 24) & 7, (meta >> 27) & 3, (meta >> 29) & 15
        contract, decl, king, team, phase = (meta >> 33) & 15, (meta >> 37) & 3, (meta >> 39) & 7, (meta >> 42) & 15, (meta >> 52) & 3
        seat = (leader + nt) & 3
        live = phase == 2
        exp = np.zeros(256)

        def put(base, m):
            for b in range(64):
                if (int(m) >> b) & 1:
                    exp[base + b] = 1
        put(0, int(st[seat, i]) | (1 << (54 + contract)))
        put(64, int(mask[i]) | ((1 << ((decl - seat) & 3)) | (1 << (4 + nt)) | (((team >> seat) & 1) << 8) | ((1 if king != 7 else 0) << 9)) << 54)
        table = sum(1 << trick[j] for j in range(nt))
        put(128, table | (((1 << king) if king != 7 else 0) | (trick_no << 4)) << 54)
        put(192, (int(st[4, i]) | int(st[5, i]) | int(st[6, i]) | int(st[7, i])) | ((1 if live else 0) << 54))
        assert (feat[i] == exp).all(), i
    env.close()


def test_selfplay_iteration_runs_and_learns_something_finite(T, S):
    import torch
    from tarok_amd import selfplay as SP
    env = T.TarokVecEnv(4096, seed=1, mix=S.MIX_ALL)
    for hidden in (128, 256):                       # torch GEMM path / fused MFMA kernel path
        sp = SP.SelfPlay(env, hidden=hidden, seed=0)
        assert sp.fused == (hidden == 256)
        before = [p.detach().clone() for p in sp.net.parameters()]
        st = sp.iterate(T=48, epochs=1, minibatches=4)
        assert np.isfinite(st["loss"]) and st["known_frac"] > 0.3 and st["rollout_steps_per_s"] > 0
        assert st["env_errors"] == 0
        assert any((a != b).any().item() for a, b in zip(before, sp.net.parameters()))
        assert not env.legal_actions().error.any().item()      # the sampled cards were always legal
        st = sp.iterate(T=48, epochs=1, minibatches=4)           # graph replay
        assert np.isfinite(st["loss"]) and st["env_errors"] == 0
    env.close()


def test_selfplay_65536_envs_vs_oracle_replay(T, O, S):
    """BASELINE config 4 at its size: SelfPlay on 65,536 envs with the fused MFMA policy, two iterations
    (graph capture, then replay).  The second rollout's recorded cards are replayed on the CPU oracle for a
    1-in-64 sample of the slots, from the env state at the rollout's start: every sampled card is legal,
    every observation word, done flag and score equals the oracle's, through the auto-resets (next game =
    episode + 1 of the slot), and the env's episode numbers / score sums moved by exactly that."""
    import torch
    from tarok_amd import selfplay as SP
    n, seed, Tn = 65536, 21, 48
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL)
    sp = SP.SelfPlay(env, hidden=256, seed=0)
    st = sp.iterate(T=Tn, epochs=1, minibatches=8)               # warm-up launches + graph capture + first replay
    assert st["env_errors"] == 0 and np.isfinite(st["loss"])
    torch.cuda.synchronize()
    lanes0 = env.state()
    ep0, ss0 = env.counters()
    words0 = sp.obs_words.cpu().numpy().view(np.uint64)
    st = sp.iterate(T=Tn, epochs=1, minibatches=8)               # pure graph replay
    assert st["env_errors"] == 0 and np.isfinite(st["loss"]) and st["rollout_steps_per_s"] > 0
    buf = sp._buf
    act = buf["act"].cpu().numpy()
    done = buf["done"].cpu().numpy().astype(bool)
    reward = buf["reward"].cpu().numpy()
    words = buf["words"].cpu().numpy().view(np.uint64)
    ep1, ss1 = env.counters()
    lanes1 = env.state()
    assert (words[0] == words0).all()
    legal_ok = ((words[:Tn] >> act.astype(np.uint64)) & np.uint64(1)).astype(bool)
    assert legal_ok.all()                                        # all 3.1 M sampled cards lie in their legal masks
    for i in range(0, n, 64):
        g = O.Game.from_lanes(lanes0[:, i])
        ep = int(ep0[i])
        ssum = ss0[i].astype(np.int64).copy()
        assert g.obs_word() == int(words[0, i]) & ~(1 << 62), i
        for t in range(Tn):
            assert g.step(int(act[t, i])) >= 0, ("illegal card", i, t)
            fin = g.done
            assert bool(done[t, i]) == fin, (i, t)
            if fin:
                assert reward[t, i].tolist() == g.scores, (i, t)
                ssum += np.array(g.scores)
                ep += 1
                g = O.Game.synth(seed, i, ep, S.MIX_ALL)
            assert g.obs_word(fin) == int(words[t + 1, i]), (i, t)
        assert ep == int(ep1[i]) and (ssum == ss1[i]).all(), i
        assert (g.lanes() == lanes1[:, i]).all(), i
    del sp
    env.close()


def test_policy_step_kernel_equals_policy_then_step(T, S):
    """tarok_policy_step (policy MLP + sampling + env step in one launch) vs tarok_policy_mlp followed
    by tarok_step: every rollout buffer, the counters and the final env state are identical; the
    ragged last workgroup and games finishing (auto-reset, refill lists) included."""
    import torch
    from tarok_amd import selfplay as SP
    n = 20000 + 77
    res = []
    for fused_step in (False, True):
        env = T.TarokVecEnv(n, seed=41, mix=S.MIX_ALL, game_offset=5)
        sp = SP.SelfPlay(env, hidden=256, seed=0, fused_step=fused_step)
        for it in range(2):                               # the second collect replays the captured graph
            b = sp.collect(40)
        torch.cuda.synchronize()
        ep, ss = env.counters()
        res.append(({k: v.clone() for k, v in b.items()}, ep.copy(), ss.copy(), env.state().copy()))
        del sp
        env.close()
    (b0, ep0, ss0, st0), (b1, ep1, ss1, st1) = res
    for k in b0:
        if k == "reward":
            d = b0["done"].bool()
            assert (b0[k][d] == b1[k][d]).all().item(), k
        else:
            assert (b0[k] == b1[k]).all().item(), k
    assert (ep0 == ep1).all() and (ss0 == ss1).all() and (st0 == st1).all()
    assert ep0.sum() > n                                  # games did finish and were replaced


def test_ppo_loss_kernel_vs_torch(T, S):
    """tarok_ppo_loss (clipped surrogate + value loss - entropy over the legal cards, forward and
    gradient in one pass) vs the same loss written with torch ops and differentiated by autograd."""
    import torch
    import torch.nn.functional as F
    from tarok_amd import selfplay as SP
    n = 20000                                         # not a multiple of 256
    env = T.TarokVecEnv(n, seed=3, mix=S.MIX_ALL)
    obs = env.reset()
    for t in range(5):
        obs, _, _ = env.step(env.policy_random(obs), auto_reset=True)
    words = obs.words.clone()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    out = (torch.randn((n, 64), device="cuda", generator=g) * 2).to(torch.bfloat16)
    legal = SP.legal_matrix(words & T.karte.OBS_MASK)
    act = torch.multinomial(legal.float(), 1, generator=g).squeeze(1)
    with torch.no_grad():
        lp_now = F.log_softmax(out[:, :54].float().masked_fill(~legal, float("-inf")), -1).gather(1, act[:, None]).squeeze(1)
    logp_old = lp_now + 0.4 * torch.randn(n, device="cuda", generator=g)            # ratios on both sides of the clip range
    adv = torch.randn(n, device="cuda", generator=g)
    ret = torch.randn(n, device="cuda", generator=g)
    w = (torch.rand(n, device="cuda", generator=g) < 0.8).float()
    clip, vf, ent_c = 0.2, 0.5, 0.01
    terms, dout = env.ppo_loss(out, words, act, logp_old, adv, ret, w, clip, vf, ent_c)
    x = out.float().requires_grad_(True)
    lg = x[:, :54].masked_fill(~legal, float("-inf"))
    logp_all = F.log_softmax(lg, dim=-1)
    logp = logp_all.gather(-1, act[:, None]).squeeze(-1)
    wsum = w.sum().clamp(min=1)
    ratio = (logp - logp_old).exp()
    pi = -(torch.min(ratio * adv, ratio.clamp(1 - clip, 1 + clip) * adv) * w).sum() / wsum
    v = (((x[:, 54] - ret) ** 2) * w).sum() / wsum
    p = logp_all.exp()
    H = (-(p * torch.where(legal, logp_all, torch.zeros_like(logp_all))).sum(-1) * w).sum() / wsum
    (pi + vf * v - ent_c * H).backward()
    ref = torch.stack([pi, v, H]).detach()
    assert torch.allclose(terms, ref, rtol=2e-3, atol=2e-4), (terms, ref)
    gref = x.grad
    err = (dout.float() - gref).abs()
    scale = gref.abs().max().item()
    assert err.max().item() < 0.01 * scale + 1e-9, (err.max().item(), scale)          # bf16 rounding of the output
    assert (dout[:, 55:] == 0).all().item() and (dout[:, :54][~legal] == 0).all().item()
    env.close()


def test_sample_policy_kernel(T, S):
    """tarok_sample_policy: always a legal card; log-prob equals torch's masked log-softmax;
    draws follow the softmax (pooled chi-square-style check on identical rows)."""
    import torch
    from tarok_amd import selfplay as SP
    n = 8192
    env = T.TarokVecEnv(n, seed=3, mix=S.MIX_ALL)
    obs = env.reset()
    for t in range(9):
        obs, _, _ = env.step(env.policy_random(obs), auto_reset=True)
    words = obs.words.clone()
    g = torch.Generator(device="cuda").manual_seed(0)
    logits = (torch.randn((n, 64), device="cuda", generator=g) * 2).to(torch.bfloat16)
    a, logp = env.sample_policy(logits, words)
    legal = SP.legal_matrix(words & T.karte.OBS_MASK)
    al = a.long()
    assert legal.gather(1, al.unsqueeze(1)).all().item()
    ref = torch.log_softmax(logits[:, :54].float().masked_fill(~legal, float("-inf")), dim=-1).gather(1, al.unsqueeze(1)).squeeze(1)
    assert (logp - ref).abs().max().item() < 2e-3
    # identical logits + identical legal sets in every game of a Klop opening lead -> frequencies ~ softmax
    env2 = T.TarokVecEnv(65536, seed=5, mix=S.MIX_FIXED + S.KLOP)
    o2 = env2.reset(deals=np.tile(np.arange(54, dtype=np.uint8), (65536, 1)), contract=np.zeros(65536, np.int8),
                    declarer=np.zeros(65536, np.int8))
    row = (torch.arange(64, device="cuda").float() * 0.15).to(torch.bfloat16)
    a2, _ = env2.sample_policy(row.repeat(65536, 1).contiguous(), o2.words)
    m = int(o2.mask_numpy()[0])
    ids = [c for c in range(54) if (m >> c) & 1]
    p = torch.softmax(row[:54].float()[ids], dim=0).cpu().numpy()
    freq = np.bincount(a2.cpu().numpy(), minlength=54)[ids] / 65536.0
    assert np.abs(freq - p).max() < 0.01
    env.close(); env2.close()


def test_checkpoint_roundtrip_and_resume(T, O, S):
    """state() -> set_state() is the identity on mid-game positions of every contract, and a
    restored env continues exactly like the original."""
    n = 8192
    a = T.TarokVecEnv(n, seed=14, mix=S.MIX_ALL)
    b = T.TarokVecEnv(n, seed=14, mix=S.MIX_ALL)
    oa = a.reset()
    b.reset()
    for t in range(48):
        if t in (0, 1, 5, 18, 31, 44, 47):
            snap = a.state()
            ob = b.set_state(snap)
            assert (b.state() == snap).all(), t
            assert (ob.words == a.legal_actions().words).all().item(), t
            act = a.policy_random(oa).clone()
            xa, ra, da = a.step(act)
            xb, rb, db = b.step(act)
            assert (xa.words == xb.words).all().item() and (da == db).all().item(), t
            assert (ra[da.bool()] == rb[db.bool()]).all().item()
            oa = xa
        else:
            oa, _, _ = a.step(a.policy_random(oa))
    assert (a.state() == b.state()).all()          # last restore was before the final card: same end state
    a.close(); b.close()


def test_reference_legal_move_cases_on_device(T, golden_dir):
    """The 600 (hand, lead) -> mozne_karte cases recorded from Klop.mozne_karte and
    Navadna_igra.mozne_karte, run through the device legal-card kernel as hand-built positions."""
    with open(os.path.join(golden_dir, "digests_v1.json")) as f:
        cases = json.load(f)["micro"]["legal_cases"]
    n = len(cases)
    env = T.TarokVecEnv(n, seed=0)
    for klop in (False, True):
        lanes = np.zeros((10, n), np.uint64)
        exp = np.zeros(n, np.uint64)
        for i, (hand, lead, nav, klo) in enumerate(cases):
            hand, lead = int(hand), int(lead)
            contract = 0 if klop else 8            # Klop vs Solo_brez (no exchange, no king)
            nt = 0 if lead < 0 else 1
            leader = 3                             # seat 3 led; seat 0 = (3+1)&3 is to move
            mover = (leader + nt) & 3
            lanes[mover, i] = hand
            rest = [c for c in range(54) if not (hand >> c) & 1 and c != lead]
            lanes[8, i] = sum(c << (6 * j) for j, c in enumerate(rest[:6]))             # a consistent talon
            lanes[4 + ((mover + 2) & 3), i] = sum(1 << c for c in rest[6:])             # every other card: taken
            trick = 0 if lead < 0 else lead
            meta = trick | (nt << 24) | (leader << 27) | (contract << 33) | (0 << 37) | (7 << 39) | ((0 if klop else 1) << 42) \
                | ((6 if klop else 0) << 46) | (7 << 49) | (2 << 52)
            lanes[9, i] = meta
            exp[i] = int(klo if klop else nav)
        obs = env.set_state(lanes)
        assert (obs.mask_numpy() == exp).all(), klop
    env.close()


def test_bad_inputs_flag_only_the_offending_games(T, S, traces):
    tr = traces
    n = 64
    deals = tr["deals"][:n].copy()
    contract = np.full(n, S.KLOP, np.int8)
    deals[3, 5] = deals[3, 6]                       # duplicate card
    deals[7, 50] = 60                               # id out of range
    contract[11] = 12                               # no such contract
    env = T.TarokVecEnv(n, seed=0)
    obs = env.reset(deals=deals, contract=contract, declarer=np.zeros(n, np.int8))
    err = obs.error.cpu().numpy()
    assert set(np.nonzero(err)[0]) == {3, 7, 11}
    # a bad talon group index
    idx = np.where((tr["contract"] == S.TRI))[0][:n]
    env.reset(deals=tr["deals"][idx], contract=tr["contract"][idx], declarer=tr["declarer"][idx],
              king_suit=np.maximum(tr["king"][idx], 0), defer_exchange=True)
    ch = tr["choice"][idx].copy()
    ch[5] = 2                                       # Tri has only groups 0 and 1
    obs = env.exchange(talon_choice=ch, discards=tr["discards"][idx])
    err = obs.error.cpu().numpy()
    assert err[5] and err.sum() == 1
    # games still waiting for the exchange ignore step() without complaint
    before = env.state()
    obs, _, done = env.step(np.full(n, 3, np.uint8))
    after = env.state()
    assert (before[:, 5] == after[:, 5]).all() and not done.cpu().numpy()[5]
    env.close()


def test_four_million_games_rollout_bit_exact(T, O, S):
    """Largest parity case: 2^22 mixed games through the fused rollout kernel vs the oracle."""
    n = 1 << 22
    env = T.TarokVecEnv(n, seed=77, mix=S.MIX_ALL, game_offset=123456789)
    out = env.rollout_random(episode=2)
    ref = O.rollout(77, 123456789, n, 2, S.MIX_ALL, threads=16, trace=False)
    assert (out["nsteps"].cpu().numpy() == ref["nsteps"]).all()
    assert (out["scores"].cpu().numpy() == ref["scores"]).all()
    env.close()


def test_million_games_headline_mode_vs_oracle(T, O, S):
    """The bench's mode (48 cards per launch, graph-replayed, auto-reset) on 2^20 games of a shard
    that does not start at game 0: episode numbers, score sums, canonical state and observation
    words after 192 lock-steps vs the oracle."""
    n, seed, steps, off = 1 << 20, 5, 192, 987654321
    ref = O.run_autoreset(seed, off, n, S.MIX_ALL, steps)
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL, game_offset=off)
    env.reset()
    env.run_random(steps, cards_per_launch=48, graph_chunk=96, auto_reset=True)
    ep, ss = env.counters()
    assert (ep == ref["episode"]).all()
    assert (ss == ref["score_sum"]).all()
    assert (env.state() == ref["lanes"]).all()
    assert (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all()
    env.close()


def test_four_million_games_headline_mode_is_deterministic_and_shards_agree(T, S):
    """2^22 games, 48 cards per launch, 96 lock-steps: two runs give identical counters and state
    (no launch-order or cache-placement dependence), and the third quarter of the batch equals a
    2^20-game env created on that shard alone."""
    n, steps = 1 << 22, 96
    res = []
    for rep in range(2):
        env = T.TarokVecEnv(n, seed=9, mix=S.MIX_ALL)
        env.reset()
        env.run_random(steps, cards_per_launch=48, graph_chunk=96, auto_reset=True)
        ep, ss = env.counters()
        res.append((ep.copy(), ss.copy(), env.state().copy()))
        env.close()
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all()
    q = n // 4
    part = T.TarokVecEnv(q, seed=9, mix=S.MIX_ALL, game_offset=2 * q)
    part.reset()
    part.run_random(steps, cards_per_launch=48, graph_chunk=96, auto_reset=True)
    ep, ss = part.counters()
    assert (ep == res[0][0][2 * q:3 * q]).all() and (ss == res[0][1][2 * q:3 * q]).all()
    assert (part.state() == res[0][2][:, 2 * q:3 * q]).all()
    part.close()


@pytest.mark.parametrize("lazy", [None, 0])
def test_mixed_launch_kinds_keep_the_refill_pipeline_consistent(T, O, S, lazy):
    """Graph replays, eager one-card steps, tricks, the two-kernel path and a mid-run reset, mixed
    (odd launch counts between graphs exercise the parity flush): still the oracle's games.  lazy (the default at
    this size): the one-card launches collect the lines they empty for a bulk deal every thirty-second launch; launches of
    the multi-card kernel in between drop what was collected."""
    n, seed = 20000, 41
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL, lazy_refill=lazy)
    env.reset()
    steps = 0
    env.run_random(96, cards_per_launch=4, graph_chunk=48, auto_reset=True); steps += 96
    for _ in range(3):
        env.step_random(auto_reset=True); steps += 1
    env.run_random(96, cards_per_launch=4, graph_chunk=48, auto_reset=True); steps += 96      # parity flipped: flush
    obs = env.legal_actions()
    for _ in range(5):
        obs, _, _ = env.step(env.policy_random(obs), auto_reset=True); steps += 1
    env.krog_random(7, auto_reset=True); steps += 7
    env.run_random(64, cards_per_launch=1, graph_chunk=32, auto_reset=True); steps += 64
    env.run_random(60, cards_per_launch=0, graph_chunk=20, auto_reset=True); steps += 60
    ref = O.run_autoreset(seed, 0, n, S.MIX_ALL, steps)
    ep, ss = env.counters()
    assert (ep == ref["episode"]).all() and (ss == ref["score_sum"]).all()
    assert (env.state() == ref["lanes"]).all()
    # a reset in the middle of everything starts clean (lists dropped, buffers refilled)
    env.reset(episode=5)
    env.run_random(192, cards_per_launch=4, graph_chunk=48, auto_reset=True)
    ref = O.run_autoreset(seed, 0, n, S.MIX_ALL, 192, episode0=5)
    ep, ss = env.counters()
    assert (ep == ref["episode"]).all() and (ss == ref["score_sum"]).all()
    assert (env.state() == ref["lanes"]).all()
    env.close()


@pytest.mark.parametrize("lazy", [None, 0])
def test_launch_kinds_mixed_at_full_size_vs_oracle(T, O, S, lazy):
    """65,536 games, forty segments of a seeded mix — graph-replayed and eager one-card launches, the two-kernel path,
    multi-card launches of 4 .. 128 cards between them (with the one-card step's bulk deals those drop the stretch
    lists, and slots come back to stale lines) — 1,867 lock-steps without a host synchronisation in between: every
    slot's episode number, score sums, canonical state and observation word vs the oracle.  (A refill role that
    fetched its list lengths with one load per wave passed every other test of this file and failed this sequence at
    lock-step 1,604: tools/soak_mixed.py is the longer version.)"""
    n, seed = 65536, 11
    rnd = np.random.RandomState(1)
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL, lazy_refill=lazy)
    env.reset()
    steps = 0
    for seg in range(40):
        kind = rnd.choice(["random", "two", "eager", "krog"])
        if kind == "random":
            k = int(rnd.choice([16, 48, 80, 112])); env.run_random(k, cards_per_launch=1, graph_chunk=16, auto_reset=True)
        elif kind == "two":
            k = int(rnd.choice([20, 60, 100])); env.run_random(k, cards_per_launch=0, graph_chunk=20, auto_reset=True)
        elif kind == "eager":
            k = int(rnd.randint(1, 23))
            for _ in range(k):
                env.step_random(auto_reset=True)
        else:
            k = int(rnd.choice([4, 8, 28, 48, 128])); env.krog_random(k, auto_reset=True)
        steps += k
    ref = O.run_autoreset(seed, 0, n, S.MIX_ALL, steps, threads=16)
    ep, ss = env.counters()
    assert (ep == ref["episode"]).all() and (ss == ref["score_sum"]).all()
    assert (env.state() == ref["lanes"]).all()
    assert (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all()
    env.close()


def _mixed_segments(env, rnd, target, between=None):
    """tools/soak_mixed.py's sequence: one-card stretches (graph-replayed, eager, the two-kernel path) cut by multi-card
    launches, no host synchronisation in between; `between(segment number)` may change the env's launch tuning."""
    steps, seg = 0, 0
    while steps < target:
        kind = rnd.choice(["random", "two", "eager", "krog"], p=[0.35, 0.3, 0.15, 0.2])
        if kind == "random":
            k = int(rnd.choice([16, 48, 80, 112])); env.run_random(k, cards_per_launch=1, graph_chunk=16, auto_reset=True)
        elif kind == "two":
            k = int(rnd.choice([20, 60, 100])); env.run_random(k, cards_per_launch=0, graph_chunk=20, auto_reset=True)
        elif kind == "eager":
            k = int(rnd.randint(1, 23))
            for _ in range(k):
                env.step_random(auto_reset=True)
        else:
            k = int(rnd.choice([4, 8, 28, 48, 128])); env.krog_random(k, auto_reset=True)
        steps += k; seg += 1
        if between:
            between(seg)
    return steps


def _assert_equals_oracle(env, O, S, seed, n, steps, mix=None):
    ref = O.run_autoreset(seed, 0, n, S.MIX_ALL if mix is None else mix, steps, threads=16)
    ep, ss = env.counters()
    assert (ep == ref["episode"]).all() and (ss == ref["score_sum"]).all()
    assert (env.state() == ref["lanes"]).all()
    assert (env.obs_words.cpu().numpy().view(np.uint64) == ref["obs"]).all()


@pytest.mark.parametrize("lazy,target", [(None, 6000), (0, 3000)])
def test_mixed_launch_soak_sequences_that_caught_the_refill_bug(T, O, S, lazy, target):
    """The two sequences of tools/soak_mixed.py at which round 3's failing refill-role builds went wrong (65,536 games:
    6,008 lock-steps with the bulk deals on, 3,058 with them off; those builds differed from the oracle in hundreds of
    slots here — profiles/r04_refill_soak_matrix.txt).  The cause was a dealt-ahead game written wrong by the refill loop
    (a gfx950 erratum the build now rules out statically: tests/test_isa_check.py); this is the end-to-end check."""
    n, seed = 65536, 11
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL, lazy_refill=lazy)
    env.reset()
    steps = _mixed_segments(env, np.random.RandomState(n % 1000 + target), target)
    _assert_equals_oracle(env, O, S, seed, n, steps)
    env.close()


@pytest.mark.parametrize("mixname,lazy", [("berac", None), ("berac", 0), ("klop", None), ("bot", None)])
def test_mixed_launch_kinds_on_batches_of_one_contract_family(T, O, S, mixname, lazy):
    """The mixed sequence on batches whose games end in step with each other or as fast as the rules allow: all Berac (games of
    4 .. 48 cards: a slot can end eight games in a stretch of 32 one-card launches — the stretch lists' capacity — and
    several per multi-card launch), all Klop (every game 48 cards: all 32,768 slots end on the same card, one stretch list
    entry per slot at once) and contracts from the on-device bidding; 32,768 games, ~2,000 lock-steps, vs the oracle."""
    n, seed, target = 32768, 23, 2000
    mix = {"berac": S.MIX_FIXED + 7, "klop": S.MIX_FIXED + 0, "bot": 2}[mixname]
    env = T.TarokVecEnv(n, seed=seed, mix=mix, lazy_refill=lazy)
    env.reset()
    steps = _mixed_segments(env, np.random.RandomState(101), target)
    _assert_equals_oracle(env, O, S, seed, n, steps, mix=mix)
    env.close()


def test_launch_tuning_changed_in_the_middle_of_a_mixed_run(T, O, S):
    """tarok_set_option between segments of a mixed run: the refill fan 1 -> 8 -> 3 -> 1 (each change restarts the env's
    launch counters and empties its refill lists behind a device synchronisation: the lines whose deal was dropped are
    dealt by their slots) and the bulk deals on -> off -> on; 65,536 games, ~2,400 lock-steps, all slots vs the oracle."""
    n, seed, target = 65536, 11, 2400
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL)
    env.reset()
    fans, lazies = [8, 3, 1, 8, 1], [0, 1, 0, 1]
    def between(seg):
        if seg % 7 == 3:
            env.set_option(refill_fan=fans[(seg // 7) % len(fans)])
        if seg % 5 == 2:
            env.set_option(lazy_refill=lazies[(seg // 5) % len(lazies)])
    steps = _mixed_segments(env, np.random.RandomState(77), target, between)
    _assert_equals_oracle(env, O, S, seed, n, steps)
    with pytest.raises(Exception):
        env.set_option(refill_fan=9)
    with pytest.raises(Exception):
        env.set_option(lazy_refill=2)
    env.close()


@pytest.mark.parametrize("n,lazy", [(65536, None), (20000, None), (65536, 0)])
def test_callers_own_graph_of_the_step_api_replayed_between_other_launches(T, O, S, n, lazy):
    """What an external policy's training loop does: ITS OWN torch.cuda.graph around `policy -> step` (seven lock-steps: a
    length that divides nothing the kernels count in), replayed dozens of times with eager steps, the library's graphs and
    multi-card launches in between.  The step kernels take their launch number from device-side counters (two kinds of
    launch, §3.3), so a replayed graph — fixed arguments — lands on the right refill lists wherever it is replayed."""
    import torch
    seed = 5
    env = T.TarokVecEnv(n, seed=seed, mix=S.MIX_ALL, lazy_refill=lazy)
    obs = env.reset()
    steps = 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up on the capture stream (allocations outside the capture)
        for _ in range(3):
            env.policy_random(); env.step(env.action, auto_reset=True); steps += 1
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(7):
            env.policy_random(); env.step(env.action, auto_reset=True)
    steps += 7                                          # (capture does not run the launches; the first replay below does)
    rnd = np.random.RandomState(3)
    g.replay()
    for it in range(60):
        kind = rnd.randint(0, 5)
        if kind <= 1:
            g.replay(); steps += 7
        elif kind == 2:
            k = int(rnd.randint(1, 9))
            for _ in range(k):
                env.policy_random(); env.step(env.action, auto_reset=True)
            steps += k
        elif kind == 3:
            env.run_random(32, cards_per_launch=1, graph_chunk=16, auto_reset=True); steps += 32
        else:
            k = int(rnd.choice([4, 12, 64])); env.krog_random(k, auto_reset=True); steps += k
    torch.cuda.synchronize()
    _assert_equals_oracle(env, O, S, seed, n, steps)
    env.close()


@pytest.mark.parametrize("n,lazy", [(65536, None), (65536, 0), (20000, None), (1 << 20, None)])
def test_refill_role_of_every_step_kernel_deals_its_lists_right(T, S, n, lazy):
    """tarok_debug_refill_selftest: the refill role as k_step (Bot policy / cards given, with and without the bulk-deal
    code) and k_play_wide compile it, fed with hand-built lists — every slot of every group, 1 / 4 / 14 entries per slot
    (one to fourteen passes of its loop on full waves), three list orders — and every line compared with a re-deal by a
    separate kernel.  Round 3's failing builds wrote 0.6 % of the lines wrong in exactly this setting (4 entries per
    slot; profiles/r04_refill_harness.txt) while every other test passed."""
    env = T.TarokVecEnv(n, seed=11, mix=S.MIX_ALL, lazy_refill=lazy)
    env.reset()
    combos = [(0, 1, 0), (0, 4, 0), (0, 14, 2), (1, 4, 1), (1, 14, 0), (2, 4, 0), (2, 14, 2), (3, 14, 1)]
    if n > (1 << 18):
        combos = [(0, 14, 0), (1, 4, 2), (2, 14, 0)]
    for kind, per_slot, order in combos:
        wrong, recs = env.refill_selftest(kind, per_slot, episode0=100 + 20 * per_slot, order=order, reps=3)
        assert wrong == 0, "kind %d, %d entries per slot, order %d: %d wrong lines, first %s" % (kind, per_slot, order, wrong, recs[:2].tolist())
    env.close()


def test_config1_single_klop_game_through_the_main_equivalent(T, O, S):
    """BASELINE config 1: ONE 4-player Klop game (the reference supports 4 players only,
    SURVEY §0) driven through the build's main-equivalent with reference-shaped players;
    scores equal the oracle's for the same deal and cards."""
    from tarok_amd import main as M
    seed = 6
    shared = {}
    players = [_SpecPlayer(i, S, seed, S.MIX_FIXED + S.KLOP, shared) for i in range(4)]
    out = M.main(st_iger=1, iterations=1, seed=seed, igralci=players, verbose=False)
    ref = O.rollout(seed, 0, 1, 0, S.MIX_FIXED + S.KLOP)
    order = {p.ime: p.seat[0] for p in players}                  # seats after main's shuffle
    assert [out[0][str(i)] for i in range(4)] == [int(ref["scores"][0][order[str(i)]]) for i in range(4)]
    assert shared["tricks"][0] == [5] * 6 + [4] * 6
    # and the Bot players of the reference run through it too
    res = M.main(st_iger=32, iterations=2, seed=1, verbose=False)
    assert len(res) == 2 and all(len(r) == 4 for r in res)


def test_fused_policy_mlp_kernel_vs_torch(T, S):
    """tarok_policy_mlp (features -> 256-256-256-64 MLP on MFMA -> masked sample): features equal
    tarok_observe's, head outputs equal a float32 torch evaluation of the same bf16 weights within
    bf16 rounding, actions legal, log-probs consistent with the kernel's own logits."""
    import torch
    from tarok_amd import selfplay as SP
    n = 10000                                        # not a multiple of 64: ragged last workgroup
    env = T.TarokVecEnv(n, seed=23, mix=S.MIX_ALL)
    obs = env.reset()
    for t in range(7):
        obs, _, _ = env.step(env.policy_random(obs), auto_reset=True)
    words = obs.words.clone()
    torch.manual_seed(0)
    net = SP.PolicyNet(256).cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(3.0)                              # spread the logits a little
    w = [net.fc1.weight.detach().to(torch.bfloat16).contiguous(), net.fc1.bias.detach().float().contiguous(),
         net.fc2.weight.detach().to(torch.bfloat16).contiguous(), net.fc2.bias.detach().float().contiguous(),
         net.head.weight.detach().to(torch.bfloat16).contiguous(), net.head.bias.detach().float().contiguous()]
    feat = torch.zeros((n, 256), dtype=torch.bfloat16, device="cuda")
    wk = [env.mfma_weight_order(w[0]), w[1], env.mfma_weight_order(w[2]), w[3], env.mfma_weight_order(w[4]), w[5]]
    fw = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    a, logp, val = env.policy_mlp(wk, words, features_out=feat, feature_words_out=fw)
    ref_feat = env.observe()
    assert (feat == ref_feat).all().item()
    assert (env.expand_feature_words(fw) == ref_feat).all().item()      # the 32-byte form of the same features
    idx = torch.randperm(n, device="cuda")[:7777]
    assert (env.gather_features(fw, idx) == ref_feat[idx]).all().item()    # gather + expansion kernel
    assert (env.gather_features(fw) == ref_feat).all().item()
    x = ref_feat.float()
    h = torch.relu(x @ w[0].float().T + w[1]).to(torch.bfloat16).float()
    h = torch.relu(h @ w[2].float().T + w[3]).to(torch.bfloat16).float()
    out = h @ w[4].float().T + w[5]
    assert (val - out[:, 54]).abs().max().item() < 0.05 * (1 + out[:, 54].abs().max().item())
    legal = SP.legal_matrix(words & T.karte.OBS_MASK)
    al = a.long()
    assert legal.gather(1, al.unsqueeze(1)).all().item()
    ref_logp = torch.log_softmax(out[:, :54].masked_fill(~legal, float("-inf")), dim=-1).gather(1, al.unsqueeze(1)).squeeze(1)
    assert (logp - ref_logp).abs().max().item() < 0.08
    assert (logp - ref_logp).abs().mean().item() < 0.01
    # same draw as the stand-alone sampler fed with the reference logits, except where rounding moves a boundary
    pad = torch.zeros((n, 64), dtype=torch.bfloat16, device="cuda")
    pad[:, :54] = out[:, :54].to(torch.bfloat16)
    a2, _ = env.sample_policy(pad, words)
    assert (a2 == a).float().mean().item() > 0.97
    env.close()


@pytest.mark.parametrize("mixname,fan,lazy", [("all", None, None), ("berac", None, None), ("berac", 3, None), ("all", None, 0), ("berac", 3, 0)])
def test_random_api_sequences_against_an_oracle_model(T, O, S, mixname, fan, lazy):
    """Model-based fuzz: a random sequence of API calls (one-card steps with legal / illegal /
    garbage cards, in-kernel-policy steps, 1..48 cards per launch, auto-reset on and off, resets)
    on the GPU env vs the same sequence applied slot by slot to the CPU oracle; canonical state,
    observation words, episode numbers and score sums compared after every call.  The all-Berac
    mix makes slots finish several games inside one launch (swap-ins from more than one
    next-game line, lines on a refill list, games dealt in place).  fan: refill lists per refill workgroup
    (three lists of a 768-slot env: one refill workgroup among three play workgroups); lazy: the one-card step's bulk deals
    (the default at this size) or per-launch lists (0)."""
    import ctypes as C
    rnd = np.random.RandomState(12345)
    n, seed, mix = 768, 77, (S.MIX_ALL if mixname == "all" else S.MIX_FIXED + 7)
    L = O.lib()
    env = T.TarokVecEnv(n, seed=seed, mix=mix, refill_fan=fan, lazy_refill=lazy)

    class Slot:
        pass
    slots = []

    def new_game(sl, ep):
        sl.ep = ep
        sl.g = O.Game.synth(seed, sl.i, ep, mix)
        sl.key = S.game_key(seed, sl.i, ep)

    def model_reset(ep):
        for sl in slots:
            new_game(sl, ep)
            sl.sum = [0, 0, 0, 0]
            sl.fin = False

    for i in range(n):
        sl = Slot(); sl.i = i; slots.append(sl)
    env.reset(episode=0)
    model_reset(0)

    def model_card(sl, a, auto):
        """one card (a = None: the Bot policy).  Mirrors k_play's per-card logic."""
        g = sl.g
        sl.fin = False
        if g.g.phase == 2:
            if a is None:
                a = L.to_policy_action(sl.key, g.g.trick_no * 4 + g.g.n_in_trick, g.legal())
            r = g.step(a)
            if r == 1:
                sl.fin = True
                for s in range(4):
                    sl.sum[s] += g.g.score[s]
        if auto and g.g.phase == 3:
            new_game(sl, sl.ep + 1)

    def check(tag):
        st = env.state()
        ep, ss = env.counters()
        words = env.obs_words.cpu().numpy().view(np.uint64)
        for sl in slots:
            assert (st[:, sl.i] == sl.g.lanes()).all(), (tag, sl.i)
            assert ep[sl.i] == sl.ep and list(ss[sl.i]) == sl.sum, (tag, sl.i)
            assert int(words[sl.i]) == int(L.to_obs_word(C.byref(sl.g.g), 1 if sl.fin else 0)), (tag, sl.i)

    for op in range(110):
        kind = rnd.choice(["random1", "explicit", "krog", "two_kernel", "reset"], p=[0.25, 0.3, 0.3, 0.12, 0.03])
        auto = bool(rnd.rand() < 0.7)
        if kind == "random1":
            env.step_random(auto_reset=auto)
            for sl in slots:
                model_card(sl, None, auto)
        elif kind == "two_kernel":
            obs = env.legal_actions()
            env.step(env.policy_random(obs), auto_reset=auto)
            for sl in slots:
                model_card(sl, None, auto)
        elif kind == "krog":
            cards = int(rnd.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 24, 48]))
            env.krog_random(cards, auto_reset=auto)
            for c in range(cards):
                for sl in slots:
                    model_card(sl, None, auto)
        elif kind == "explicit":
            acts = np.zeros(n, np.uint8)
            for sl in slots:
                m = sl.g.legal()
                u = rnd.rand()
                if m and u < 0.85:
                    ids = [c for c in range(54) if (m >> c) & 1]
                    acts[sl.i] = ids[rnd.randint(len(ids))]
                elif u < 0.95:
                    acts[sl.i] = rnd.randint(0, 54)          # often illegal
                else:
                    acts[sl.i] = rnd.randint(54, 256)        # garbage
            env.step(acts, auto_reset=auto)
            for sl in slots:
                model_card(sl, int(acts[sl.i]), auto)
        else:
            ep0 = int(rnd.randint(0, 1000))
            env.reset(episode=ep0)
            model_reset(ep0)
            env.legal_actions()
        if kind != "reset" or True:
            check((op, kind, auto))
    env.close()
