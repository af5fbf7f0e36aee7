"""The CPU oracle (oracle/tarok_oracle.c) against fixtures produced by RUNNING the
reference (oracle/gen_golden.py -> tests/golden/).  This is what pins the oracle;
the GPU parity tests (test_gpu_*.py) then compare the HIP path with the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import tarok_spec as S


@pytest.fixture(scope="module")
def traces(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "traces_v1.npz")))


@pytest.fixture(scope="module")
def digests(golden_dir):
    with open(os.path.join(golden_dir, "digests_v1.json")) as f:
        return json.load(f)


def test_micro_prestej_and_values(digests):
    m = digests["micro"]
    L = O.lib()
    assert m["prestej_full_deck"] == 70 == L.to_prestej(S.DECK)
    assert m["v_id_roundtrip"] is True
    assert [L.to_vrednost(i) for i in range(54)] == m["vrednost_by_id"]
    disc = L.to_discardable(S.DECK)
    assert [i for i in range(54) if (disc >> i) & 1] == m["discardable_ids"]
    assert disc == S.DISCARDABLE
    assert [L.to_prestej(1 << i) for i in range(54)] == m["single_card_prestej"]
    for mask, val in m["random_piles"]:
        assert L.to_prestej(int(mask)) == val


def test_micro_legal_moves(digests):
    L = O.lib()
    for hand, lead, nav, klop in digests["micro"]["legal_cases"]:
        assert L.to_legal_navadna(int(hand), lead) == int(nav)
        assert L.to_legal_klop(int(hand), lead) == int(klop)


def test_micro_trick_winner(digests):
    L = O.lib()
    for a, b, c, d, w in digests["micro"]["trick_cases"]:
        assert L.to_trick_winner(O.u8arr([a, b, c, d])) == w


def replay(tr, i):
    """Replay fixture game i on the oracle, checking every seat and legal mask."""
    g = O.Game(tr["deals"][i], tr["contract"][i], tr["declarer"][i], tr["king"][i])
    if g.g.phase == 1:
        n = S.N_DISCARD[int(tr["contract"][i])]
        assert g.exchange(tr["choice"][i], tr["discards"][i][:n]) == 0
    n = int(tr["nsteps"][i])
    for t in range(n):
        assert not g.done
        assert g.seat() == tr["seats"][i, t], (i, t)
        assert g.legal() == int(tr["masks"][i, t]), (i, t)
        r = g.step(tr["actions"][i, t])
        assert r == (1 if t == n - 1 else 0), (i, t, r)
        lt = int(g.g.last_trick)
        if t % 4 == 3:      # what rezultat_stiha was told: (sem_pobral seat, Roka.vrednost_stiha(stih))
            assert lt == 0x8000 | (int(tr["trick_value"][i, t // 4]) << 4) | int(tr["trick_winner"][i, t // 4]), (i, t)
        else:
            assert lt == 0
    assert g.done
    assert g.scores == [int(x) for x in tr["scores"][i]], i
    for s in range(4):
        assert int(g.g.pile[s]) == int(tr["piles"][i, s]), (i, s)
        assert int(g.g.hand[s]) == int(tr["hands_end"][i, s]), (i, s)
    return g


def test_traces_all_contracts(traces):
    seen = set()
    for i in range(len(traces["contract"])):
        replay(traces, i)
        seen.add(int(traces["contract"][i]))
    assert seen == set(range(10))


def test_traces_cover_the_quirks(traces):
    """The fixture set exercises the reference quirks listed in SURVEY A.8."""
    c = traces["contract"]
    sc = traces["scores"]
    klop = sc[c == S.KLOP]
    assert (klop == 0).all(axis=1).any()            # A.8-1: someone >35 -> all zero
    assert (klop < 0).any()
    ber = np.isin(c, [S.BERAC, S.ODPRTI_BERAC])
    assert (traces["nsteps"][ber] < 48).any() and (traces["nsteps"][ber] == 48).any()
    assert set(np.unique(np.abs(sc[c == S.BERAC]))) <= {0, 70}
    assert set(np.unique(np.abs(sc[c == S.ODPRTI_BERAC]))) <= {0, 90}
    nav = (c >= S.TRI) & (c <= S.ENA)
    assert ((sc[nav] != 0).sum(axis=1) == 2).any()  # called-king partner
    assert ((sc[nav] != 0).sum(axis=1) == 1).any()  # king in talon / own hand -> alone


def test_illegal_action_sets_error_and_keeps_state(traces):
    g = O.Game(traces["deals"][0], S.KLOP, 0, -1)
    before = g.lanes().copy()
    legal = g.legal()
    bad = [i for i in range(54) if not (legal >> i) & 1][0]
    assert g.step(bad) == -1 and g.g.error == 1
    after = g.lanes()
    assert (before[:9] == after[:9]).all()
    assert g.step(255) == -1


def test_spec_rng_c_equals_python():
    L = O.lib()
    for seed, gidx, ep in [(0, 0, 0), (1, 5, 2), (2**63 + 11, 2**40 + 3, 77), (12345, 65535, 1)]:
        k = S.game_key(seed, gidx, ep)
        assert L.to_game_key(seed, gidx, ep) == k
        for i in (0, 1, 53, 64, 70, 128, 175):
            assert L.to_rng32(k, i) == S.rng32(k, i)
        perm = O.u8arr([0] * 54)
        L.to_deal_perm(k, perm)
        assert list(perm) == S.deal(k)
        assert sorted(perm) == list(range(54))


def digest(p):
    h = hashlib.sha256()
    for name in ("nsteps", "seats", "masks", "actions", "scores"):
        h.update(np.ascontiguousarray(p[name]).tobytes())
    return h.hexdigest()


def test_synth_small_full_arrays(golden_dir):
    ref = dict(np.load(os.path.join(golden_dir, "synth_small_v1.npz")))
    got = O.rollout(seed=2, gidx0=0, n=2048, episode=3, mix=S.MIX_ALL, threads=2)
    for name in ("nsteps", "seats", "masks", "actions", "scores"):
        assert (got[name] == ref[name]).all(), name
    # and the deals / setup themselves
    L = O.lib()
    import ctypes as C
    for g in range(0, 2048, 97):
        k = S.game_key(2, g, 3)
        perm = O.u8arr([0] * 54)
        L.to_deal_perm(k, perm)
        assert list(perm) == list(ref["deals"][g])
        c, d, kk = C.c_int(), C.c_int(), C.c_int()
        L.to_sample_setup(k, S.MIX_ALL, C.byref(c), C.byref(d), C.byref(kk))
        assert (c.value, d.value, kk.value) == (ref["contract"][g], ref["declarer"][g], ref["king"][g])


def test_synth_digests_match_reference(digests):
    """Full-size configs pinned by SHA-256 of the
    reference engine's own masks/seats/actions/scores on the same deals."""
    for r in digests["synthetic"]:
        got = O.rollout(r["seed"], 0, r["n"], r["episode"], r["mix"], threads=4)
        assert got["total_steps"] == r["total_steps"], r["name"]
        assert [int(x) for x in got["scores"].astype(np.int64).sum(0)] == r["score_sums_by_seat"]
        assert digest(got) == r["sha256"], r["name"]
