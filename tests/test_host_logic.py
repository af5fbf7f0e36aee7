"""Host-side pieces of the boundary against fixtures produced by running the reference:
the bidding round (Igra.licitacija) and the Karta / Roka value types the callbacks get."""
import json
import os

import pytest

from tarok_amd import igralec as I
from tarok_amd import karte as K
from tarok_amd import licitacija as L


@pytest.fixture(scope="module")
def micro(golden_dir):
    with open(os.path.join(golden_dir, "digests_v1.json")) as f:
        return json.load(f)["micro"]


def test_licitacija_same_calls_same_outcome(golden_dir):
    """Every licitiram call (seat, min_igra, obvezno, prednost) in the reference's order,
    and the same (declarer, contract), over 1500 scripted rounds."""
    with open(os.path.join(golden_dir, "licitacija_v1.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 1000
    for c in cases:
        script = [list(s) for s in c["scripts"]]
        calls = []

        def ask(seat, min_igra, obvezno, prednost):
            want = script[seat].pop(0)
            calls.append([seat, min_igra, obvezno, prednost, want])
            return L.base_filter(want, min_igra, obvezno, prednost)
        got = L.licitacija(ask)
        assert calls == c["calls"]
        assert list(got) == c["result"]


def test_licitacija_detects_broken_protocol():
    with pytest.raises(RuntimeError):
        # two bidders, then everybody (incl. the holder) says Naprej: the reference spins forever
        answers = iter([L.DVE, L.ENA, L.NAPREJ, L.NAPREJ] + [L.NAPREJ] * 10)
        L.licitacija(lambda *a: next(answers))


def test_karta_and_roka_match_the_reference(micro):
    cards = [I.Karta.iz_id(i) for i in range(54)]
    assert [k.v_id() for k in cards] == list(range(54))
    assert [k.vrednost() for k in cards] == micro["vrednost_by_id"]
    assert sorted(k.v_id() for k in I.Roka(cards).mozno_zalozit()) == micro["discardable_ids"]
    assert I.Roka.prestej(cards) == micro["prestej_full_deck"] == 70
    assert [I.Roka.prestej([k]) for k in cards] == micro["single_card_prestej"]
    for mask, val in micro["random_piles"]:
        assert I.Roka.prestej([I.Karta.iz_id(i) for i in K.mask_to_ids(int(mask))]) == val
    assert str(I.Karta(K.Barva.KARA, 8)) == "KARA_KR" and str(I.Karta(K.Barva.TAROK, 21)) == "TAROK_21"
    r = I.Roka([I.Karta.iz_id(i) for i in (3, 40, 1, 33)])
    assert len(r) == 4 and I.Karta.iz_id(40) in r and I.Karta.iz_id(41) not in r
    r.igraj_karto(I.Karta.iz_id(40))
    assert len(r) == 3 and [k.v_id() for k in r] == [1, 3, 33]


def test_base_player_filter_and_bot_bids():
    p = I.Igralec("a")
    assert p.licitiram(L.DVE, L.TRI, 0) == L.DVE
    assert p.licitiram(L.TRI, L.TRI, 0) == L.NAPREJ
    assert p.licitiram(L.TRI, L.TRI, 0, prednost=True) == L.TRI
    assert p.licitiram(L.NAPREJ, L.NAPREJ, 0, obvezno=L.KLOP) == L.KLOP
    import random
    b = I.Bot_igralec("b", rng=random.Random(1))
    assert all(b.licitiram(L.TRI, 0) in (L.NAPREJ, L.DVE, L.ENA) for _ in range(50))


def test_bot_bidding_three_statements_agree():
    """TAROK_MIX_BOT: the host bidding round (pinned to Igra.licitacija by the fixture above)
    driven by Bot wishes from the spec draws == the spec's straight-line form == the C oracle's."""
    import ctypes as C
    from oracle import oracle as O
    from oracle import tarok_spec as S
    Lc = O.lib()
    hist = {}
    for g in range(20000):
        key = S.game_key(9, g, g % 5)
        calls = [0]

        def ask(seat, min_igra, obvezno, prednost):
            wish = S.bot_wish(S.rng32(key, S.DRAW_BID + calls[0]))
            calls[0] += 1
            return L.base_filter(wish, min_igra, obvezno, prednost)
        d, value = L.licitacija(ask)
        assert (d, value // 10) == S.bot_bidding(key), g
        c_, d_ = C.c_int(), C.c_int()
        Lc.to_bot_bidding(key, C.byref(c_), C.byref(d_))
        assert (d_.value, c_.value) == (d, value // 10), g
        hist[value] = hist.get(value, 0) + 1
    assert set(hist) == {0, 10, 20, 30}          # the Bot never bids above Ena (Igralec.py:151)
    assert hist[0] > 1000 and hist[30] > hist[10]
