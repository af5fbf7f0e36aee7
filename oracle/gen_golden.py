#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/* by RUNNING THE REFERENCE.

Run in the build container only (the reference does not exist on the GPU box):

    python3 -B oracle/gen_golden.py [--reference /root/reference] [--out tests/golden]

It imports the reference's rule modules (Karta, Roka, Klop, Berac, Navadna_igra,
Igra, Tarok — stdlib-only) *as they lie* under /root/reference, injects
deals through the ``Igra.shuffle`` module attribute (Igra.py:10,67), drives the
engines with a build-authored duck-typed player (the reference's own Igralec.py
cannot be imported: it needs pytorch_lightning and a torch_models module that is
not in the reference), and records, per game:

    deal permutation, contract, declarer, king suit, talon-group choice, discards,
    per step (seat, legal-mask, action), final scores, final piles.

Nothing from the reference is copied: the fixtures are inputs and observed outputs.

Files written:
    traces_v1.npz    ~2.8k games over all 10 contracts, random legal play
    digests_v1.json  SHA-256 digests of the large synthetic configs
                     driven by oracle/tarok_spec.py's RNG, + known-answer micro vectors
    paralel_v1.npz   Tarok.paralel_start lock-step runs (Tarok.py:30-62) for the adapter
    licitacija_v1.json  scripted bidding rounds: every licitiram call Igra.licitacija makes
                     (seat, min_igra, obvezno, prednost, answer) and the outcome (Igra.py:75-114)
"""
import argparse
import hashlib
import json
import os
import random
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tarok_spec as S  # noqa: E402

REF = None  # set in main(): dict of reference modules


def load_reference(path):
    global REF
    if REF is not None:
        return REF
    sys.path.insert(0, path)
    import Karta, Tip_igre, Roka, Klop, Berac, Navadna_igra, Igra, Tarok  # noqa: E401
    REF = dict(Karta=Karta, Tip_igre=Tip_igre, Roka=Roka, Klop=Klop, Berac=Berac,
               Navadna_igra=Navadna_igra, Igra=Igra, Tarok=Tarok)
    return REF


def cards_mask(cards):
    m = 0
    for k in cards:
        m |= 1 << k.v_id()
    return m


class TracePlayer:
    """Duck-typed stand-in for Igralec (behaviour of Igralec.py:32-122): owns
    roka[id] / kupcek[id]; every decision is delegated to ``ctl`` and recorded."""

    def __init__(self, ime, ctl):
        self.ime = str(ime)
        self.ctl = ctl
        self.roka = {}
        self.kupcek = {}
        self.bid = {}          # id_igre -> desired Tip_igre
        self.seat = {}         # id_igre -> seat in that game

    def __str__(self):
        return "T" + self.ime

    __repr__ = __str__

    # --- protocol -------------------------------------------------------
    def nova_igra(self, roka, igralci, id_igre):
        self.roka[id_igre] = roka
        self.kupcek[id_igre] = []
        self.seat[id_igre] = [i for i, p in enumerate(igralci) if p is self][0]
        self.ctl.on_nova_igra(self, roka, id_igre)

    def pripavi_licitiram(self, id_igre):
        pass

    def predict_licitiram(self):
        pass

    def licitiram(self, min_igra, id_igre, obvezno=None, prednost=False):
        T = REF["Tip_igre"].Tip_igre
        want = self.ctl.bid(self, id_igre, min_igra, obvezno, prednost)
        # base filter, Igralec.py:58-74
        if prednost:
            ok = want >= min_igra
        else:
            ok = want > min_igra
        if ok:
            return want
        return T.Naprej if obvezno is None else obvezno

    def izberi_barvo_kralja(self, id_igre):
        return REF["Karta"].Barva(self.ctl.king(self, id_igre))

    def konec_licitiranja(self, igralec_ki_igra, tip_igre, id_igre, barva_kralja=None):
        self.ctl.on_konec_licitiranja(self, igralec_ki_igra, tip_igre, id_igre, barva_kralja)

    def pripravi_izbral_iz_talona(self, talon, st_kupcka, id_igre):
        pass

    def predict_izberi_iz_talona(self):
        pass

    def menjaj_iz_talona(self, kupcki, st_kart, id_igre):
        groups = [cards_mask(g) for g in kupcki]
        hand_before = cards_mask(self.roka[id_igre])
        choice = self.ctl.choose_group(self, id_igre, groups, st_kart)
        self.roka[id_igre].dodaj_karte(kupcki[choice])
        hand = cards_mask(self.roka[id_igre])
        assert hand == hand_before | groups[choice]
        ids = self.ctl.choose_discards(self, id_igre, hand, st_kart)
        izberi = [REF["Karta"].Karta.iz_id(i) for i in ids]
        self.kupcek[id_igre].extend(izberi)
        for k in izberi:
            self.roka[id_igre].igraj_karto(k)
        return choice

    def izbral_iz_talona(self, talon, st_kupcka, id_igre):
        pass

    def poglej_karte_odprtega_beraca(self, roka, id_igre):
        pass

    def pripravi_igraj_karto(self, karte_na_mizi, mozne, zgodovina, id_igre):
        pass

    def predict_igraj_karto(self):
        pass

    def igraj_karto(self, karte_na_mizi, mozne, zgodovina, id_igre):
        mask = cards_mask(mozne)
        seat = self.seat[id_igre]
        cid = self.ctl.choose_card(self, id_igre, seat, mask, cards_mask(karte_na_mizi))
        karta = [k for k in mozne if k.v_id() == cid][0]
        self.roka[id_igre].igraj_karto(karta)      # Igralec.py:82-85
        return karta

    def rezultat_stiha(self, stih, sem_pobral, id_igre):
        if sem_pobral:
            self.ctl.on_trick(self, id_igre, [k.v_id() for k in stih], REF["Roka"].Roka.vrednost_stiha(stih))

    def rezultat_igre(self, st_tock, povzetek_igre, id_igre):
        self.ctl.on_result(self, id_igre, st_tock)


class Recorder:
    """Controller for ONE game: forces contract/declarer/king and records."""

    def __init__(self, contract, declarer, king, group_fn, discard_fn, card_fn):
        self.contract, self.declarer, self.king_suit = contract, declarer, king
        self.group_fn, self.discard_fn, self.card_fn = group_fn, discard_fn, card_fn
        self.seats, self.masks, self.actions = [], [], []
        self.tricks = []                 # (winner seat, [card ids])
        self.scores = [None] * 4
        self.choice, self.discards = -1, []
        self.hands0 = [0] * 4

    def on_nova_igra(self, p, roka, gid):
        self.hands0[p.seat[gid]] = cards_mask(roka)

    def bid(self, p, gid, min_igra, obvezno, prednost):
        T = REF["Tip_igre"].Tip_igre
        if p.seat[gid] == self.declarer and self.contract != S.KLOP:
            return T(self.contract * 10)
        return T.Naprej

    def king(self, p, gid):
        return self.king_suit

    def on_konec_licitiranja(self, p, decl, tip, gid, barva):
        assert int(tip) == self.contract * 10, (tip, self.contract)
        assert decl.seat[gid] == self.declarer

    def choose_group(self, p, gid, groups, k):
        self.choice = self.group_fn(groups, k)
        return self.choice

    def choose_discards(self, p, gid, hand, k):
        self.discards = list(self.discard_fn(hand, k))
        return self.discards

    def choose_card(self, p, gid, seat, mask, table):
        step = len(self.actions)
        a = self.card_fn(step, seat, mask)
        self.seats.append(seat)
        self.masks.append(mask)
        self.actions.append(a)
        return a

    def on_trick(self, p, gid, ids, value=0):
        self.tricks.append((p.seat[gid], ids, int(value)))

    def on_result(self, p, gid, pts):
        self.scores[p.seat[gid]] = int(pts)


def run_reference_game(perm, contract, declarer, king, group_fn, discard_fn, card_fn, flow="direct"):
    """Play one game on the reference engine.  flow='igra' goes through
    Igra.start (Igra.py:26-62, incl. licitacija); flow='direct' constructs the
    contract engine exactly as Igra.py:38-55 does, skipping only the bidding —
    needed for (contract, declarer) pairs the bidding rules cannot produce."""
    R = REF
    T = R["Tip_igre"].Tip_igre
    IgraMod = R["Igra"]
    perm = list(perm)

    def fake_shuffle(lst):
        lst[:] = perm
    IgraMod.shuffle = fake_shuffle

    rec = Recorder(contract, declarer, king, group_fn, discard_fn, card_fn)
    players = [TracePlayer(i, rec) for i in range(4)]
    ig = IgraMod.Igra(players, multi_games=False, id=0)
    if flow == "igra":
        res = next(ig.start())
    else:
        talon = ig.razdeli()
        tip = T(contract * 10)
        barva = None
        if tip in (T.Ena, T.Dve, T.Tri, T.Solo_brez, T.Solo_ena, T.Solo_dve, T.Solo_tri):
            if tip in (T.Ena, T.Dve, T.Tri):
                barva = players[declarer].izberi_barvo_kralja(0)
            eng = R["Navadna_igra"].Navadna_igra(players, tip, barva, players[declarer], talon, id_igre=0)
        elif tip == T.Klop:
            eng = R["Klop"].Klop(players, talon, 0)
        elif tip == T.Berac:
            eng = R["Berac"].Berac(players, players[declarer], talon, False, 0)
        elif tip == T.Odprti_berac:
            eng = R["Berac"].Berac(players, players[declarer], talon, True, 0)
        else:
            raise ValueError(tip)
        for p in players:
            p.konec_licitiranja(players[declarer], tip, 0, barva)
        res = list(eng.start())[-1]
    scores = [int(res.get(p, 0)) for p in players]
    assert scores == rec.scores, (scores, rec.scores)
    piles = [cards_mask(p.kupcek[0]) for p in players]
    hands_end = [cards_mask(p.roka[0]) for p in players]
    return dict(seats=rec.seats, masks=rec.masks, actions=rec.actions, scores=scores,
                piles=piles, hands_end=hands_end, hands0=rec.hands0,
                choice=rec.choice, discards=rec.discards, tricks=rec.tricks)


def igra_reachable(contract, declarer):
    """(contract, declarer) pairs Igra.licitacija (Igra.py:75-114) can produce
    when exactly one seat bids: seat 0 anything; seats 1-3 only above Tri."""
    if contract == S.KLOP:
        return declarer == 0
    return declarer == 0 or contract >= S.DVE


# --------------------------------------------------------------------------
# trace fixture
# --------------------------------------------------------------------------
def pack_games(games):
    G = len(games)
    out = dict(
        deals=np.zeros((G, 54), np.uint8), contract=np.zeros(G, np.int8),
        declarer=np.zeros(G, np.int8), king=np.full(G, -1, np.int8),
        choice=np.full(G, -1, np.int8), discards=np.full((G, 3), 255, np.uint8),
        nsteps=np.zeros(G, np.int16), seats=np.full((G, 48), -1, np.int8),
        masks=np.zeros((G, 48), np.uint64), actions=np.full((G, 48), 255, np.uint8),
        scores=np.zeros((G, 4), np.int16), piles=np.zeros((G, 4), np.uint64),
        hands_end=np.zeros((G, 4), np.uint64), flow=np.zeros(G, np.uint8),
        trick_winner=np.full((G, 12), -1, np.int8), trick_value=np.full((G, 12), -1, np.int8),
    )
    for i, g in enumerate(games):
        out["deals"][i] = g["perm"]
        out["contract"][i] = g["contract"]
        out["declarer"][i] = g["declarer"]
        out["king"][i] = g["king"]
        out["choice"][i] = g["choice"]
        for j, d in enumerate(g["discards"]):
            out["discards"][i, j] = d
        n = len(g["actions"])
        out["nsteps"][i] = n
        out["seats"][i, :n] = g["seats"]
        out["masks"][i, :n] = np.array(g["masks"], dtype=np.uint64)
        out["actions"][i, :n] = g["actions"]
        out["scores"][i] = g["scores"]
        out["piles"][i] = np.array(g["piles"], dtype=np.uint64)
        out["hands_end"][i] = np.array(g["hands_end"], dtype=np.uint64)
        out["flow"][i] = 1 if g["flow"] == "igra" else 0
        for j, (w, ids, val) in enumerate(g["tricks"]):
            out["trick_winner"][i, j] = w          # rezultat_stiha(sem_pobral=True) seat
            out["trick_value"][i, j] = val         # Roka.vrednost_stiha(stih), Roka.py:76-95
    return out


def gen_traces(per_contract, seed):
    rnd = random.Random(seed)
    games = []
    for contract in range(10):
        for n in range(per_contract):
            perm = list(range(54))
            rnd.shuffle(perm)
            declarer = 0 if contract == S.KLOP else rnd.randrange(4)
            king = rnd.randrange(4) if contract in (S.TRI, S.DVE, S.ENA) else -1
            flow = "igra" if (n % 2 == 0 and igra_reachable(contract, declarer)) else "direct"
            # every 4th game discards from the WHOLE hand (the env does not validate
            # discards, Navadna_igra.py:62; kings/trula may land in the pile -> quirk A.8-5)
            free_discards = (n % 4 == 3)

            def group_fn(groups, k, rnd=rnd):
                return rnd.randrange(len(groups))

            def discard_fn(hand, k, rnd=rnd, free=free_discards):
                cand = hand if free else (hand & S.DISCARDABLE)
                ids = [i for i in range(54) if (cand >> i) & 1]
                if len(ids) < k:
                    ids = [i for i in range(54) if (hand >> i) & 1]
                return rnd.sample(ids, k)

            def card_fn(step, seat, mask, rnd=rnd):
                ids = [i for i in range(54) if (mask >> i) & 1]
                return rnd.choice(ids)

            g = run_reference_game(perm, contract, declarer, king, group_fn, discard_fn, card_fn, flow)
            g.update(perm=perm, contract=contract, declarer=declarer, king=king, flow=flow)
            games.append(g)
    return pack_games(games)


# --------------------------------------------------------------------------
# large synthetic configs -> digests
# --------------------------------------------------------------------------
def synth_game(seed, gidx, episode, mix):
    key = S.game_key(seed, gidx, episode)
    perm = S.deal(key)
    contract, declarer, king = S.sample_setup(key, mix)
    g = run_reference_game(
        perm, contract, declarer, king,
        lambda groups, k: 0,
        lambda hand, k: S.bot_discards(key, hand, k),
        lambda step, seat, mask: S.policy_action(key, step, mask),
        "direct")
    g.update(perm=perm, contract=contract, declarer=declarer, king=king, flow="direct")
    return g


def _synth_chunk(args):
    ref, seed, lo, hi, episode, mix = args
    load_reference(ref)
    return pack_games([synth_game(seed, g, episode, mix) for g in range(lo, hi)])


def digest_arrays(p):
    h = hashlib.sha256()
    for name in ("nsteps", "seats", "masks", "actions", "scores"):
        h.update(np.ascontiguousarray(p[name]).tobytes())
    return h.hexdigest()


def gen_synth(ref, seed, n, episode, mix, procs):
    import multiprocessing as mp
    step = max(1, min(512, n // max(1, procs)))
    jobs = [(ref, seed, lo, min(n, lo + step), episode, mix) for lo in range(0, n, step)]
    if procs > 1:
        with mp.Pool(procs) as pool:
            parts = pool.map(_synth_chunk, jobs)
    else:
        parts = [_synth_chunk(j) for j in jobs]
    p = {k: np.concatenate([q[k] for q in parts]) for k in parts[0]}
    return p


def micro_vectors():
    K = REF["Karta"]
    Rk = REF["Roka"].Roka
    full = [K.Karta.iz_id(i) for i in range(54)]
    vec = {}
    vec["prestej_full_deck"] = Rk.prestej(full)                                  # 70
    vec["discardable_ids"] = sorted(k.v_id() for k in Rk(full).mozno_zalozit())
    vec["vrednost_by_id"] = [K.Karta.iz_id(i).vrednost() for i in range(54)]
    vec["single_card_prestej"] = [Rk.prestej([K.Karta.iz_id(i)]) for i in range(54)]
    vec["v_id_roundtrip"] = all(K.Karta.iz_id(i).v_id() == i for i in range(54))
    # random piles: (mask, prestej)
    rnd = random.Random(7)
    piles = []
    for _ in range(400):
        n = rnd.randrange(0, 55)
        ids = rnd.sample(range(54), n)
        rnd.shuffle(ids)
        piles.append([str(sum(1 << i for i in ids)), Rk.prestej([K.Karta.iz_id(i) for i in ids])])
    vec["random_piles"] = piles
    # legal-move micro cases: (hand mask, lead id or -1) -> (navadna mask, klop mask)
    kl = REF["Klop"].Klop([], [], 0)
    nv = REF["Navadna_igra"].Navadna_igra.__new__(REF["Navadna_igra"].Navadna_igra)
    cases = []
    for _ in range(600):
        n = rnd.randrange(1, 13)
        ids = rnd.sample(range(54), n)
        if rnd.random() < 0.4 and 32 not in ids:
            ids[0] = 32
        lead = rnd.choice([-1] + [i for i in range(54) if i not in ids])
        hand = Rk([K.Karta.iz_id(i) for i in ids])
        spodnja = None if lead < 0 else K.Karta.iz_id(lead)
        mk = cards_mask(kl.mozne_karte(spodnja, hand))
        mn = cards_mask(nv.mozne_karte(spodnja, hand))
        cases.append([str(sum(1 << i for i in ids)), lead, str(mn), str(mk)])
    vec["legal_cases"] = cases
    # trick-winner micro cases: 4 ids -> winner offset (Klop.py:81-94)
    tw = []
    for _ in range(400):
        ids = rnd.sample(range(54), 4)
        tw.append(ids + [kl.pobere_stih([K.Karta.iz_id(i) for i in ids])])
    vec["trick_cases"] = tw
    return vec


# --------------------------------------------------------------------------
# Tarok.paralel_start runs (Tarok.py:30-62) -- adapter fixture
# --------------------------------------------------------------------------
class ParalelCtl:
    """Controller shared by the 4 global players of a Tarok.paralel_start run.
    Per game id it forces (contract, declarer seat, king) and plays the spec's
    random policy keyed by the game id, so the run is reproducible without
    recording per-step data."""

    def __init__(self, seed, mix, perms):
        self.seed, self.mix, self.perms = seed, mix, perms
        self.setup = {}
        self.steps = {}
        self.scores = {}

    def key(self, gid):
        return S.game_key(self.seed, gid, 0)

    def on_nova_igra(self, p, roka, gid):
        if gid not in self.setup:
            c, d, k = S.sample_setup(self.key(gid), self.mix)
            if not igra_reachable(c, d):
                d = 0
            self.setup[gid] = (c, d, k)
            self.steps[gid] = 0

    def bid(self, p, gid, min_igra, obvezno, prednost):
        T = REF["Tip_igre"].Tip_igre
        c, d, k = self.setup[gid]
        if p.seat[gid] == d and c != S.KLOP:
            return T(c * 10)
        return T.Naprej

    def king(self, p, gid):
        return self.setup[gid][2]

    def on_konec_licitiranja(self, *a):
        pass

    def choose_group(self, p, gid, groups, k):
        return 0

    def choose_discards(self, p, gid, hand, k):
        return S.bot_discards(self.key(gid), hand, k)

    def choose_card(self, p, gid, seat, mask, table):
        t = self.steps[gid]
        self.steps[gid] = t + 1
        return S.policy_action(self.key(gid), t, mask)

    def on_trick(self, *a, **k):
        pass

    def on_result(self, p, gid, pts):
        self.scores.setdefault(gid, {})[p.ime] = int(pts)


# the callbacks of the player protocol (Igralec.py:32-122), in the numbering of the recorded call log
CALLBACKS = ["nova_igra", "pripavi_licitiram", "predict_licitiram", "licitiram", "izberi_barvo_kralja", "konec_licitiranja",
             "pripravi_izbral_iz_talona", "predict_izberi_iz_talona", "menjaj_iz_talona", "izbral_iz_talona",
             "poglej_karte_odprtega_beraca", "pripravi_igraj_karto", "predict_igraj_karto", "igraj_karto", "rezultat_stiha",
             "rezultat_igre"]
# position of the game id among each callback's positional arguments (-1: a batch barrier, no game)
ID_ARG = [2, 0, -1, 1, 0, 2, 2, -1, 2, 2, 1, 3, -1, 3, 2, 2]


def logged(cls, log):
    """Subclass of a duck-typed player that appends (callback number, game id or -1, player name) to `log`
    before every protocol callback: the ORDER in which the reference's scheduler and engines call up."""
    def wrap(name, k):
        def f(self, *a, **kw):
            gid = -1 if ID_ARG[k] < 0 else int(kw["id_igre"] if "id_igre" in kw else a[ID_ARG[k]])
            log.append((k, gid, int(self.ime)))
            return getattr(cls, name)(self, *a, **kw)
        f.__name__ = name
        return f
    return type("Logged" + cls.__name__, (cls,), {name: wrap(name, k) for k, name in enumerate(CALLBACKS)})


def gen_paralel(seed, n_games, mix):
    R = REF
    perms = [S.deal(S.game_key(seed, g, 0)) for g in range(n_games)]
    it = iter(perms)

    def fake_shuffle(lst):
        lst[:] = next(it)           # Igra objects deal in id order (Tarok.py:36-38)
    R["Igra"].shuffle = fake_shuffle
    ctl = ParalelCtl(seed, mix, perms)
    calls = []
    players = [logged(TracePlayer, calls)(i, ctl) for i in range(4)]
    t = R["Tarok"].Tarok(players, n_games)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        t.paralel_start()
    totals = [int(t.rezultati[p]) for p in players]
    per_game = np.zeros((n_games, 4), np.int16)   # by GLOBAL player index
    for g in range(n_games):
        for i in range(4):
            per_game[g, i] = ctl.scores[g][str(i)]
    setup = np.array([ctl.setup[g] for g in range(n_games)], np.int8)
    return dict(seed=np.int64(seed), mix=np.int32(mix), deals=np.array(perms, np.uint8),
                setup=setup, per_game_scores=per_game, totals=np.array(totals, np.int64),
                calls=np.array(calls, np.int16), call_names=np.array(CALLBACKS))


# --------------------------------------------------------------------------
# Igra.licitacija call/response scripts (Igra.py:75-114) -- host bidding fixture
# --------------------------------------------------------------------------
class ScriptCtl:
    """Every licitiram call pops the next scripted desire of that seat; calls are logged."""

    def __init__(self, scripts):
        self.scripts = [list(x) for x in scripts]
        self.calls = []
        self.result = None

    def on_nova_igra(self, p, roka, gid):
        pass

    def bid(self, p, gid, min_igra, obvezno, prednost):
        T = REF["Tip_igre"].Tip_igre
        seat = p.seat[gid]
        want = self.scripts[seat].pop(0)
        self.calls.append([seat, int(min_igra), (None if obvezno is None else int(obvezno)), bool(prednost), want])
        return T(want)

    def king(self, p, gid):
        return 0

    def on_konec_licitiranja(self, p, decl, tip, gid, barva):
        self.result = [decl.seat[gid], int(tip)]


def gen_bidding(n_cases, seed):
    rnd = random.Random(seed)
    T = REF["Tip_igre"].Tip_igre
    vals = [int(t) for t in T]
    cases = []
    IgraMod = REF["Igra"]
    IgraMod.shuffle = lambda lst: None
    for n in range(n_cases):
        style = n % 3
        scripts = []
        for seat in range(4):
            if style == 0:      # the Bot's distribution (Igralec.py:151)
                sc = [rnd.choice([-10, -10, -10, 10, 20, 30]) for _ in range(40)]
            elif style == 1:    # anything goes
                sc = [rnd.choice(vals) for _ in range(40)]
            else:               # stubborn: constant desire
                sc = [rnd.choice(vals)] * 40
            scripts.append(sc)
        ctl = ScriptCtl(scripts)
        players = [TracePlayer(i, ctl) for i in range(4)]
        ig = IgraMod.Igra(players, multi_games=True, id=0)
        gen = ig.start()
        next(gen)                        # 'Pripravljen_licitirat'
        try:
            next(gen)                    # runs licitacija + konec_licitiranja, returns the engine generator
        except Exception as e:           # noqa: BLE001  (e.g. scripts exhausted: skip the case)
            continue
        cases.append(dict(scripts=[sc[:12] for sc in scripts], calls=ctl.calls, result=ctl.result))
        assert all(len(c) for c in ctl.calls) and max(len(ctl.calls), 0) < 40
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    ap.add_argument("--per-contract", type=int, default=288)
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--skip-large", action="store_true")
    ap.add_argument("--only-bidding", action="store_true")
    ap.add_argument("--only-paralel", action="store_true", help="regenerate paralel_v1.npz alone")
    args = ap.parse_args()
    load_reference(args.reference)
    os.makedirs(args.out, exist_ok=True)
    if args.only_paralel:
        par = gen_paralel(seed=5, n_games=96, mix=S.MIX_ALL)
        np.savez_compressed(os.path.join(args.out, "paralel_v1.npz"), **par)
        print("paralel totals:", par["totals"], "calls:", len(par["calls"]))
        return

    bid = gen_bidding(1500, seed=77)
    with open(os.path.join(args.out, "licitacija_v1.json"), "w") as f:
        json.dump(bid, f, separators=(",", ":"))
    print("bidding cases:", len(bid))
    if args.only_bidding:
        return

    tr = gen_traces(args.per_contract, seed=20261004)
    np.savez_compressed(os.path.join(args.out, "traces_v1.npz"), **tr)
    print("traces:", len(tr["contract"]), "games; digest", digest_arrays(tr))

    dig = {"format": "sha256 over nsteps|seats|masks|actions|scores (C-order, little-endian)",
           "micro": micro_vectors(), "synthetic": []}
    runs = [dict(name="config2_navadna3_4096", seed=0, n=4096, episode=0, mix=S.MIX_NAVADNA3),
            dict(name="mixed_4096_seed1", seed=1, n=4096, episode=0, mix=S.MIX_ALL),
            dict(name="mixed_2048_seed2_ep3", seed=2, n=2048, episode=3, mix=S.MIX_ALL),
            dict(name="bot_bidding_4096_seed4", seed=4, n=4096, episode=0, mix=S.MIX_BOT)]
    if not args.skip_large:
        runs.append(dict(name="config3_mixed_65536", seed=0, n=65536, episode=0, mix=S.MIX_ALL))
    for r in runs:
        p = gen_synth(args.reference, r["seed"], r["n"], r["episode"], r["mix"], args.procs)
        r = dict(r)
        r["sha256"] = digest_arrays(p)
        r["total_steps"] = int(p["nsteps"].sum())
        r["score_sums_by_seat"] = [int(x) for x in p["scores"].astype(np.int64).sum(0)]
        r["contract_hist"] = [int(x) for x in np.bincount(p["contract"], minlength=10)]
        dig["synthetic"].append(r)
        print(r["name"], r["sha256"], r["total_steps"])
        if r["name"] == "mixed_2048_seed2_ep3":
            # keep this small one in full as well: pins the spec RNG/deal/policy
            np.savez_compressed(os.path.join(args.out, "synth_small_v1.npz"), **p)
    with open(os.path.join(args.out, "digests_v1.json"), "w") as f:
        json.dump(dig, f, indent=1)

    par = gen_paralel(seed=5, n_games=96, mix=S.MIX_ALL)
    np.savez_compressed(os.path.join(args.out, "paralel_v1.npz"), **par)
    print("paralel totals:", par["totals"])


if __name__ == "__main__":
    main()
