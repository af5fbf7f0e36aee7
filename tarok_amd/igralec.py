"""The reference's player ("plugin") protocol served from the GPU environment.

The reference engines are cooperative generators that call UP into duck-typed players
(`Igralec`, Igralec.py:32-122) one game at a time; `Tarok.paralel_start` (Tarok.py:30-62)
lock-steps N of them with batched `predict_*` barriers in between.  This module keeps
that surface — same class and callback names, same argument meaning, same call order,
same exception on an illegal card — while every rule (legal cards, tricks, scoring) is
evaluated by libtarokenv on the device:

    Karta, Roka          value types the callbacks receive (Karta.py, Roka.py interface)
    Igralec              base player: owns roka[id] / kupcek[id]  (Igralec.py:32-122)
    Bot_igralec          uniform-random player                     (Igralec.py:142-171)
    Tarok(...).paralel_start()   the N-game lock-step scheduler    (Tarok.py:30-62)

A reference-shaped agent (one that implements the callbacks) can be passed to `Tarok`
unchanged.  It is the compatibility surface, not the fast one: the callbacks are per-game
Python calls.  Agents that want throughput talk to `TarokVecEnv` in batches.

Call order.  Within every game the callbacks arrive in the reference's order, and so do the batch
barriers (`predict_*`) between them — pinned by the call log recorded from the reference's own
`Tarok.paralel_start` (tests/golden/paralel_v1.npz `calls`,
test_paralel_start_adapter_matches_reference_run).  Two differences, both on purpose:
  * ACROSS games the interleaving follows the batch: the cards of all games are collected
    (`igraj_karto` of game 0, 1, 2 ...), the device steps them together, then the results are handed
    out game by game; the reference finishes `igraj_karto -> rezultat_* -> pripravi_igraj_karto` of one
    game before it touches the next (Tarok.py:52-56).  A player that keys its state by game id — every
    player of the reference does — cannot tell.
  * a Solo_brez game is driven in the same phase as every other game.  In the reference it runs one
    `next()` ahead of the scheduler because its generator skips the 'Pripravljen menjat' yield
    (Navadna_igra.py:48,67-68): its callbacks come in the same order, but one batch barrier early.
"""
import random
import warnings

import numpy as np

from . import karte as K
from . import licitacija as L
from .karte import Barva


class Karta:
    """A card: `barva` (Barva) and `st` (1..8 in a suit, 8 = king; 1..22 for taroks)."""
    __slots__ = ("barva", "st")

    def __init__(self, barva, st):
        self.barva = Barva(int(barva))
        self.st = int(st)

    def vrednost(self):                       # Karta.py:10-16 (discard-filter value, NOT pile points)
        if self.barva == Barva.TAROK and self.st in (1, 21, 22):
            return 5
        return self.st - 3 if self.st > 4 else 1

    def v_id(self):                           # Karta.py:19-23
        return K.card_id(self.barva, self.st)

    @staticmethod
    def iz_id(cid):                           # Karta.py:32-47
        b, st = K.card_from_id(int(cid))
        return Karta(b, st)

    def __eq__(self, other):
        return isinstance(other, Karta) and (self.barva, self.st) == (other.barva, other.st)

    def __hash__(self):
        return self.v_id()

    def __lt__(self, other):                  # Karta.py:62-67
        return (self.barva, self.st) < (other.barva, other.st)

    def __str__(self):
        return K.card_name(self.v_id())

    __repr__ = __str__


def _cards(mask):
    return [Karta.iz_id(i) for i in K.mask_to_ids(mask)]


def _mask(cards):
    return K.ids_to_mask(k.v_id() for k in cards)


_PILE_VALUE = [1] * 54
for _i in (7, 15, 23, 31, 32, 52, 53):
    _PILE_VALUE[_i] = 5
for _b in range(4):
    _PILE_VALUE[_b * 8 + 4], _PILE_VALUE[_b * 8 + 5], _PILE_VALUE[_b * 8 + 6] = 2, 3, 4


class Roka:
    """A hand: `karte` = {Barva: sorted list of Karta}  (Roka.py:5-50)."""

    def __init__(self, karte):
        self.karte = {b: [] for b in Barva}
        for k in karte:
            self.karte[k.barva].append(k)
        for v in self.karte.values():
            v.sort()

    def igraj_karto(self, k):
        self.karte[k.barva].remove(k)

    def dodaj_karte(self, karte):
        for k in karte:
            self.karte[k.barva].append(k)

    def mozno_zalozit(self):                  # Roka.py:23-27
        return [k for v in self.karte.values() for k in v if k.vrednost() < 5]

    def __contains__(self, karta):
        return isinstance(karta, Karta) and karta in self.karte[karta.barva]

    def __iter__(self):
        for v in self.karte.values():
            yield from v

    def __len__(self):
        return sum(len(v) for v in self.karte.values())

    def __str__(self):
        return str(sorted(self))

    __repr__ = __str__

    @staticmethod
    def vrednost_stiha(stih):                 # Roka.py:76-95
        v = sum(_PILE_VALUE[k.v_id()] for k in stih)
        return v - 1 if len(stih) in (1, 2) else v - 2

    @staticmethod
    def prestej(kupcek):                      # Roka.py:56-60,97-98
        n = len(kupcek)
        return sum(_PILE_VALUE[k.v_id()] for k in kupcek) - 2 * (n // 3) - (1 if n % 3 else 0)


_counter = 0


class Igralec:
    """Base player (Igralec.py:32-122): keeps its own hand and won pile per game id and
    answers the env's callbacks.  Subclasses decide; the env validates."""

    def __init__(self, ime=None):
        global _counter
        if ime is None:
            ime = _counter
            _counter += 1
        self.ime = str(ime)
        self.roka, self.igra, self.kupcek = {}, {}, {}

    def nova_igra(self, roka, igralci, id_igre):
        self.roka[id_igre] = roka
        self.igra[id_igre] = None
        self.kupcek[id_igre] = []

    def pripavi_licitiram(self, id_igre):
        pass

    def predict_licitiram(self):
        pass

    def licitiram(self, licitiram, min_igra, id_igre, obvezno=None, prednost=False):
        return L.base_filter(int(licitiram), int(min_igra), None if obvezno is None else int(obvezno), prednost)

    def izberi_barvo_kralja(self, id_igre):
        raise NotImplementedError()

    def konec_licitiranja(self, igralec_ki_igra, tip_igre, id_igre, barva_kralja=None):
        pass

    def pripravi_izbral_iz_talona(self, talon, st_kupcka, id_igre):
        pass

    def predict_izberi_iz_talona(self):
        pass

    def menjaj_iz_talona(self, kupcki, st_kart, id_igre):
        raise NotImplementedError()

    def izbral_iz_talona(self, talon, st_kupcka, id_igre):
        pass

    def poglej_karte_odprtega_beraca(self, roka, id_igre):
        warnings.warn("Ne uporablam podatka za odprtega beraca")

    def pripravi_igraj_karto(self, karte_na_mizi, mozne, zgodovina, id_igre):
        pass

    def predict_igraj_karto(self):
        pass

    def igraj_karto(self, karta, id_igre):
        self.roka[id_igre].igraj_karto(karta)
        return karta

    def rezultat_stiha(self, stih, sem_pobral, id_igre):
        pass

    def rezultat_igre(self, st_tock, povzetek_igre, id_igre):
        pass

    def __str__(self):
        return "Igralec_" + self.ime

    __repr__ = __str__


class Bot_igralec(Igralec):
    """Uniform-random player (Igralec.py:142-171)."""

    def __init__(self, ime=None, rng=None):
        super().__init__(ime)
        self.rng = rng or random.Random()

    def licitiram(self, min_igra, id_igre, obvezno=None, prednost=False):
        zelim = self.rng.choices([L.NAPREJ, L.TRI, L.DVE, L.ENA], weights=[3, 1, 1, 1])[0]
        return super().licitiram(zelim, min_igra, id_igre, obvezno, prednost)

    def izberi_barvo_kralja(self, id_igre):
        return self.rng.choice([Barva.SRCE, Barva.KRIZ, Barva.KARA, Barva.PIK])

    def igraj_karto(self, karte_na_mizi, mozne, zgodovina, id_igre):
        return super().igraj_karto(self.rng.choice(mozne), id_igre)

    def menjaj_iz_talona(self, kupcki, st_kart, id_igre):
        self.roka[id_igre].dodaj_karte(kupcki[0])
        izberi = self.rng.sample(self.roka[id_igre].mozno_zalozit(), k=st_kart)
        self.kupcek[id_igre].extend(izberi)
        for k in izberi:
            self.roka[id_igre].igraj_karto(k)
        return 0


def _call(obj, name, *args):
    fn = getattr(obj, name, None)
    return fn(*args) if fn is not None else None


class Tarok:
    """N games in lock-step on one GPU behind the reference's scheduler surface
    (Tarok.py:9-62): `Tarok(igralci, st_iger).paralel_start()`; scores accumulate in
    `self.rezultati[player]`.  `seed`/`episode` pick the deals (the reference deals with
    an unseeded random.shuffle); pass `deals=[N,54]` to inject permutations instead."""

    def __init__(self, igralci, st_iger=None, seed=0, device=0, deals=None):
        assert len({i.ime for i in igralci}) == 4
        self.igralci = list(igralci)
        self.rezultati = {i: 0 for i in igralci}
        self.st_iger = st_iger
        self.seed, self.device, self.deals = seed, device, deals
        self.episode = 0
        self.zadnje_igre = None      # per-game (declarer seat, contract value, scores by seat) of the last batch

    # -- helpers ----------------------------------------------------------
    @staticmethod
    def _tip(value):
        """Contracts are handed to players as ints equal to int(Tip_igre); a player that
        wants the enum can wrap them."""
        return int(value)

    def paralel_start(self):
        from .env import TarokVecEnv
        n = int(self.st_iger)
        env = TarokVecEnv(n, device=self.device, seed=self.seed, mix=K.MIX_FIXED + int(K.Tip.Klop))
        try:
            return self._play(env, n)
        finally:
            env.close()

    def _play(self, env, n):
        seats = [self.igralci[g % 4:] + self.igralci[:g % 4] for g in range(n)]          # Tarok.py:34
        # --- deal (Igra.razdeli, Igra.py:65-73): on device, read back once
        if self.deals is None:
            env.reset(episode=self.episode)
            st = env.state()
            deals = np.zeros((n, 54), np.uint8)
            for g in range(n):
                row = []
                for s in range(4):
                    row += K.mask_to_ids(int(st[s, g]))
                row += [(int(st[8, g]) >> (6 * i)) & 63 for i in range(6)]
                deals[g] = row
        else:
            deals = np.asarray(self.deals, np.uint8)
        talon = [[Karta.iz_id(c) for c in deals[g, 48:54]] for g in range(n)]
        for g in range(n):                                                               # Tarok.py:36-38, one game at a time:
            for s, p in enumerate(seats[g]):                                              # Igra.razdeli (Igra.py:68-72) ...
                p.nova_igra(Roka([Karta.iz_id(c) for c in deals[g, 12 * s:12 * s + 12]]), seats[g], g)
            for p in seats[g]:                                                            # ... then Igra.licitacija's first lines (Igra.py:77-79)
                _call(p, "pripavi_licitiram", g)
        # --- bidding (Igra.licitacija, Igra.py:75-114), batch barrier 1 (Tarok.py:38-40)
        for p in self.igralci:
            _call(p, "predict_licitiram")
        contract = np.zeros(n, np.int8)
        declarer = np.zeros(n, np.int8)
        king = np.zeros(n, np.int8)
        for g in range(n):
            def ask(seat, min_igra, obvezno, prednost, g=g):
                return int(seats[g][seat].licitiram(self._tip(min_igra), g, None if obvezno is None else self._tip(obvezno), prednost))
            d, value = L.licitacija(ask)
            if value not in L.ALL_BIDS or value == L.NAPREJ:
                raise Exception("Igra ni definirana:" + str(value))                       # Igra.py:55
            barva = None
            if value in (L.TRI, L.DVE, L.ENA):
                barva = Barva(int(seats[g][d].izberi_barvo_kralja(g)))                   # Igra.py:43
                assert barva != Barva.TAROK                                               # Navadna_igra.py:21
                king[g] = int(barva)
            contract[g], declarer[g] = value // 10, d
            for p in seats[g]:
                p.konec_licitiranja(seats[g][d], self._tip(value), g, barva)              # Igra.py:57-58
        obs = env.reset(episode=self.episode, deals=deals, contract=contract, declarer=declarer, king_suit=king,
                        defer_exchange=True)
        zgodovina = [[] for _ in range(n)]
        # --- talon exchange (Navadna_igra.py:36-66), batch barrier 2 (Tarok.py:42-45)
        ex = [g for g in range(n) if 1 <= contract[g] <= 6]
        groups = {}
        for g in ex:
            gs = 3 - ((int(contract[g]) - 1) % 3)
            groups[g] = ([talon[g][i:i + gs] for i in range(0, 6, gs)], gs)
            _call(seats[g][declarer[g]], "pripravi_izbral_iz_talona", [list(x) for x in groups[g][0]], gs, g)
        for p in self.igralci:
            _call(p, "predict_izberi_iz_talona")
        choice = np.zeros(n, np.int8)
        discards = np.full((n, 3), 255, np.uint8)
        for g in ex:
            kupcki, gs = groups[g]
            igr = seats[g][declarer[g]]
            before = _mask(igr.roka[g])
            idx = int(igr.menjaj_iz_talona([list(x) for x in kupcki], gs, g))            # the PLAYER moves the cards
            after = _mask(igr.roka[g])
            gone = K.mask_to_ids((before | _mask(kupcki[idx])) & ~after)
            if len(gone) != gs:
                raise Exception("%s zalozil %d kart namesto %d" % (igr, len(gone), gs))
            choice[g] = idx
            discards[g, :gs] = gone
            zgodovina[g].append(("Talon", (idx, [list(x) for x in kupcki])))
            for p in seats[g]:
                _call(p, "izbral_iz_talona", [list(x) for x in kupcki], idx, g)
        if ex:
            obs = env.exchange(talon_choice=choice, discards=discards)
        # --- 48 lock-steps (Tarok.py:48-56)
        stih = [[] for _ in range(n)]
        klop_talon = [list(talon[g]) if contract[g] == 0 else [] for g in range(n)]
        leader = [int(declarer[g]) if contract[g] in (7, 9) else 0 for g in range(n)]
        koncano = [None] * n
        for korak in range(48):
            mask = obs.mask_numpy()
            seat = obs.seat.cpu().numpy()
            live = [g for g in range(n) if koncano[g] is None]
            if not live:
                break
            mozne = {}
            for g in live:
                if contract[g] == 9 and korak == 4:                                       # Berac.py:22-25
                    berac = seats[g][declarer[g]]
                    for p in seats[g]:
                        if p is not berac:
                            _call(p, "poglej_karte_odprtega_beraca", berac.roka, g)
                mozne[g] = _cards(int(mask[g]))
                seats[g][seat[g]].pripravi_igraj_karto(list(stih[g]), mozne[g], zgodovina[g], g)
            for p in self.igralci:
                _call(p, "predict_igraj_karto")
            action = np.full(n, 255, np.uint8)
            for g in live:
                igr = seats[g][seat[g]]
                karta = igr.igraj_karto(list(stih[g]), mozne[g], zgodovina[g], g)
                if not isinstance(karta, Karta) or not (int(mask[g]) >> karta.v_id()) & 1:
                    raise Exception(str(igr) + str(igr.__class__) + " Karte ne mores igarti. Karta: " + str(karta)
                                    + " karte na mizi:" + str(stih[g]) + " Mozne" + str(mozne[g]))   # Klop.py:57-60
                action[g] = karta.v_id()
                zgodovina[g].append((igr, karta))
                stih[g].append(karta)
            obs, reward, done = env.step(action)
            if obs.error.any().item():
                raise Exception("okolje je zavrnilo karto")
            if korak % 4 == 3:                                                            # trick complete
                nseat = obs.seat.cpu().numpy()
                fin = done.cpu().numpy().astype(bool)
                rew = reward.cpu().numpy()
                for g in live:
                    if klop_talon[g]:                                                     # Klop.py:67-71
                        t = klop_talon[g].pop()
                        stih[g].append(t)
                        zgodovina[g].append((None, t))
                    zmagovalec = int(nseat[g])
                    kup = getattr(seats[g][zmagovalec], "kupcek", None)
                    if isinstance(kup, dict) and g in kup:
                        kup[g].extend(stih[g])
                    for s, p in enumerate(seats[g]):
                        p.rezultat_stiha(list(stih[g]), s == zmagovalec, g)
                    stih[g] = []
                    leader[g] = zmagovalec
                    if fin[g]:
                        koncano[g] = [int(x) for x in rew[g]]
                        for s, p in enumerate(seats[g]):
                            p.rezultat_igre(koncano[g][s], zgodovina[g], g)
        # --- totals (Tarok.py:59-61)
        for g in range(n):
            for s, p in enumerate(seats[g]):
                self.rezultati[p] += koncano[g][s]
        self.zadnje_igre = [(int(declarer[g]), int(contract[g]) * 10, koncano[g]) for g in range(n)]
        self.episode += 1
        return self.rezultati
