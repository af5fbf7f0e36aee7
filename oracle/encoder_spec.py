"""TEST INFRASTRUCTURE — CPU restatement of the reference agent's observation encoders and of what
`rezultat_igre` folds into the last transition, each function citing the lines of
/root/reference/Igralec.py it follows.  Only tests/ may import this; the product never does.

PARITY UNPINNED: the reference holds no fixture for these tensors and its `Igralec.py` cannot be
imported here (`pytorch_lightning` and the reference's own missing `torch_models`), so nothing the
reference itself produced pins this file.  It is a restatement written from the text of the source;
the device kernels (`tarok_observe_ref` ...) are tested against it, and it against hand-worked cases.

Vocabulary kept from the reference: `zgodovina` is the engines' history list — (player, card) per
card played (Klop.py:63, Navadna_igra.py:127), (None, card) for a talon card that Klop adds to a trick
(Klop.py:71) and ("Talon", (chosen group, groups)) once after the talon exchange (Navadna_igra.py:63).
Here players are seats 0..3 and cards are ids 0..53.
"""
import numpy as np

TALON = "Talon"

# Nevronski_igralec.tip_igre_v_tip_izbire (Igralec.py:180-189), keyed by contract code = int(Tip_igre)//10
TIP_IZBIRE = {0: "Klop", 1: "Navadna_igra", 2: "Navadna_igra", 3: "Navadna_igra", 4: "Solo", 5: "Solo", 6: "Solo",
              7: "Berac", 8: "Solo", 9: "Berac"}
TIPI_NN = {"Klop": 0, "Navadna_igra": 1, "Solo": 2, "Berac": 3}          # Igralec.py:174-178
GROUP = {1: 3, 2: 2, 3: 1, 4: 3, 5: 2, 6: 1}                             # odpri_talon's korak (Navadna_igra.py:36-44)


def igralci2index(me):
    """nova_igra (Igralec.py:271-274): the other players in the order of the game's `igralci` list get
    0, 1, 2; the player himself is 3."""
    d = {}
    for s in range(4):
        if s != me:
            d[s] = len(d)
    d[me] = 3
    return d


def history_length(zgodovina):
    """Igralec.py:455-460.  The loop's condition `ig is not None or ig != 'Talon'` holds for every entry
    (None is != 'Talon'; anything else is not None), so every entry counts — talon entries included;
    then the next multiple of 8, a full 8 more when the count already is one."""
    n = 0
    for ig, _k in zgodovina:
        if ig is not None or ig != TALON:
            n += 1
    return n + (8 - n % 8)


def stanje_v_vektor(me, tip, zgodovina, zacetna_roka, zalozil, barva_kralja, declarer, mozne):
    """stanje_v_vektor_rek_navadna (Igralec.py:453-533) for player `me`.
    zacetna_roka: ids of the hand as dealt (:264); zalozil: ids the player discarded or None (:462-464);
    barva_kralja: suit or None (:467-470); mozne: legal ids.  Returns the list the reference returns,
    every array with the leading batch axis of :521-531."""
    T = history_length(zgodovina)
    roka = np.zeros(54)
    zal = np.zeros(54)
    if zalozil is not None:
        zal[list(zalozil)] = 1
    roka[list(zacetna_roka)] = 1
    kralj = np.zeros(4)
    if barva_kralja is not None:
        kralj[int(barva_kralja)] = 1
    roka_input = np.zeros((T, 54))                                       # :473-474 (T is never 0)
    nasprotniki = np.zeros((T, 3, 54))
    if tip in ("Navadna_igra", "Solo"):                                  # :480-487
        talon_input = np.zeros((6, 55))
    elif tip == "Klop":
        talon_input = np.zeros((54,))
    elif tip == "Berac":
        talon_input = None
    else:
        raise Exception("Ni implementerano" + str(tip))
    index = np.zeros(4)
    i2i = igralci2index(me)
    index[i2i[declarer]] = 1                                             # :493-494 with :449-451
    i = 0
    for igralec, k in zgodovina:                                         # :497-516
        if igralec is None:
            talon_input[k] = 1
        elif igralec == TALON:
            stevilka_kupcka, kupcki = k
            r = 0
            for st_kupcka, kup in enumerate(kupcki):
                for karta in kup:
                    talon_input[r, karta] = 1
                    if st_kupcka == stevilka_kupcka:
                        talon_input[r, 54] = 1
                    r += 1
        elif igralec != me:
            nasprotniki[i, i2i[igralec], k] = 1
            i += 1
        else:
            roka_input[i, :] = roka
            roka[k] = 0
            i += 1
    mozne_vec = np.zeros(54)
    mozne_vec[list(mozne)] = 1
    if tip == "Navadna_igra":                                            # :520-531
        r = [nasprotniki, kralj, roka_input, talon_input, index, zal, mozne_vec]
    elif tip == "Solo":
        r = [nasprotniki, roka_input, talon_input, index, zal, mozne_vec]
    elif tip == "Klop":
        r = [nasprotniki, roka_input, talon_input, mozne_vec]
    else:
        r = [nasprotniki, roka_input, index, mozne_vec]
    return [np.expand_dims(x, axis=0) for x in r]


def menjaj_talon_v_vektor(roka, kupcki, contract, barva_kralja):
    """Igralec.py:535-543: [roka (1,54), talon (1,54,6), igra (1,15)] (the fourth, np.empty((1,1)), carries
    nothing).  igra's index is igra_zalozi2index (Igralec.py:717-745): contracts in ascending Tip_igre
    order, Tri/Dve/Ena once per suit in ascending Barva order, then Solo_tri, Solo_dve, Solo_ena."""
    r = np.zeros((1, 54))
    talon = np.zeros((1, 54, 6))
    igra = np.zeros((1, 15))
    igra[0, (contract - 1) * 4 + int(barva_kralja) if contract <= 3 else 12 + (contract - 4)] = 1
    r[0, list(roka)] = 1
    for i, k in enumerate(kupcki):
        talon[0, list(k), i] = 1
    return [r, talon, igra]


def rezultat_igre_st_tock(st_tock, tip, is_declarer, cards_in_hand):
    """Igralec.py:433-438: the final reward written into a player's last transition — his score, except for
    a Berac defender: -20 with an empty hand at the end (all twelve tricks played), +20 otherwise."""
    if tip == "Berac" and not is_declarer:
        return -20 if cards_in_hand == 0 else 20
    return st_tock


def rezultat_stiha_dy(mozne_vec, igrana_karta, vrednost_stiha, sem_pobral, tip_igre_dict=None):
    """Igralec.py:387-419, the training target of ONE transition before the game's end is folded in:
    `dy` = -70 on every card that was not legal at the player's decision (:392-393; `self.stanje[id][-1]` is
    `mozne_vec`), and on the card played the trick's value (Roka.vrednost_stiha of the 4 — Klop: 5 — cards)
    with the sign of `sem_pobral` (:412-416).  The Klop branch (:394-398) and the Berac branch (:399-409)
    are DEAD: they compare `self.tip_igre`, which is a dict keyed by game id (:200,258), with a string, so the
    comparison is False for every contract and the `else` of :411 always runs — kept here as the reference has
    them, with the same comparison (tip_igre_dict: the dict; any value gives the same result)."""
    dy = np.zeros(54)
    dy[np.asarray(mozne_vec) == 0] = -70
    tip_igre = {} if tip_igre_dict is None else tip_igre_dict
    if tip_igre == "Klop":                                  # never true (:394)
        dy[igrana_karta] = -vrednost_stiha if sem_pobral else vrednost_stiha
    elif tip_igre == "Berac":                               # never true (:399)
        dy[igrana_karta] = -1 if sem_pobral else 1
    else:
        dy[igrana_karta] = vrednost_stiha if sem_pobral else -vrednost_stiha
    return dy


def rezultat_igre_dy(transitions, st_tock, final_reword_factor):
    """Igralec.py:417-418,439-442.  `transitions` = the player's [dy, igrana_karta, next_Q_max] in play order,
    next_Q_max = the value the agent stored at its NEXT decision (:351: the predicted Q of the card its network
    ranks first), which rezultat_stiha writes into the previous transition (:417-418); the last transition gets
    the final reward instead (:439: st_tock after the Berac rule of :434-437, rezultat_igre_st_tock).  Then every
    target's played card gains next_max * final_reword_factor (:441).  Returns the list of dy."""
    out = []
    for k, (dy, karta, next_q) in enumerate(transitions):
        next_max = st_tock if k == len(transitions) - 1 else next_q
        dy = np.array(dy, dtype=np.float64)
        dy[karta] = dy[karta] + next_max * final_reword_factor
        out.append(dy)
    return out


# ---------------------------------------------------------------------------
# helpers for the tests: the reference's history of a recorded game, and the device record layout
# ---------------------------------------------------------------------------
def zgodovina_of(deal, contract, choice, seats, actions, upto):
    """The engines' history list when card number `upto` of the game is about to be played:
    Navadna_igra.py:63 (the Talon entry), :127 / Klop.py:63 (cards), Klop.py:67-71 (talon.pop() after each
    of the first six tricks).  deal = the 54-permutation (talon = deal[48:54], Igra.py:68)."""
    z = []
    talon = [int(c) for c in deal[48:54]]
    if contract in GROUP:
        gs = GROUP[contract]
        z.append((TALON, (int(choice), [talon[j:j + gs] for j in range(0, 6, gs)])))
    klop_talon = list(talon) if contract == 0 else []
    for p in range(upto):
        z.append((int(seats[p]), int(actions[p])))
        if p % 4 == 3 and klop_talon:
            z.append((None, klop_talon.pop()))
    return z


def pack_record(me, tip, lists):
    """The restated tensors in the layout of tarok_observe_ref's record (include/tarok_env.h TAROK_REF_*):
    12,544 bytes, history tensors padded with zero rows to 56."""
    rec = np.zeros(12544, np.uint8)
    names = {"Navadna_igra": ["opp", "king", "own", "talon", "index", "disc", "legal"],
             "Solo": ["opp", "own", "talon", "index", "disc", "legal"],
             "Klop": ["opp", "own", "talon", "legal"], "Berac": ["opp", "own", "index", "legal"]}[tip]
    d = {k: v[0] for k, v in zip(names, lists)}
    T = d["opp"].shape[0]
    opp = np.zeros((56, 3, 54), np.uint8); opp[:T] = d["opp"]
    own = np.zeros((56, 54), np.uint8); own[:T] = d["own"]
    rec[0:9072] = opp.reshape(-1)
    rec[9072:12096] = own.reshape(-1)
    if "talon" in d:
        t = d["talon"].reshape(-1)
        rec[12096:12096 + len(t)] = t
    if "king" in d:
        rec[12426:12430] = d["king"]
    if "index" in d:
        rec[12430:12434] = d["index"]
    if "disc" in d:
        rec[12434:12488] = d["disc"]
    rec[12488:12542] = d["legal"]
    return rec, T
