"""Card / contract vocabulary of the reference (Karta.py, Tip_igre.py) as plain ints.

card id = suit*8 + rank-1 for suits KARA=0, SRCE=1, PIK=2, KRIZ=3 (rank 1..8, 8 = king);
taroks: id = 32 + n-1 for n = 1..22 (Karta.v_id, Karta.py:19-23).
contract code = int(Tip_igre) // 10 (Tip_igre.py:4-15).
"""
import enum


class Barva(enum.IntEnum):          # Karta.py:69-74
    KARA = 0
    SRCE = 1
    PIK = 2
    KRIZ = 3
    TAROK = 4


class Tip(enum.IntEnum):            # Tip_igre.py:4-15, value = int(Tip_igre)//10
    Klop = 0
    Tri = 1
    Dve = 2
    Ena = 3
    Solo_tri = 4
    Solo_dve = 5
    Solo_ena = 6
    Berac = 7
    Solo_brez = 8
    Odprti_berac = 9


DECK = (1 << 54) - 1
PAGAT, MOND, SKIS = 32, 52, 53

# synthetic contract mixes (include/tarok_env.h TAROK_MIX_*)
MIX_ALL = 0
MIX_NAVADNA3 = 1
MIX_BOT = 2
MIX_FIXED = 16

# flags (include/tarok_env.h)
DEFER_EXCHANGE = 1
AUTO_RESET = 2
CLEAR_COUNTERS = 4
REWARD_REF = 8           # step: reward_out = what rezultat_igre folds into the last transition (Igralec.py:421-437)
HISTORY = 16             # TarokVecEnv(history=True): keep the play history (needed by observe_ref)
OPT_REFILL_FAN = 2        # tarok_set_option
OPT_LAZY_REFILL = 3
GAMES_AHEAD = 14          # TAROK_GAMES_AHEAD: games every slot keeps dealt ahead for auto-reset
# the flat parameter vector of the 256-256-256-64 policy (include/tarok_env.h TAROK_MLP_*)
MLP_W1, MLP_B1, MLP_W2, MLP_B2, MLP_W3, MLP_B3, MLP_PARAMS = 0, 65536, 65792, 131328, 131584, 147968, 148032
LEARN_PAD = 256           # TAROK_LEARN_PAD: padding rows of the fused learner's activation arrays

# the reference-layout observation record (include/tarok_env.h TAROK_REF_*)
REF_ROWS = 56
REF_OPP, REF_OWN, REF_TALON, REF_KING, REF_INDEX, REF_DISCARDS, REF_LEGAL = 0, 9072, 12096, 12426, 12430, 12434, 12488
REF_RECORD_BYTES = 12544
REF_EXCHANGE_BYTES = 400
REF_TYPES = ("Klop", "Navadna_igra", "Solo", "Berac")    # meta[:, 1] (Nevronski_igralec.Tipi_NN, Igralec.py:174-178)

# observation word
OBS_MASK = DECK
OBS_SEAT_SHIFT = 54
OBS_STEP_SHIFT = 56
OBS_DONE_BIT = 62
OBS_ERROR_BIT = 63

PHASE_EXCHANGE, PHASE_PLAY, PHASE_DONE = 1, 2, 3


def card_id(barva, st):
    """Karta(barva, st).v_id()  (Karta.py:19-23)"""
    return 32 + st - 1 if int(barva) == 4 else int(barva) * 8 + st - 1


def card_from_id(cid):
    """Karta.iz_id (Karta.py:32-47) -> (barva, st)"""
    return (Barva.TAROK, cid - 31) if cid > 31 else (Barva(cid // 8), cid % 8 + 1)


def card_name(cid):
    """str(Karta) (Karta.py:52-57)"""
    b, st = card_from_id(cid)
    m = {5: "J", 6: "K", 7: "D", 8: "KR"}[st] if b != Barva.TAROK and st > 4 else st
    return "%s_%s" % (b.name, m)


def mask_to_ids(mask):
    return [i for i in range(54) if (int(mask) >> i) & 1]


def ids_to_mask(ids):
    m = 0
    for i in ids:
        m |= 1 << int(i)
    return m
