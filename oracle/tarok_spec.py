"""TEST INFRASTRUCTURE — not product code.

Pure-Python statement of the build's *own* synthetic-input generators (the
reference has none: it deals with an unseeded ``random.shuffle``, Igra.py:10,67,
and its Bot picks cards with ``random.choice``, Igralec.py:159).  Everything in
here is integer arithmetic on Python ints, so it is the slow-but-obvious form of
what ``oracle/tarok_oracle.c`` and the HIP kernels compute:

  * counter-based RNG keyed by (seed, global game index, episode)
  * the deal: sort the 54 cards by a per-card random key   (-> Igra.razdeli's perm)
  * the synthetic contract mixes of BASELINE.md configs 2 and 3
  * the Bot talon exchange (group 0 + random discards, Igralec.py:161-171)
  * the uniform random card policy (Igralec.py:158-159)

Only ``tests/``, ``oracle/gen_golden.py`` and other checkers may import this.
"""

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1

# ---- card constants (Karta.py:19-23: id = suit*8 + rank-1, taroks 32..53) ----
DECK = (1 << 54) - 1
SUIT = [0xFF << (8 * c) for c in range(4)] + [((1 << 22) - 1) << 32]
TAROK = SUIT[4]
PAGAT = 1 << 32


def _bits(ids):
    m = 0
    for i in ids:
        m |= 1 << i
    return m


V5 = _bits([7, 15, 23, 31, 32, 52, 53])   # kings + trula      (Roka.py:76-95)
V4 = _bits([6, 14, 22, 30])               # queens
V3 = _bits([5, 13, 21, 29])               # knights
V2 = _bits([4, 12, 20, 28])               # jacks
# Roka.mozno_zalozit (Roka.py:23-27) with Karta.vrednost (Karta.py:10-16)
DISCARDABLE = _bits(list(range(0, 7)) + list(range(8, 15)) + list(range(16, 23))
                    + list(range(24, 31)) + list(range(33, 39)))

# contract code = int(Tip_igre) // 10   (Tip_igre.py:4-15)
KLOP, TRI, DVE, ENA, SOLO_TRI, SOLO_DVE, SOLO_ENA, BERAC, SOLO_BREZ, ODPRTI_BERAC = range(10)
CONTRACT_NAMES = ["Klop", "Tri", "Dve", "Ena", "Solo_tri", "Solo_dve", "Solo_ena",
                  "Berac", "Solo_brez", "Odprti_berac"]
GROUP_SIZE = [0, 3, 2, 1, 3, 2, 1, 0, 1, 0]    # Navadna_igra.py:36-44 (korak)
N_DISCARD = [0, 3, 2, 1, 3, 2, 1, 0, 0, 0]     # Navadna_igra.py:48-57

# mix modes (tarok_env.h TAROK_MIX_*)
MIX_ALL = 0          # config 3: 1/3 Klop, 1/3 Berac (1/2 open), 1/3 Navadna/Solo over 7 types
MIX_NAVADNA3 = 1     # config 2: Tri/Dve/Ena uniform
MIX_BOT = 2          # contracts come out of a bidding round between four Bot players (Igra.py:75-114, Igralec.py:148-156)
MIX_FIXED = 16       # MIX_FIXED + code: every game plays that contract

# RNG draw indices
DRAW_CARD = 0        # 0..53  deal keys
DRAW_FAMILY = 64
DRAW_TYPE = 65
DRAW_DECLARER = 66
DRAW_KING = 67
DRAW_DISCARD = 68    # 68..70
DRAW_BID = 72        # 72 + n: the n-th licitiram call of the game's bidding round (MIX_BOT)
BID_ROUND_CAP = 8    # the re-bidding loop (Igra.py:98-113) is cut after 8 rounds: the holder plays
DRAW_POLICY = 128    # 128 + step


def mix64(x):
    x &= M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x


def game_key(seed, gidx, episode):
    a = (gidx * 0x9E3779B97F4A7C15 + episode * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & M64
    return mix64((seed & M64) ^ mix64(a))


def rng32(key, i):
    lo = key & M32
    hi = (key >> 32) & M32
    x = lo ^ ((i * 0x9E3779B1) & M32)
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & M32
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & M32
    x ^= x >> 16
    x ^= hi
    x = (x * 0x27D4EB2F) & M32
    x ^= x >> 15
    return x


def pick(r, n):
    """uniform index in [0, n) from a 32-bit draw (multiply-high)."""
    return (r * n) >> 32


def kth_bit(mask, k):
    """index of the k-th (0-based) set bit of mask."""
    for i in range(64):
        if (mask >> i) & 1:
            if k == 0:
                return i
            k -= 1
    raise ValueError("kth_bit out of range")


def popcount(m):
    return bin(m).count("1")


def deal(key):
    """perm[0:54]: hands are perm[12s:12s+12], talon perm[48:54] (Igra.py:65-73)."""
    keys = [((rng32(key, DRAW_CARD + c) & 0xFFFFFFC0) | c) for c in range(54)]
    keys.sort()
    return [k & 63 for k in keys]


def bot_wish(r):
    """Bot_igralec's wish (Igralec.py:151): Naprej 1/2, Tri / Dve / Ena 1/6 each, as int(Tip_igre)."""
    w = pick(r, 6)
    return -10 if w < 3 else (w - 2) * 10


def bot_bidding(key):
    """One bidding round between four Bots, straight-line restatement of Igra.licitacija
    (Igra.py:75-114) with the player-side filter (Igralec.py:58-74); the n-th licitiram
    call consumes draw DRAW_BID + n.  Returns (declarer seat, contract code)."""
    calls = [0]

    def ask(min_igra, obvezno, prednost):
        wish = bot_wish(rng32(key, DRAW_BID + calls[0]))
        calls[0] += 1
        ok = wish >= min_igra if prednost else wish > min_igra
        return wish if ok else (-10 if obvezno is None else obvezno)

    still, top = 0, 10
    for seat in (1, 2, 3):
        b = ask(top, None, False)
        if b != -10:
            still |= 1 << seat
        top = max(top, b)
    if top == 10:
        return 0, ask(-10, 0, False) // 10
    b = ask(top, None, True)
    if b != -10:
        still |= 1
    top = max(top, b)
    holder = min(i for i in range(4) if (still >> i) & 1)
    rounds = 0
    while popcount(still) != 1 and rounds < BID_ROUND_CAP:
        rounds += 1
        nxt = 0
        for seat in (1, 2, 3, 0):
            if (still >> seat) & 1:
                b = ask(top, top if seat == holder else None, False)
                if b != -10:
                    nxt |= 1 << seat
                    holder, top = seat, b
        still = nxt
    return holder, top // 10


def sample_setup(key, mix):
    """(contract, declarer, king_suit) for one synthetic game."""
    if mix == MIX_BOT:
        declarer, contract = bot_bidding(key)
        king = pick(rng32(key, DRAW_KING), 4) if contract in (TRI, DVE, ENA) else -1
        return contract, declarer, king
    if mix >= MIX_FIXED:
        contract = mix - MIX_FIXED
    elif mix == MIX_NAVADNA3:
        contract = TRI + pick(rng32(key, DRAW_TYPE), 3)
    else:
        fam = pick(rng32(key, DRAW_FAMILY), 3)
        r = rng32(key, DRAW_TYPE)
        if fam == 0:
            contract = KLOP
        elif fam == 1:
            contract = BERAC if pick(r, 2) == 0 else ODPRTI_BERAC
        else:
            contract = [TRI, DVE, ENA, SOLO_TRI, SOLO_DVE, SOLO_ENA, SOLO_BREZ][pick(r, 7)]
    declarer = 0 if contract == KLOP else pick(rng32(key, DRAW_DECLARER), 4)
    king = pick(rng32(key, DRAW_KING), 4) if contract in (TRI, DVE, ENA) else -1
    return contract, declarer, king


def bot_discards(key, hand_after_pickup, n):
    """Bot_igralec.menjaj_iz_talona (Igralec.py:161-171): random.sample of the
    discardable cards.  If fewer than n are discardable the reference raises
    (random.sample ValueError); the build falls back to the whole hand."""
    cand = hand_after_pickup & DISCARDABLE
    if popcount(cand) < n:
        cand = hand_after_pickup
    out = []
    for j in range(n):
        k = pick(rng32(key, DRAW_DISCARD + j), popcount(cand))
        c = kth_bit(cand, k)
        out.append(c)
        cand &= ~(1 << c)
    return out


def policy_action(key, step, mask):
    """Bot_igralec.igraj_karto (Igralec.py:158-159): uniform over the legal set."""
    return kth_bit(mask, pick(rng32(key, DRAW_POLICY + step), popcount(mask)))
