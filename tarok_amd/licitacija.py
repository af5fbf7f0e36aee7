"""Host-side bidding round — the control flow of `Igra.licitacija` (Igra.py:75-114).

One call per game and policy-driven, so it stays on the host; the
GPU env takes its outcome (declarer seat, contract) as `tarok_reset` inputs.

Bids are the reference's `int(Tip_igre)` values (Tip_igre.py:4-15) so that the `>` / `>=`
comparisons of the player-side filter (Igralec.py:58-74) mean the same thing.
"""

NAPREJ, KLOP, TRI, DVE, ENA = -10, 0, 10, 20, 30
SOLO_TRI, SOLO_DVE, SOLO_ENA, BERAC, SOLO_BREZ, ODPRTI_BERAC = 40, 50, 60, 70, 80, 90
ALL_BIDS = (NAPREJ, KLOP, TRI, DVE, ENA, SOLO_TRI, SOLO_DVE, SOLO_ENA, BERAC, SOLO_BREZ, ODPRTI_BERAC)


def base_filter(zelim, min_igra, obvezno=None, prednost=False):
    """What `Igralec.licitiram` (Igralec.py:58-74) lets through: the wish if it beats
    (`prednost`: at least matches) the standing bid, else the forced bid, else Naprej."""
    ok = zelim >= min_igra if prednost else zelim > min_igra
    if ok:
        return zelim
    return NAPREJ if obvezno is None else obvezno


def licitacija(licitiram, max_rounds=64):
    """Run one bidding round.

    `licitiram(seat, min_igra, obvezno, prednost) -> int` is asked exactly as the
    reference asks `igralci[seat].licitiram(min_igra, id, obvezno, prednost)`, in the same
    order.  Returns `(declarer_seat, contract_value)`.

    Order of play (Igra.py:81-114): seats 1,2,3 bid over a floor of Tri; if nobody did,
    seat 0 must play at least Klop; otherwise seat 0 may match the top bid (priority), and
    the bidders keep going — seat 0 asked last, the current holder allowed to stand on its
    bid — until one is left.
    """
    still_in = set()
    top = TRI
    for seat in (1, 2, 3):
        b = licitiram(seat, top, None, False)
        if b != NAPREJ:
            still_in.add(seat)
        top = max(top, b)
    if top == TRI:
        return 0, licitiram(0, NAPREJ, KLOP, False)
    b = licitiram(0, top, None, True)
    if b != NAPREJ:
        still_in.add(0)
    top = max(top, b)
    holder = min(still_in)
    rounds = 0
    while len(still_in) != 1:
        if not still_in or rounds >= max_rounds:
            # the reference spins forever here; a player that drops a bid it holds breaks the protocol
            raise RuntimeError("licitacija: no bidder left / bidding does not terminate")
        rounds += 1
        order = sorted(still_in)
        if order[0] == 0:
            order = order[1:] + [0]
        nxt = set()
        for seat in order:
            b = licitiram(seat, top, top if seat == holder else None, False)
            if b != NAPREJ:
                nxt.add(seat)
                holder = seat
                top = b
        still_in = nxt
    return holder, top
