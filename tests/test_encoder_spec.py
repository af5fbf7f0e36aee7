"""The restatement of the reference agent's encoders (oracle/encoder_spec.py) on hand-worked cases.
Parity unpinned (see that file's header): these cases were worked out from the text of
Igralec.py:453-543, not produced by running it."""
import numpy as np

from oracle import encoder_spec as E


def test_history_length_counts_every_entry_and_always_pads():
    assert E.history_length([]) == 8                                   # 0 -> 8 (Igralec.py:460)
    assert E.history_length([(0, 1)] * 7) == 8
    assert E.history_length([(0, 1)] * 8) == 16                        # already a multiple: a full 8 more
    assert E.history_length([(E.TALON, (0, [[1], [2]]))] + [(0, 1)] * 7) == 16     # the Talon entry counts
    assert E.history_length([(0, 1)] * 4 + [(None, 5)] + [(1, 2)] * 3) == 16       # Klop's talon card counts
    assert E.history_length([(0, 1)] * 47 + [(None, 5)] * 6) == 56                 # the longest history


def test_igralci2index_skips_the_player_himself():
    assert E.igralci2index(0) == {1: 0, 2: 1, 3: 2, 0: 3}
    assert E.igralci2index(2) == {0: 0, 1: 1, 3: 2, 2: 3}


def test_klop_rows_follow_plays_not_entries():
    # trick 1: seats 0,1,2,3 play 10,11,12,13; talon card 40 joins; then seat 1 leads 20; seat 2 (me) to move
    z = [(0, 10), (1, 11), (2, 12), (3, 13), (None, 40), (1, 20)]
    r = E.stanje_v_vektor(2, "Klop", z, zacetna_roka=[12, 5, 6], zalozil=None, barva_kralja=None, declarer=0, mozne=[5, 6])
    opp, own, talon, legal = r
    assert opp.shape == (1, 8, 3, 54) and own.shape == (1, 8, 54) and talon.shape == (1, 54) and legal.shape == (1, 54)
    assert opp[0, 0, 0, 10] == 1 and opp[0, 1, 1, 11] == 1 and opp[0, 3, 2, 13] == 1      # seat 3 -> channel 2
    assert opp[0, 4, 1, 20] == 1                                       # the 5th PLAY is row 4: the talon entry took no row
    assert opp.sum() == 4
    assert own[0, 2].tolist() == [1 if c in (12, 5, 6) else 0 for c in range(54)]          # hand before the own play
    assert own.sum() == 3
    assert talon[0, 40] == 1 and talon.sum() == 1
    assert legal[0, 5] == 1 and legal[0, 6] == 1 and legal.sum() == 2


def test_navadna_talon_rows_discards_and_dealt_hand():
    groups = [[30, 31, 32], [33, 34, 35]]
    z = [(E.TALON, (1, groups)), (0, 1), (1, 33)]                      # declarer (seat 1) plays a card he picked up
    r = E.stanje_v_vektor(1, "Navadna_igra", z[:2], zacetna_roka=[2, 3, 4], zalozil=[3, 34], barva_kralja=2, declarer=1,
                          mozne=[2])
    opp, king, own, talon, index, disc, legal = r
    assert opp.shape == (1, 8, 3, 54) and talon.shape == (1, 6, 55)
    assert king[0].tolist() == [0, 0, 1, 0] and index[0].tolist() == [0, 0, 0, 1]         # the declarer himself: 3
    assert [int(talon[0, r_].argmax()) for r_ in range(6)] == [30, 31, 32, 33, 34, 35]
    assert talon[0, :, 54].tolist() == [0, 0, 0, 1, 1, 1]              # the chosen group's rows
    assert disc[0, 3] == 1 and disc[0, 34] == 1 and disc.sum() == 2
    # next observation of the same player: his own play sits in row 1 with the hand AS DEALT — the discarded 3
    # still in it, the picked-up cards not (zacetna_roka, Igralec.py:264,465)
    r2 = E.stanje_v_vektor(1, "Navadna_igra", z + [(2, 9)], [2, 3, 4], [3, 34], 2, 1, [2])
    own2 = r2[2]
    assert own2[0, 1].tolist() == [1 if c in (2, 3, 4) else 0 for c in range(54)] and own2.sum() == 3
    # an opponent's view: declarer seat 1 seen from seat 3 is index 1; no discards
    r3 = E.stanje_v_vektor(3, "Navadna_igra", z, [7], None, 2, 1, [7])
    assert r3[4][0].tolist() == [0, 1, 0, 0] and r3[5].sum() == 0 and r3[0][0, 1, 1, 33] == 1


def test_berac_and_solo_lists():
    r = E.stanje_v_vektor(0, "Berac", [], [1], None, None, 2, [1])
    assert [x.shape for x in r] == [(1, 8, 3, 54), (1, 8, 54), (1, 4), (1, 54)] and r[2][0].tolist() == [0, 1, 0, 0]
    r = E.stanje_v_vektor(0, "Solo", [], [1], None, None, 0, [1])
    assert [x.shape for x in r] == [(1, 8, 3, 54), (1, 8, 54), (1, 6, 55), (1, 4), (1, 54), (1, 54)]


def test_exchange_vector_and_final_reward():
    roka, talon, igra = E.menjaj_talon_v_vektor([0, 53], [[10, 11], [12, 13], [14, 15]], contract=2, barva_kralja=3)
    assert roka.sum() == 2 and talon[0, 12, 1] == 1 and talon[0, 15, 2] == 1 and talon.sum() == 6
    assert igra[0].argmax() == 7                                       # (Dve, KRIZ): 4 + 3
    assert E.menjaj_talon_v_vektor([], [[1]] * 6, contract=6, barva_kralja=None)[2][0].argmax() == 14   # Solo_ena
    assert E.rezultat_igre_st_tock(0, "Berac", False, 0) == -20 and E.rezultat_igre_st_tock(0, "Berac", False, 7) == 20
    assert E.rezultat_igre_st_tock(-70, "Berac", True, 7) == -70 and E.rezultat_igre_st_tock(35, "Solo", False, 0) == 35


def test_pack_record_layout():
    z = [(0, 10), (1, 11)]
    lists = E.stanje_v_vektor(2, "Klop", z, [12], None, None, 0, [12])
    rec, T = E.pack_record(2, "Klop", lists)
    assert T == 8 and rec.shape == (12544,)
    assert rec[0 * 162 + 0 * 54 + 10] == 1 and rec[1 * 162 + 1 * 54 + 11] == 1 and rec[:9072].sum() == 2
    assert rec[12488 + 12] == 1 and rec[12488:12542].sum() == 1 and rec[9072:12488].sum() == 0
