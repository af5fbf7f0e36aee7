"""The build's `main`-equivalent: the shape of the reference's self-play driver
(main.py:95-156 — shuffle the seats, `Tarok(igralci, st_iger).paralel_start()`, collect the
scores, repeat) with the card play on the GPU.  The reference's own main.py cannot even be
imported here (it needs pytorch_lightning and a torch_models module that is not in the
reference), and its learning agents are out of scope; any reference-shaped player works.

    python -m tarok_amd.main --games 2000 --iterations 3
"""
import argparse
import random
import time
import warnings

from .igralec import Bot_igralec, Tarok


def main(st_iger=2000, iterations=1, seed=0, device=0, igralci=None, verbose=True):
    """Returns the list of per-iteration score dicts {player name: total}."""
    rnd = random.Random(seed)
    igralci = list(igralci) if igralci is not None else [Bot_igralec(i, rng=random.Random(1000 * seed + i)) for i in range(4)]
    scores = []
    for it in range(iterations):
        rnd.shuffle(igralci)                                     # main.py:112
        t0 = time.time()
        igra = Tarok(igralci, st_iger, seed=seed + it, device=device)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                      # Igralec.poglej_karte_odprtega_beraca warns by design
            rez = igra.paralel_start()                           # main.py:114-115
        scores.append({p.ime: int(v) for p, v in rez.items()})
        if verbose:
            print("Time need for %d game %.2fs" % (st_iger, time.time() - t0), scores[-1])   # main.py:116
    return scores


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=2000)
    ap.add_argument("--iterations", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    main(a.games, a.iterations, a.seed, a.device)
