"""CPU-side check of the DEVICE rule code itself: tarok_amd/csrc/tarok_device.h is compiled by g++
with the gfx950 builtins it uses emulated (tests/host_emu/), random games are played through its
deal / setup / exchange / legal-mask / policy / apply-step / scoring functions, and every seat,
legal mask, card, score and game length is compared with the CPU oracle.  No GPU involved: this
pins the rule arithmetic (bit-plane selects, bitop3 truth tables, k-th-bit select, prestej) before
the kernels ever run."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "host_emu")


@pytest.fixture(scope="module")
def host_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "device_rules_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-I", EMU, "-o", out, os.path.join(EMU, "device_rules_host.cpp")])
    return out


@pytest.mark.parametrize("mix,seed,offset,episode", [(0, 5, 0, 0), (0, 123456789, 987654321, 3), (1, 7, 11, 1), (2, 9, 0, 0),
                                                      (16, 1, 0, 0), (23, 1, 0, 2), (25, 2, 5, 0), (24, 3, 0, 0)])
def test_device_rules_on_the_host_equal_the_oracle(host_binary, tmp_path, mix, seed, offset, episode):
    from oracle import oracle as O
    n = 6000
    path = str(tmp_path / "out.bin")
    subprocess.check_call([host_binary, str(seed), str(offset), str(n), str(episode), str(mix), path])
    rec = np.dtype([("seats", np.int8, 48), ("masks", np.uint64, 48), ("actions", np.uint8, 48), ("scores", np.int16, 4),
                    ("nsteps", np.int16)])
    got = np.fromfile(path, dtype=rec)
    assert got.shape == (n,)
    ref = O.rollout(seed, offset, n, episode, mix)
    assert (got["nsteps"] == ref["nsteps"]).all()
    assert (got["scores"] == ref["scores"]).all()
    live = np.arange(48)[None, :] < ref["nsteps"][:, None]
    assert (got["masks"][live] == ref["masks"][live]).all()
    assert (got["seats"][live] == ref["seats"][live]).all()
    assert (got["actions"][live] == ref["actions"][live]).all()
    assert (got["masks"][~live] == 0).all()


def test_device_rules_on_the_host_replay_the_reference_traces(host_binary, tmp_path):
    """The 2,880 games recorded from the reference engine (tests/golden/traces_v1.npz: all ten
    contracts, recorded exchanges and cards) replayed card by card through the device rule code on
    the CPU: every seat, legal mask (`mozne`), per-trick winner and value, the game length and the
    final scores equal what the reference produced."""
    from oracle import tarok_spec as S
    tr = np.load(os.path.join(ROOT, "tests", "golden", "traces_v1.npz"))
    n = len(tr["contract"])
    rec = np.dtype([("deal", np.uint8, 54), ("contract", np.int8), ("declarer", np.int8), ("king", np.int8), ("choice", np.int8),
                    ("discards", np.uint8, 3), ("nsteps", np.uint8), ("actions", np.uint8, 48)])
    assert rec.itemsize == 110
    a = np.zeros(n, rec)
    a["deal"] = tr["deals"]; a["contract"] = tr["contract"]; a["declarer"] = tr["declarer"]; a["king"] = tr["king"]
    a["choice"] = np.where(tr["choice"] < 0, 0, tr["choice"])
    disc = np.array(tr["discards"]).astype(np.int64)
    ndisc = np.array([S.N_DISCARD[int(c)] for c in tr["contract"]])
    disc = np.where(np.arange(3)[None, :] < ndisc[:, None], disc, 255)          # unused slots: no card
    a["discards"] = disc.astype(np.uint8)
    a["nsteps"] = tr["nsteps"]
    acts = np.array(tr["actions"]).astype(np.int64)
    a["actions"] = np.where(acts < 0, 255, acts).astype(np.uint8)
    pin, pout = str(tmp_path / "traces.bin"), str(tmp_path / "replayed.bin")
    a.tofile(pin)
    subprocess.check_call([host_binary, "replay", pin, pout])
    out = np.dtype([("ok", np.int8), ("done", np.int8), ("seats", np.int8, 48), ("masks", np.uint64, 48), ("tinfo", np.uint16, 48),
                    ("rc", np.int8, 48), ("scores", np.int16, 4)])
    got = np.fromfile(pout, dtype=out)
    assert got.shape == (n,)
    assert (got["ok"] == 1).all() and (got["done"] == 1).all()
    nst = tr["nsteps"].astype(np.int64)
    live = np.arange(48)[None, :] < nst[:, None]
    assert (got["seats"][live] == tr["seats"][live]).all()
    assert (got["masks"][live] == tr["masks"][live].astype(np.uint64)).all()
    last = np.arange(48)[None, :] == (nst[:, None] - 1)
    assert (got["rc"][live & ~last] == 0).all() and (got["rc"][last] == 1).all()
    assert (got["scores"] == tr["scores"]).all()
    # what rezultat_stiha was told after every 4th card: 0x8000 | vrednost_stiha << 4 | winner seat
    for k in range(12):
        t = 4 * k + 3
        sel = nst > t
        exp = 0x8000 | (tr["trick_value"][sel, k].astype(np.int64) << 4) | tr["trick_winner"][sel, k].astype(np.int64)
        assert (got["tinfo"][sel, t].astype(np.int64) == exp).all(), k


def test_device_rules_under_address_and_ub_sanitizers(tmp_path):
    """The same host build with -fsanitize=address,undefined (GPU sanitizers are not available on the
    pool: the CPU build is where the rule code's shifts, indexing and integer arithmetic are checked)."""
    exe = str(tmp_path / "device_rules_host_san")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", EMU, "-o", exe, os.path.join(EMU, "device_rules_host.cpp")])
    for mix, seed in ((0, 5), (2, 9), (25, 1)):
        subprocess.check_call([exe, str(seed), "0", "1500", "0", str(mix), str(tmp_path / "o.bin")])
