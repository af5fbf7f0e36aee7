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
