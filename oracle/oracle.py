"""TEST INFRASTRUCTURE — ctypes binding of oracle/libtarok_oracle.so (the CPU
restatement of the reference rules).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product never does."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtarok_oracle.so")


class ToGame(C.Structure):
    _fields_ = [
        ("hand", C.c_uint64 * 4), ("pile", C.c_uint64 * 4),
        ("talon", C.c_uint8 * 6), ("trick", C.c_uint8 * 4),
        ("n_in_trick", C.c_uint8), ("leader", C.c_uint8), ("trick_no", C.c_uint8),
        ("contract", C.c_uint8), ("declarer", C.c_uint8), ("king", C.c_int8),
        ("team", C.c_uint8), ("talon_left", C.c_uint8), ("choice", C.c_int8),
        ("phase", C.c_uint8), ("error", C.c_uint8), ("score", C.c_int16 * 4), ("last_trick", C.c_uint16),
    ]


def build(force=False):
    src = os.path.join(HERE, "tarok_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    u64, i64, i32, u32 = C.c_uint64, C.c_int64, C.c_int, C.c_uint32
    P = C.POINTER
    L.to_prestej.restype = i32; L.to_prestej.argtypes = [u64]
    L.to_vrednost.restype = i32; L.to_vrednost.argtypes = [i32]
    L.to_discardable.restype = u64; L.to_discardable.argtypes = [u64]
    L.to_legal_navadna.restype = u64; L.to_legal_navadna.argtypes = [u64, i32]
    L.to_legal_klop.restype = u64; L.to_legal_klop.argtypes = [u64, i32]
    L.to_vrednost_stiha.restype = i32; L.to_vrednost_stiha.argtypes = [u64, i32]
    L.to_trick_winner.restype = i32; L.to_trick_winner.argtypes = [P(C.c_uint8)]
    L.to_new_game.restype = None; L.to_new_game.argtypes = [P(ToGame), P(C.c_uint8), i32, i32, i32]
    L.to_exchange.restype = i32; L.to_exchange.argtypes = [P(ToGame), i32, P(C.c_uint8)]
    L.to_seat.restype = i32; L.to_seat.argtypes = [P(ToGame)]
    L.to_legal.restype = u64; L.to_legal.argtypes = [P(ToGame)]
    L.to_step.restype = i32; L.to_step.argtypes = [P(ToGame), i32]
    L.to_export_lanes.restype = None; L.to_export_lanes.argtypes = [P(ToGame), P(u64)]
    L.to_game_key.restype = u64; L.to_game_key.argtypes = [u64, u64, u64]
    L.to_rng32.restype = u32; L.to_rng32.argtypes = [u64, u32]
    L.to_deal_perm.restype = None; L.to_deal_perm.argtypes = [u64, P(C.c_uint8)]
    L.to_sample_setup.restype = None; L.to_sample_setup.argtypes = [u64, i32, P(i32), P(i32), P(i32)]
    L.to_bot_bidding.restype = None; L.to_bot_bidding.argtypes = [u64, P(i32), P(i32)]
    L.to_bot_discards.restype = None; L.to_bot_discards.argtypes = [u64, u64, i32, P(C.c_uint8)]
    L.to_policy_action.restype = i32; L.to_policy_action.argtypes = [u64, i32, u64]
    L.to_synth_game.restype = None; L.to_synth_game.argtypes = [P(ToGame), u64, u64, u64, i32]
    vp = C.c_void_p
    L.to_rollout.restype = i64; L.to_rollout.argtypes = [u64, u64, i64, u64, i32, vp, vp, vp, vp, vp]
    L.to_rollout_mt.restype = i64; L.to_rollout_mt.argtypes = [i32, u64, u64, i64, u64, i32, vp, vp, vp, vp, vp]
    L.to_obs_word.restype = u64; L.to_obs_word.argtypes = [P(ToGame), i32]
    L.to_run_autoreset.restype = i64; L.to_run_autoreset.argtypes = [u64, u64, i64, i32, u32, i64, vp, vp, vp, vp]
    _lib = L
    return L


def u8arr(seq):
    return (C.c_uint8 * len(seq))(*[int(x) for x in seq])


class Game:
    """One game on the CPU oracle."""

    def __init__(self, perm=None, contract=0, declarer=0, king=-1):
        self.g = ToGame()
        if perm is not None:
            lib().to_new_game(C.byref(self.g), u8arr(perm), int(contract), int(declarer), int(king))

    @classmethod
    def synth(cls, seed, gidx, episode, mix):
        self = cls()
        lib().to_synth_game(C.byref(self.g), seed, gidx, episode, mix)
        return self

    @classmethod
    def from_lanes(cls, lanes):
        """A game rebuilt from the canonical lanes (include/tarok_env.h TAROK_LANE_*): the inverse of lanes()."""
        self = cls()
        g = self.g
        for s in range(4):
            g.hand[s], g.pile[s] = int(lanes[s]), int(lanes[4 + s])
        tal, m = int(lanes[8]), int(lanes[9])
        for i in range(6):
            g.talon[i] = (tal >> (6 * i)) & 63
        g.n_in_trick = (m >> 24) & 7
        for i in range(4):
            g.trick[i] = (m >> (6 * i)) & 63 if i < g.n_in_trick else 0
        g.leader, g.trick_no, g.contract, g.declarer = (m >> 27) & 3, (m >> 29) & 15, (m >> 33) & 15, (m >> 37) & 3
        king, choice = (m >> 39) & 7, (m >> 49) & 7
        g.king = -1 if king == 7 else king
        g.team, g.talon_left = (m >> 42) & 15, (m >> 46) & 7
        g.choice = -1 if choice == 7 else choice
        g.phase, g.error = (m >> 52) & 3, (m >> 54) & 1
        return self

    def obs_word(self, finished_now=False):
        return int(lib().to_obs_word(C.byref(self.g), 1 if finished_now else 0))

    def exchange(self, choice, discards):
        d = list(discards) + [255] * (3 - len(discards))
        return lib().to_exchange(C.byref(self.g), int(choice), u8arr(d))

    def seat(self):
        return lib().to_seat(C.byref(self.g))

    def legal(self):
        return int(lib().to_legal(C.byref(self.g)))

    def step(self, action):
        return lib().to_step(C.byref(self.g), int(action))

    def lanes(self):
        out = (C.c_uint64 * 10)()
        lib().to_export_lanes(C.byref(self.g), out)
        return np.array(list(out), dtype=np.uint64)

    @property
    def done(self):
        return self.g.phase == 3

    @property
    def scores(self):
        return [int(x) for x in self.g.score]


def rollout(seed, gidx0, n, episode, mix, threads=1, trace=True):
    """Random-policy rollouts of synthetic games; returns dict of arrays."""
    L = lib()
    out = dict(nsteps=np.zeros(n, np.int16), scores=np.zeros((n, 4), np.int16))
    if trace:
        out.update(seats=np.zeros((n, 48), np.int8), masks=np.zeros((n, 48), np.uint64),
                   actions=np.zeros((n, 48), np.uint8))

    def ptr(name):
        return out[name].ctypes.data if name in out else None
    total = L.to_rollout_mt(threads, seed, gidx0, n, episode, mix, ptr("nsteps"), ptr("seats"),
                            ptr("masks"), ptr("actions"), ptr("scores"))
    out["total_steps"] = int(total)
    return out


def run_autoreset(seed, gidx0, n, mix, n_steps, episode0=0, threads=1):
    """CPU statement of tarok_run_random(..., TAROK_AUTO_RESET).  threads > 1: the slots are split into
    contiguous ranges run in parallel (slots are independent; the C call releases the GIL)."""
    L = lib()

    def part(lo, hi):
        m = hi - lo
        o = dict(episode=np.zeros(m, np.uint32), score_sum=np.zeros((m, 4), np.int32),
                 lanes=np.zeros((10, m), np.uint64), obs=np.zeros(m, np.uint64))
        o["total_steps"] = int(L.to_run_autoreset(seed, gidx0 + lo, m, mix, episode0, n_steps, o["episode"].ctypes.data,
                                                  o["score_sum"].ctypes.data, o["lanes"].ctypes.data, o["obs"].ctypes.data))
        return o
    threads = max(1, min(int(threads), n))
    if threads == 1:
        return part(0, n)
    from concurrent.futures import ThreadPoolExecutor
    cuts = [n * k // threads for k in range(threads + 1)]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda k: part(cuts[k], cuts[k + 1]), range(threads)))
    return dict(episode=np.concatenate([q["episode"] for q in parts]), score_sum=np.concatenate([q["score_sum"] for q in parts]),
                lanes=np.concatenate([q["lanes"] for q in parts], axis=1), obs=np.concatenate([q["obs"] for q in parts]),
                total_steps=sum(q["total_steps"] for q in parts))


def policy_action(key, step, legal):
    """The Bot policy's card (oracle/tarok_spec.py: uniform among the legal cards on draw 128 + step)."""
    return int(lib().to_policy_action(int(key), int(step), int(legal)))


def game_key(seed, gidx, episode):
    return int(lib().to_game_key(int(seed), int(gidx), int(episode)))
