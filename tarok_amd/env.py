"""TarokVecEnv — N lock-stepped 4-player Tarok games on one MI355X.

Host-side mirror of what the reference does with N `Igra` objects inside
`Tarok.paralel_start` (Tarok.py:30-62): `reset()` deals and sets the contracts
up (Igra.razdeli + engine constructors + talon exchange), `legal_actions()` is
`mozne_karte` for the seat to move, `step()` is one `next(g)` per game.  All
compute is in libtarokenv.so (HIP, gfx950); torch only provides device memory
and streams.  There is no CPU path: constructing an env without the library or
without a GPU raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _native
from . import karte as K


def _u64(t):
    """int64 torch tensor -> numpy uint64 (host)."""
    return t.detach().cpu().numpy().view(np.uint64)


class Obs:
    """Per-game observation words (include/tarok_env.h TAROK_OBS_*) as an int64
    device tensor with decoded views."""

    def __init__(self, words):
        self.words = words

    @property
    def mask(self):            # legal-card mask of the seat to move (= `mozne`)
        return self.words & K.OBS_MASK

    @property
    def seat(self):
        return (self.words >> K.OBS_SEAT_SHIFT) & 3

    @property
    def step(self):            # cards played so far in the game
        return (self.words >> K.OBS_STEP_SHIFT) & 63

    @property
    def done(self):
        return ((self.words >> K.OBS_DONE_BIT) & 1).bool()

    @property
    def error(self):
        return self.words < 0   # bit 63

    def mask_numpy(self):
        return _u64(self.mask)


class TarokVecEnv:
    def __init__(self, n_games, device=0, seed=0, mix=K.MIX_ALL, game_offset=0, history=False, refill_fan=None, lazy_refill=None):
        """refill_fan, lazy_refill: launch tuning (tarok_set_option; None = the library's default for the batch size);
        results never depend on them."""
        self._h = None
        L = _native.lib()
        if not torch.cuda.is_available() or L.tarok_device_count() == 0:
            raise _native.TarokNativeError("TarokVecEnv needs an MI355X: no GPU is visible and there is no CPU fallback")
        self.L = L
        self.n = int(n_games)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.seed, self.mix, self.game_offset = int(seed), int(mix), int(game_offset)
        h = C.c_void_p()
        self.history = bool(history)
        _native.check(L.tarok_create(C.byref(h), self.device_index, self.n, self.game_offset, self.seed, self.mix,
                                     K.HISTORY if history else 0))
        self._h = h
        import os
        if refill_fan is None and os.environ.get("TAROK_REFILL_FAN"):      # diagnostics (tools/: A/B runs of whole scripts)
            refill_fan = int(os.environ["TAROK_REFILL_FAN"])
        if refill_fan is not None:
            _native.check(L.tarok_set_option(h, K.OPT_REFILL_FAN, int(refill_fan)))
        if lazy_refill is None and os.environ.get("TAROK_LAZY_REFILL"):
            lazy_refill = int(os.environ["TAROK_LAZY_REFILL"])
        if lazy_refill is not None:
            _native.check(L.tarok_set_option(h, K.OPT_LAZY_REFILL, int(lazy_refill)))
        with torch.cuda.device(self.device):
            self.obs_words = torch.zeros(self.n, dtype=torch.int64, device=self.device)
            self.reward = torch.zeros((self.n, 4), dtype=torch.int16, device=self.device)
            self.done = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
            self.action = torch.full((self.n,), 255, dtype=torch.uint8, device=self.device)
            self.trick = None     # allocated by step(..., tricks=True)

    # ------------------------------------------------------------------
    def set_option(self, refill_fan=None, lazy_refill=None):
        """Launch tuning (tarok_set_option) at any point of a run; results never depend on it.  A change of the refill fan
        after the first step launch synchronises the device (include/tarok_env.h)."""
        if refill_fan is not None:
            _native.check(self.L.tarok_set_option(self._h, K.OPT_REFILL_FAN, int(refill_fan)))
        if lazy_refill is not None:
            _native.check(self.L.tarok_set_option(self._h, K.OPT_LAZY_REFILL, int(lazy_refill)))

    def refill_selftest(self, kind, per_slot, episode0=100, order=0, reps=1):
        """tarok_debug_refill_selftest (tests only; reset() afterwards): (wrong lines, [first records])."""
        import numpy as np
        rep = np.zeros(49, np.uint64)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _native.check(self.L.tarok_debug_refill_selftest(self._h, int(kind), int(per_slot), int(episode0), int(order), int(reps),
                                                             rep.ctypes.data))
        return int(rep[0]), rep[1:].reshape(8, 6)[: min(8, int(rep[0]))]

    def close(self):
        if self._h is not None:
            torch.cuda.synchronize(self.device)
            self.L.tarok_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, x, dtype, shape):
        """None -> NULL; tensor / array -> contiguous device tensor of dtype."""
        if x is None:
            return None
        t = torch.as_tensor(x)
        t = t.to(device=self.device, dtype=dtype).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    # ------------------------------------------------------------------
    def reset(self, episode=0, deals=None, contract=None, declarer=None, king_suit=None,
              talon_choice=None, discards=None, defer_exchange=False, clear_counters=True):
        """Deal + set up all N games (Igra.py:38-55,65-73; Navadna_igra.py:20-68).
        Returns the first observation."""
        n = self.n
        deals = self._dev(deals, torch.uint8, (n, 54))
        contract = self._dev(contract, torch.int8, (n,))
        declarer = self._dev(declarer, torch.int8, (n,))
        king_suit = self._dev(king_suit, torch.int8, (n,))
        talon_choice = self._dev(talon_choice, torch.int8, (n,))
        discards = self._dev(discards, torch.uint8, (n, 3))
        flags = (K.DEFER_EXCHANGE if defer_exchange else 0) | (K.CLEAR_COUNTERS if clear_counters else 0)
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_reset(self._h, int(episode), self._p(deals), self._p(contract), self._p(declarer),
                                             self._p(king_suit), self._p(talon_choice), self._p(discards), flags,
                                             self._stream()))
        return self.legal_actions()

    def exchange(self, talon_choice=None, discards=None):
        """menjaj_iz_talona for games still waiting (Navadna_igra.py:60-66)."""
        talon_choice = self._dev(talon_choice, torch.int8, (self.n,))
        discards = self._dev(discards, torch.uint8, (self.n, 3))
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_exchange(self._h, self._p(talon_choice), self._p(discards), self._stream()))
        return self.legal_actions()

    def legal_actions(self):
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_legal_actions(self._h, self._p(self.obs_words), None, self._stream()))
        return Obs(self.obs_words)

    def _trick_buf(self, tricks):
        if tricks and self.trick is None:
            with torch.cuda.device(self.device):
                self.trick = torch.zeros(self.n, dtype=torch.int16, device=self.device)
        return self.trick if tricks else None

    def step(self, action, auto_reset=False, tricks=False, obs_out=None, reward_out=None, done_out=None, reward_ref=False):
        """One card per game.  Returns (Obs, reward[N,4] i16 — valid where done, done[N] u8).
        reward_ref=True: reward carries what rezultat_igre folds into the last transition (Igralec.py:421-437:
        the scores, but -20 / +20 for the defenders of a Berac) instead of the plain scores.
        tricks=True also fills self.trick [N] i16: 0, or 0x8000 | vrednost_stiha<<4 | winner seat
        for games whose trick this card completed (what rezultat_stiha is told).
        obs_out / reward_out / done_out: caller-owned device tensors to write into instead of
        the env's own buffers (e.g. rows of a rollout buffer)."""
        a = self._dev(action, torch.uint8, (self.n,))
        ob = self.obs_words if obs_out is None else obs_out
        rw = self.reward if reward_out is None else reward_out
        dn = self.done if done_out is None else done_out
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_step(self._h, self._p(a), self._p(rw), self._p(dn),
                                            self._p(self._trick_buf(tricks)), self._p(ob),
                                            (K.AUTO_RESET if auto_reset else 0) | (K.REWARD_REF if reward_ref else 0),
                                            self._stream()))
        return Obs(ob), rw, dn

    def sample_policy(self, logits, obs_words, action_out=None, logp_out=None):
        """Masked categorical sample from bf16 logits [N,64] (tarok_sample_policy)."""
        assert logits.dtype == torch.bfloat16 and logits.is_contiguous() and tuple(logits.shape) == (self.n, 64)
        with torch.cuda.device(self.device):
            if action_out is None:
                action_out = torch.empty(self.n, dtype=torch.uint8, device=self.device)
            if logp_out is None:
                logp_out = torch.empty(self.n, dtype=torch.float32, device=self.device)
            _native.check(self.L.tarok_sample_policy(self._h, self._p(logits), self._p(obs_words), self._p(action_out),
                                                     self._p(logp_out), self._stream()))
        return action_out, logp_out

    def policy_random(self, obs=None):
        """Bot_igralec.igraj_karto on device (Igralec.py:158-159)."""
        words = self.obs_words if obs is None else obs.words
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_policy_random(self._h, self._p(words), self._p(self.action), self._stream()))
        return self.action

    def step_random(self, auto_reset=False, tricks=False, reward_ref=False):
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_step_random(self._h, self._p(self.action), self._p(self.reward), self._p(self.done),
                                                   self._p(self._trick_buf(tricks)), self._p(self.obs_words),
                                                   (K.AUTO_RESET if auto_reset else 0) | (K.REWARD_REF if reward_ref else 0),
                                                   self._stream()))
        return Obs(self.obs_words), self.reward, self.done

    def prefetch(self):
        """Fill every next-game line that reset() emptied (reset() calls it itself; afterwards the step
        launches keep the slots' seven dealt-ahead games full on their own: callers never need this)."""
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_prefetch(self._h, self._stream()))

    def _krog_bufs(self, cards):
        if getattr(self, "_kb_cards", 0) != cards:
            with torch.cuda.device(self.device):
                self._kb = dict(action=torch.full((cards, self.n), 255, dtype=torch.uint8, device=self.device),
                                reward=torch.zeros((cards, self.n, 4), dtype=torch.int16, device=self.device),
                                done=torch.zeros((cards, self.n), dtype=torch.uint8, device=self.device),
                                trick=torch.zeros((cards, self.n), dtype=torch.int16, device=self.device),
                                obs=torch.zeros((cards, self.n), dtype=torch.int64, device=self.device))
            self._kb_cards = cards
        return self._kb

    def krog_random(self, cards=4, auto_reset=False, reward_ref=False, tricks=True):
        """`cards` cards of every game in one launch, Bot policy in-kernel (cards=4: one trick =
        one pass of the reference's krog).  Returns dict of [cards,N] tensors: action, reward
        [cards,N,4] (valid where done), done, trick, obs (observation words); also updates
        self.obs_words to the last row.  tricks=False: no per-trick rows (trick_out = NULL: the
        set of outputs tarok_run_random asks for, i.e. the card loop the bench times)."""
        kb = self._krog_bufs(cards)
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_krog_random(self._h, int(cards), self.n, self._p(kb["action"]), self._p(kb["reward"]),
                                                   self._p(kb["done"]), self._p(kb["trick"]) if tricks else None, self._p(kb["obs"]),
                                                   (K.AUTO_RESET if auto_reset else 0) | (K.REWARD_REF if reward_ref else 0),
                                                   self._stream()))
            self.obs_words.copy_(kb["obs"][cards - 1])
        return kb

    def run_random(self, n_steps, fused=False, graph_chunk=0, auto_reset=True, prefetch_every=0, cards_per_launch=None, done_rows=True):
        """n_steps lock-steps of the random policy launched from C (optionally graph-replayed).
        cards_per_launch: 0 = policy + step kernels, 1 = fused one-card kernel, >= 2 = that many
        cards per launch (tarok_krog_random); default from `fused`.  done_rows=False (one-card modes): done_out = NULL —
        a consumer reads "finished by this step" off bit 62 of the observation word instead of a byte row."""
        cards = (1 if fused else 0) if cards_per_launch is None else int(cards_per_launch)
        with torch.cuda.device(self.device):
            if cards >= 2:
                kb = self._krog_bufs(cards)
                _native.check(self.L.tarok_run_random(self._h, int(n_steps), cards, int(graph_chunk), int(prefetch_every),
                                                      self._p(kb["action"]), self._p(kb["reward"]), self._p(kb["done"]),
                                                      self._p(kb["obs"]), K.AUTO_RESET if auto_reset else 0, self._stream()))
                self.obs_words.copy_(kb["obs"][cards - 1])
            else:
                _native.check(self.L.tarok_run_random(self._h, int(n_steps), cards, int(graph_chunk), int(prefetch_every),
                                                      self._p(self.action), self._p(self.reward), self._p(self.done) if done_rows else None,
                                                      self._p(self.obs_words), K.AUTO_RESET if auto_reset else 0,
                                                      self._stream()))

    def rollout_random(self, episode=0, trace=False):
        """Whole random-policy games in one launch.  Returns dict of device tensors:
        scores [N,4] i16, nsteps [N] i16 and, with trace, step-major seats/masks/actions [48,N]."""
        n = self.n
        with torch.cuda.device(self.device):
            out = dict(scores=torch.empty((n, 4), dtype=torch.int16, device=self.device),
                       nsteps=torch.empty(n, dtype=torch.int16, device=self.device))
            if trace:
                out["seats"] = torch.empty((48, n), dtype=torch.int8, device=self.device)
                out["masks"] = torch.empty((48, n), dtype=torch.int64, device=self.device)
                out["actions"] = torch.empty((48, n), dtype=torch.uint8, device=self.device)
            _native.check(self.L.tarok_rollout_random(self._h, int(episode), self._p(out["scores"]), self._p(out["nsteps"]),
                                                      self._p(out.get("seats")), self._p(out.get("masks")),
                                                      self._p(out.get("actions")), self._stream()))
        return out

    def observe(self, out=None):
        """[N,256] bf16 features of the seat to move (include/tarok_env.h tarok_observe)."""
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((self.n, 256), dtype=torch.bfloat16, device=self.device)
            _native.check(self.L.tarok_observe(self._h, self._p(out), self._stream()))
        return out

    def observe_ref(self, out=None, meta=None):
        """The reference's own observation layout for the seat to move (tarok_observe_ref;
        Igralec.py:453-533).  Returns (record [N, REF_RECORD_BYTES] u8, meta [N,4] i32 = T, type, rows
        used, seat); `ref_views` slices the record into the reference's tensors."""
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((self.n, K.REF_RECORD_BYTES), dtype=torch.uint8, device=self.device)
            if meta is None:
                meta = torch.empty((self.n, 4), dtype=torch.int32, device=self.device)
            _native.check(self.L.tarok_observe_ref(self._h, self._p(out), self._p(meta), self._stream()))
        return out, meta

    @staticmethod
    def ref_views(record):
        """[N, REF_RECORD_BYTES] record -> dict of views named as in Igralec.py:455-519."""
        n = record.shape[0]
        cut = lambda a, b: record[:, a:b]
        return {"input_layer_nasprotiki": cut(K.REF_OPP, K.REF_OWN).reshape(n, K.REF_ROWS, 3, 54),
                "roka_input": cut(K.REF_OWN, K.REF_TALON).reshape(n, K.REF_ROWS, 54),
                "talon_input": cut(K.REF_TALON, K.REF_KING).reshape(n, 6, 55),
                "talon_input_klop": cut(K.REF_TALON, K.REF_TALON + 54),
                "barva_kralja": cut(K.REF_KING, K.REF_INDEX), "index_tistega_ki_igra": cut(K.REF_INDEX, K.REF_DISCARDS),
                "zalozil": cut(K.REF_DISCARDS, K.REF_LEGAL), "mozne_vec": cut(K.REF_LEGAL, K.REF_LEGAL + 54)}

    def observe_exchange_ref(self, out=None):
        """menjaj_talon_v_vektor (Igralec.py:535-543) for the games waiting for the exchange:
        [N, REF_EXCHANGE_BYTES] u8 = roka 54 | talon (54,6) | igra 15 | pad."""
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((self.n, K.REF_EXCHANGE_BYTES), dtype=torch.uint8, device=self.device)
            _native.check(self.L.tarok_observe_exchange_ref(self._h, self._p(out), self._stream()))
        return out

    def observe_hands_ref(self, out=None):
        """The bidding input (Igralec.py:278-281): [N,4,54] u8, every seat's hand one-hot."""
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((self.n, 4, 54), dtype=torch.uint8, device=self.device)
            _native.check(self.L.tarok_observe_hands_ref(self._h, self._p(out), self._stream()))
        return out

    def get_history(self):
        """[48,N] u8 device tensor: card p of every slot's current game (TarokVecEnv(history=True))."""
        with torch.cuda.device(self.device):
            h = torch.empty((48, self.n), dtype=torch.uint8, device=self.device)
            _native.check(self.L.tarok_get_history(self._h, self._p(h), self._stream()))
        return h

    def set_history(self, hist):
        h = self._dev(hist, torch.uint8, (48, self.n))
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_set_history(self._h, self._p(h), self._stream()))

    @staticmethod
    def mfma_weight_order(weight):
        """torch.nn.Linear.weight [out, 256] -> the bf16 fragment order tarok_policy_mlp reads
        (include/tarok_env.h): [out/32, 16 k-steps, 2 halves, 32 rows, 8] flattened."""
        out = weight.shape[0]
        assert weight.shape[1] == 256 and out % 32 == 0
        return weight.detach().to(torch.bfloat16).view(out // 32, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(out, 256)

    def policy_mlp(self, weights, obs_words, action_out=None, logp_out=None, value_out=None, features_out=None,
                   feature_words_out=None):
        """Fused learned-policy step (tarok_policy_mlp).  weights = (w1, b1, w2, b2, w3, b3): w* bf16 in
        mfma_weight_order() ([256,256], [256,256], [64,256]), b* f32; returns (action u8 [N], logp f32 [N],
        value f32 [N]).  features_out [N,256] bf16 and/or feature_words_out [N,4] int64 (the same features
        as bits: expand_feature_words) receive the network input for the learner."""
        w1, b1, w2, b2, w3, b3 = weights
        for w, shp in ((w1, (256, 256)), (w2, (256, 256)), (w3, (64, 256))):
            assert w.dtype == torch.bfloat16 and w.is_contiguous() and tuple(w.shape) == shp
        for b, k in ((b1, 256), (b2, 256), (b3, 64)):
            assert b.dtype == torch.float32 and b.is_contiguous() and tuple(b.shape) == (k,)
        with torch.cuda.device(self.device):
            if action_out is None:
                action_out = torch.empty(self.n, dtype=torch.uint8, device=self.device)
            if logp_out is None:
                logp_out = torch.empty(self.n, dtype=torch.float32, device=self.device)
            if value_out is None:
                value_out = torch.empty(self.n, dtype=torch.float32, device=self.device)
            _native.check(self.L.tarok_policy_mlp(self._h, self._p(w1), self._p(b1), self._p(w2), self._p(b2), self._p(w3),
                                                  self._p(b3), self._p(obs_words), self._p(action_out), self._p(logp_out),
                                                  self._p(value_out), self._p(features_out), self._p(feature_words_out),
                                                  self._stream()))
        return action_out, logp_out, value_out

    def policy_step(self, weights, obs_words, obs_out, action_out, logp_out=None, value_out=None, feature_words_out=None,
                    reward_out=None, done_out=None, auto_reset=True):
        """policy_mlp + step in one launch (tarok_policy_step): samples a card per game from the MLP
        policy on `obs_words` and plays it; obs_out receives the next observation words."""
        w1, b1, w2, b2, w3, b3 = weights
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_policy_step(self._h, self._p(w1), self._p(b1), self._p(w2), self._p(b2), self._p(w3),
                                                   self._p(b3), self._p(obs_words), self._p(action_out), self._p(logp_out),
                                                   self._p(value_out), self._p(feature_words_out), self._p(reward_out),
                                                   self._p(done_out), None, self._p(obs_out),
                                                   K.AUTO_RESET if auto_reset else 0, self._stream()))
        return action_out

    def ppo_loss(self, out, obs_words, action, logp_old, advantage, ret, weight, clip, vf_coef, ent_coef):
        """tarok_ppo_loss: (loss terms f32 [3] = weighted means of the policy loss, the squared value
        error and the entropy; d loss / d out [B,64] bf16) for loss = pi + vf_coef v - ent_coef H."""
        B = out.shape[0]
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and tuple(out.shape) == (B, 64)
        f = lambda t: t.to(torch.float32).contiguous()
        logp_old, advantage, ret, weight = f(logp_old), f(advantage), f(ret), f(weight)
        action = action.to(torch.int64).contiguous()
        obs_words = obs_words.to(torch.int64).contiguous()
        with torch.cuda.device(self.device):
            inv = (1.0 / weight.sum().clamp(min=1.0)).reshape(1).contiguous()
            dout = torch.empty_like(out)
            part = torch.empty(((B + 255) // 256, 4), dtype=torch.float32, device=self.device)
            _native.check(self.L.tarok_ppo_loss(self._h, int(B), self._p(out), self._p(obs_words), self._p(action),
                                                self._p(logp_old), self._p(advantage), self._p(ret), self._p(weight),
                                                float(clip), float(vf_coef), float(ent_coef), self._p(inv), self._p(dout),
                                                self._p(part), self._stream()))
            terms = part.sum(0)[:3] * inv
        return terms, dout

    def targets_ref(self, obs_before, action, trick, done, reward, next_q=None, factor=0.1):
        """The reference agent's transition targets (tarok_targets_ref; Igralec.py:387-446) of a recorded rollout:
        rows [T,N] (reward [T,N,4] recorded with reward_ref=True), T a multiple of 4 from a trick boundary.
        Returns (dy [T/4,N,4,54] f32, meta [T/4,N,4] u8)."""
        Tn = obs_before.shape[0]
        with torch.cuda.device(self.device):
            dy = torch.empty((Tn // 4, self.n, 4, 54), dtype=torch.float32, device=self.device)
            meta = torch.empty((Tn // 4, self.n, 4), dtype=torch.uint8, device=self.device)
            c = lambda t: None if t is None else t.contiguous()
            _native.check(self.L.tarok_targets_ref(self._h, int(Tn), self._p(c(obs_before)), self._p(c(action)), self._p(c(trick)),
                                                   self._p(c(done)), self._p(c(reward)), self._p(c(next_q)), float(factor),
                                                   self._p(dy), self._p(meta), self._stream()))
        return dy, meta

    # ---- the fused learner (include/tarok_env.h tarok_learn_*; driven by tarok_amd.selfplay.SelfPlay.update_fused)
    def learn_returns(self, T, done, reward, words, logp, val, act, reward_scale, rec, stats, scratch):
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_learn_returns(self._h, int(T), self._p(done), self._p(reward), self._p(words), self._p(logp),
                                                     self._p(val), self._p(act), float(reward_scale), self._p(rec), self._p(stats),
                                                     self._p(scratch), self._stream()))

    def learn_chain(self, B, words, index, rec, stats, clip, vf_coef, ent_coef, wf, bias, Xw, H1, H2, dOut, dH2, dH1, scratch, terms,
                    running=None):
        """wf: dict of the bf16 fragment-order weight copies (w1, w2, w3, w3t, w2t: learn_adam), bias: (b1, b2, b3) f32."""
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_learn_chain(self._h, int(B), self._p(words), self._p(index), self._p(rec), self._p(stats),
                                                   float(clip), float(vf_coef), float(ent_coef), self._p(wf["w1"]), self._p(bias[0]),
                                                   self._p(wf["w2"]), self._p(bias[1]), self._p(wf["w3"]), self._p(bias[2]),
                                                   self._p(wf["w3t"]), self._p(wf["w2t"]), self._p(Xw), self._p(H1), self._p(H2), self._p(dOut),
                                                   self._p(dH2), self._p(dH1), self._p(scratch), self._p(terms), self._p(running),
                                                   self._stream()))

    def learn_workspace_bytes(self):
        return int(self.L.tarok_learn_workspace_bytes(self._h))

    def learn_dw(self, B, Xw, H1, H2, dOut, dH2, dH1, terms, work, grad):
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_learn_dw(self._h, int(B), self._p(Xw), self._p(H1), self._p(H2), self._p(dOut),
                                                self._p(dH2), self._p(dH1), self._p(terms), self._p(work), self._p(grad), self._stream()))

    def learn_adam(self, param, grad, m, v, step, wf, lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=1.0, gnorm=None, apply=True):
        wf = wf or {}
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_learn_adam(self._h, self._p(param), self._p(grad), self._p(m), self._p(v), self._p(step), float(lr),
                                                  float(beta1), float(beta2), float(eps), float(max_norm), self._p(wf.get("w1")),
                                                  self._p(wf.get("w2")), self._p(wf.get("w3")), self._p(wf.get("w3t")),
                                                  self._p(wf.get("w2t")), self._p(gnorm), 1 if apply else 0, self._stream()))

    def gather_features(self, feature_words, index=None, out=None):
        """tarok_expand_features: [M,4] int64 feature words (+ optional int64 sample index [B]) ->
        [B,256] bf16 network input, gather and bit expansion in one kernel."""
        fw = feature_words.contiguous()
        B = fw.shape[0] if index is None else index.shape[0]
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((B, 256), dtype=torch.bfloat16, device=self.device)
            idx = None if index is None else index.to(torch.int64).contiguous()
            _native.check(self.L.tarok_expand_features(self._h, int(B), self._p(fw), self._p(idx), self._p(out), self._stream()))
        return out

    @staticmethod
    def expand_feature_words(words, dtype=torch.bfloat16):
        """[..., 4] int64 feature words (tarok_policy_mlp feature_words_out) -> [..., 256] 0/1 features."""
        b = words.contiguous().view(torch.uint8).view(*words.shape[:-1], 32, 1)
        shifts = torch.arange(8, device=words.device, dtype=torch.uint8)
        return ((b >> shifts) & 1).view(*words.shape[:-1], 256).to(dtype)

    def state(self):
        """Canonical lanes [10,N] (H0-3, P0-3, TAL, META) as host numpy uint64."""
        with torch.cuda.device(self.device):
            lanes = torch.empty((10, self.n), dtype=torch.int64, device=self.device)
            _native.check(self.L.tarok_get_state(self._h, self._p(lanes), self._stream()))
        return _u64(lanes)

    def set_state(self, lanes):
        """Restore from canonical lanes [10,N] (numpy uint64 / int64 tensor): inverse of state()."""
        if isinstance(lanes, np.ndarray):
            lanes = torch.from_numpy(np.ascontiguousarray(lanes).view(np.int64))
        lanes = self._dev(lanes, torch.int64, (10, self.n))
        with torch.cuda.device(self.device):
            _native.check(self.L.tarok_set_state(self._h, self._p(lanes), self._stream()))
        return self.legal_actions()

    def counters(self):
        """(episode[N] int64, score_sum[N,4] int32) host numpy."""
        with torch.cuda.device(self.device):
            ep = torch.empty(self.n, dtype=torch.int32, device=self.device)
            ss = torch.empty((self.n, 4), dtype=torch.int32, device=self.device)
            _native.check(self.L.tarok_get_counters(self._h, self._p(ep), self._p(ss), self._stream()))
        return ep.cpu().numpy().astype(np.int64), ss.cpu().numpy()
