"""Self-play policy-gradient harness on top of the GPU env (BASELINE.json configs 4-5,
SURVEY §8f row 4).  Build-owned: the reference trains a double-Q LSTM agent with
pytorch-lightning (Igralec.py:545-714) and has no PPO, so nothing here is pinned to it —
this exists to show the env being driven by a learner and to exercise the one collective
of the whole build, the gradient all-reduce (RCCL over xGMI via torch.distributed).

Shape of the loop (per rank, one env shard each; games need no communication):

    obs  = env.observe()                        [N,256] bf16 features     (HIP kernel)
    pi,v = net(obs)                             small MLP, bf16           (torch / rocBLAS)
    a    = masked categorical sample            legal mask = observation word
    env.step(a, auto_reset=True)                                          (HIP kernel)
    ... T steps ...
    returns: every card is credited with its seat's final score of that game (Monte-Carlo,
    gamma = 1); clipped-surrogate PPO update; gradients summed over ranks in ONE flattened
    all-reduce per minibatch (a few hundred KB: latency-bound, so one bucket, not many).
"""
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import karte as K
from . import sharding


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b whose weight gradient is computed as a batch of partial products.

    dW = dY^T X has M = N = 256 or 64 and K = the minibatch (393,216 rows in the bench): as ONE
    GEMM it has 16 output tiles, i.e. 240 of the 256 CUs idle (hipBLASLt picks a 64x64x256 tile:
    0.81 ms, 2.5 % of the matrix peak, 44 % of the whole update).  Split along K into S batches
    (torch.bmm) it is S x 16 tiles; the partials are summed in f32."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        cd = torch.bfloat16 if (x.is_cuda and torch.is_autocast_enabled()) else x.dtype
        with torch.autocast(x.device.type, enabled=False):
            xq, wq, bq = x.to(cd), w.to(cd), b.to(cd)
            if relu and x.is_cuda:
                y = torch._addmm_activation(bq, xq, wq.t())      # bias + ReLU in the GEMM's epilogue
            else:
                y = F.linear(xq, wq, bq)
                if relu:
                    y = F.relu(y)
            ctx.save_for_backward(xq, wq, y if relu else None)
            ctx.out_dtypes = (x.dtype, w.dtype, b.dtype)
            return y

    @staticmethod
    def backward(ctx, gy):
        xq, wq, y = ctx.saved_tensors
        dx, dw, db = ctx.out_dtypes
        gy = gy.contiguous()
        if y is not None:
            gy = torch.ops.aten.threshold_backward(gy, y, 0)     # ReLU'
        gx = (gy @ wq).to(dx) if ctx.needs_input_grad[0] else None
        B = xq.shape[0]
        S = next((s for s in (32, 24, 16, 12, 8, 4, 2) if B % s == 0 and B // s >= 512), 1)      # S x 16 tiles >= 256 CUs
        if S > 1:
            gw = torch.bmm(gy.view(S, B // S, -1).transpose(1, 2), xq.view(S, B // S, -1)).sum(0, dtype=torch.float32)
        else:
            gw = (gy.t() @ xq).float()
        return gx, gw.to(dw), gy.sum(0, dtype=torch.float32).to(db), None


class PolicyNet(nn.Module):
    """256 features -> 2 x hidden ReLU -> one 64-wide head: outputs 0..53 = card logits, output
    54 = state value (the layout tarok_policy_mlp evaluates in one fused MFMA kernel when hidden = 256)."""

    def __init__(self, hidden=256):
        super().__init__()
        self.fc1 = nn.Linear(256, hidden)
        self.fc2 = nn.Linear(hidden, hidden)
        self.head = nn.Linear(hidden, 64)

    def forward_raw(self, x):
        """all 64 head outputs [N,64]"""
        if x.dim() == 2 and x.shape[0] >= 16384 and torch.is_grad_enabled():    # a training minibatch
            h = _LinearSplitK.apply(x, self.fc1.weight, self.fc1.bias, True)
            h = _LinearSplitK.apply(h, self.fc2.weight, self.fc2.bias, True)
            return _LinearSplitK.apply(h, self.head.weight, self.head.bias, False)
        h = F.relu(self.fc1(x))
        h = F.relu(self.fc2(h))
        return self.head(h)

    def forward(self, x):
        out = self.forward_raw(x)
        return out[..., :54], out[..., 54]


def legal_matrix(mask_words):
    """int64 [N] observation words -> bool [N,54] legal-card matrix."""
    bits = torch.arange(54, device=mask_words.device, dtype=torch.int64)
    return ((mask_words.unsqueeze(-1) >> bits) & 1).bool()


def sample_masked(logits, legal, generator=None):
    """Sample one legal card per row.  Returns (action int64 [N], log-prob f32 [N]).
    Rows without any legal card (finished games when auto-reset is off) get action 255."""
    none = ~legal.any(dim=-1)
    lg = logits.float().masked_fill(~legal, float("-inf"))
    lg = torch.where(none.unsqueeze(-1), torch.zeros_like(lg), lg)
    logp_all = F.log_softmax(lg, dim=-1)
    action = torch.multinomial(logp_all.exp(), 1, generator=generator).squeeze(-1)
    logp = logp_all.gather(-1, action.unsqueeze(-1)).squeeze(-1)
    return torch.where(none, torch.full_like(action, 255), action), torch.where(none, torch.zeros_like(logp), logp)


def assign_returns(done, reward, seat):
    """Credit every transition with its seat's final score of the game it belongs to.

    done [T,N] bool: the card played at t finished a game; reward [T,N,4]: scores by seat,
    valid where done; seat [T,N]: who played at t.  Returns (ret [T,N] f32, known [T,N] bool):
    `known` is False for the cards of games still unfinished when the rollout ends."""
    T, N = done.shape
    ret = torch.zeros((T, N), dtype=torch.float32, device=done.device)
    known = torch.zeros((T, N), dtype=torch.bool, device=done.device)
    cur = torch.zeros((N, 4), dtype=torch.float32, device=done.device)
    have = torch.zeros(N, dtype=torch.bool, device=done.device)
    for t in range(T - 1, -1, -1):
        d = done[t]
        cur = torch.where(d.unsqueeze(-1), reward[t].float(), cur)
        have = have | d
        ret[t] = cur.gather(-1, seat[t].long().unsqueeze(-1)).squeeze(-1)
        known[t] = have
    return ret, known


def allreduce_flat(flat):
    """Average ONE flat gradient vector over all ranks in place (the fused learner's gradients already are one
    buffer: no gather, no scatter).  No-op without an initialised process group / with one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    return flat.numel() * flat.element_size()


def allreduce_gradients(params):
    """Sum the gradients of `params` over all ranks in ONE flattened all-reduce and
    average them.  No-op without an initialised process group / with one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return flat.numel() * flat.element_size()


class SelfPlay:
    """All four seats of every game share one policy.

    The rollout is captured once into a graph (torch.cuda.graph: the library's kernels are
    launched on torch's capture stream) and replayed: per lock-step ONE tarok_policy_mlp launch
    (features -> MLP on the matrix cores -> masked sample, hidden = 256) and one tarok_step
    launch, both writing straight into their rows of static rollout buffers.  With another
    hidden size the policy runs as tarok_observe -> torch GEMMs -> tarok_sample_policy."""

    def __init__(self, env, hidden=256, lr=3e-4, clip=0.2, vf_coef=0.5, ent_coef=0.01, reward_scale=1.0 / 70.0, seed=0,
                 use_graph=True, fused=None, fused_loss=None, fused_step=None, fused_learner=None, max_grad_norm=1.0):
        self.env = env
        self.device = env.device
        torch.manual_seed(seed)                       # same initial weights on every rank
        self.net = PolicyNet(hidden).to(self.device)
        self.lr, self.max_grad_norm = lr, max_grad_norm
        # ONE flat parameter vector (tarok_env.h TAROK_MLP_*): the module's parameters are views into it, so the
        # torch paths and the fused learner (tarok_learn_*: flat gradient, flat Adam state) see the same weights
        self.flat = None
        if hidden == 256:
            ps = [self.net.fc1.weight, self.net.fc1.bias, self.net.fc2.weight, self.net.fc2.bias, self.net.head.weight, self.net.head.bias]
            self.flat = torch.cat([p.detach().reshape(-1) for p in ps]).contiguous()
            assert self.flat.numel() == K.MLP_PARAMS
            off = 0
            for p in ps:
                p.data = self.flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr)
        self.clip, self.vf_coef, self.ent_coef, self.reward_scale = clip, vf_coef, ent_coef, reward_scale
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(1234 + 7919 * sharding.world()[0])
        self.hgen = torch.Generator()                 # host side: the fused learner's epoch permutations
        self.hgen.manual_seed(4321 + 7919 * sharding.world()[0])
        self.shuffle = "affine"
        self.obs_words = env.reset().words.clone()
        self.use_graph = use_graph
        self.fused = (hidden == 256) if fused is None else bool(fused)
        assert not self.fused or hidden == 256, "tarok_policy_mlp is built for hidden = 256"
        # the loss and its gradient in one kernel (tarok_ppo_loss) instead of ~40 framework kernels
        self.fused_loss = True if fused_loss is None else bool(fused_loss)
        # policy and env step in one launch per lock-step (tarok_policy_step)
        self.fused_step = self.fused if fused_step is None else bool(fused_step)
        # the whole update as tarok_learn_* launches: returns, forward + loss + backward chain, weight gradients,
        # clip + Adam on the flat vectors (update_fused)
        self.fused_learner = (self.fused and env.device.type == "cuda") if fused_learner is None else bool(fused_learner)
        assert not self.fused_learner or self.fused, "the fused learner is built for the fused policy (hidden = 256)"
        self._graph, self._buf, self._T = None, None, 0
        self._w = None                                # rollout copies of the weights (bf16) / biases (f32)
        self._learn = None                            # the fused learner's buffers
        if self.fused_learner:
            dev = self.device
            bf = lambda k: torch.empty(k, dtype=torch.bfloat16, device=dev)
            self._wf = dict(w1=bf(65536), w2=bf(65536), w3=bf(16384), w3t=bf(16384), w2t=bf(65536))
            self.gflat = torch.zeros(K.MLP_PARAMS, dtype=torch.float32, device=dev)
            self.adam_m = torch.zeros_like(self.gflat)
            self.adam_v = torch.zeros_like(self.gflat)
            self.adam_step = torch.zeros(1, dtype=torch.int32, device=dev)
            self.sync_weights()
            f = self.flat
            sl = lambda a, k: f[a:a + k]
            # the rollout reads the learner's own copies: fragment-order weights, biases inside the flat vector
            self._w = [self._wf["w1"].view(256, 256), sl(K.MLP_B1, 256), self._wf["w2"].view(256, 256), sl(K.MLP_B2, 256),
                       self._wf["w3"].view(64, 256), sl(K.MLP_B3, 64)]

    def sync_weights(self):
        """Rebuild the kernels' bf16 fragment-order weight copies from the flat parameter vector (after loading a
        checkpoint or changing the parameters by hand; the fused Adam step keeps them current itself)."""
        self.env.learn_adam(self.flat, None, None, None, None, self._wf, apply=False)

    def _refresh_rollout_weights(self):
        if self.fused_learner:                        # (tarok_learn_adam wrote them with the update)
            return
        with torch.no_grad():
            n = self.net
            src = [(n.fc1.weight, torch.bfloat16), (n.fc1.bias, torch.float32), (n.fc2.weight, torch.bfloat16),
                   (n.fc2.bias, torch.float32), (n.head.weight, torch.bfloat16), (n.head.bias, torch.float32)]
            conv = (lambda p, dt: self.env.mfma_weight_order(p) if (self.fused and dt == torch.bfloat16)
                    else p.detach().to(dt).contiguous().clone())
            if self._w is None:
                self._w = [conv(p, dt) for p, dt in src]
            else:
                for d, (p, dt) in zip(self._w, src):
                    d.copy_(conv(p, dt))              # in place: the captured graph reads these tensors

    def _alloc(self, T):
        n, dev = self.env.n, self.device
        self._T = T
        # the network input of every step: as 4 x 64 feature bits per game on the fused path (expanded
        # per minibatch in update()), as [256] bf16 otherwise
        obs = (torch.empty((T, n, 4), dtype=torch.int64, device=dev) if self.fused
               else torch.empty((T, n, 256), dtype=torch.bfloat16, device=dev))
        self._buf = dict(obs=obs,
                         words=torch.empty((T + 1, n), dtype=torch.int64, device=dev),
                         act=torch.empty((T, n), dtype=torch.uint8, device=dev),
                         logp=torch.empty((T, n), dtype=torch.float32, device=dev),
                         val=torch.empty((T, n), dtype=torch.float32, device=dev),
                         done=torch.empty((T, n), dtype=torch.uint8, device=dev),
                         reward=torch.zeros((T, n, 4), dtype=torch.int16, device=dev))   # written only where done
        self._graph = None

    def _rollout_body(self, T):
        env, buf, w = self.env, self._buf, self._w
        for t in range(T):
            if self.fused and self.fused_step:
                env.policy_step(w, buf["words"][t], buf["words"][t + 1], buf["act"][t], buf["logp"][t], buf["val"][t],
                                feature_words_out=buf["obs"][t], reward_out=buf["reward"][t], done_out=buf["done"][t])
                continue
            if self.fused:
                env.policy_mlp(w, buf["words"][t], buf["act"][t], buf["logp"][t], buf["val"][t], feature_words_out=buf["obs"][t])
            else:
                env.observe(buf["obs"][t])
                h = F.relu(F.linear(buf["obs"][t], w[0], w[1].to(torch.bfloat16)))
                h = F.relu(F.linear(h, w[2], w[3].to(torch.bfloat16)))
                out = F.linear(h, w[4], w[5].to(torch.bfloat16))
                buf["val"][t].copy_(out[:, 54])
                env.sample_policy(out, buf["words"][t], buf["act"][t], buf["logp"][t])
            env.step(buf["act"][t], auto_reset=True, obs_out=buf["words"][t + 1], reward_out=buf["reward"][t],
                     done_out=buf["done"][t])

    def _collect_body(self, T):
        """Everything one rollout does on the device, in launch order (captured as ONE graph: the
        weight conversion and the buffer housekeeping are ~20 small launches that would otherwise
        cost more host time than the 48 lock-steps take on the GPU)."""
        buf = self._buf
        self._refresh_rollout_weights()
        buf["reward"].zero_()
        buf["words"][0].copy_(self.obs_words)
        self._rollout_body(T)
        self.obs_words.copy_(buf["words"][T])

    @torch.no_grad()
    def collect(self, T):
        """T lock-steps of self-play into the static rollout buffers."""
        if self._buf is None or self._T != T:
            self._alloc(T)
        buf = self._buf
        if not self.use_graph:
            self._collect_body(T)
            return buf
        if self._graph is None:
            # warm up outside capture (lazy library / GEMM-workspace initialisation, rollout copies of
            # the weights allocated), then continue from the warmed-up env state
            s = torch.cuda.Stream(self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                self._collect_body(2)                 # (any number: the refill-list parity lives on the device)
            torch.cuda.current_stream(self.device).wait_stream(s)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._collect_body(T)
            self._graph = g
        self._graph.replay()
        return buf

    def update(self, buf, epochs=2, minibatches=8):
        """Clipped-surrogate policy-gradient update on one rollout.  Returns stats."""
        T, n = buf["act"].shape
        words_t = buf["words"][:T]
        seat = (words_t >> K.OBS_SEAT_SHIFT) & 3
        ret, known = assign_returns(buf["done"].bool(), buf["reward"], seat)
        ret = ret * self.reward_scale
        flat = lambda x: x.reshape(T * n, *x.shape[2:])
        obs, words, act, logp0 = flat(buf["obs"]), flat(words_t), flat(buf["act"]).long(), flat(buf["logp"])
        val0 = flat(buf["val"]).float()
        ret, known = flat(ret), flat(known)
        adv = ret - val0
        m = known.float()
        mean = (adv * m).sum() / m.sum().clamp(min=1)
        std = (((adv - mean) ** 2 * m).sum() / m.sum().clamp(min=1)).sqrt().clamp(min=1e-6)
        adv = (adv - mean) / std
        stats = dict(loss=0.0, pi_loss=0.0, v_loss=0.0, entropy=0.0, allreduce_bytes=0, known_frac=float(m.mean()))
        sums = torch.zeros(4, dtype=torch.float32, device=self.device)     # loss terms summed on the device: no host sync per minibatch
        params = [p for p in self.net.parameters()]
        count = 0
        for _ in range(epochs):
            perm = torch.randperm(T * n, device=self.device, generator=self.gen)
            for idx in perm.chunk(minibatches):
                x = self.env.gather_features(obs, idx) if self.fused else obs[idx]
                w = m[idx]
                a = adv[idx]
                if self.fused_loss:
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        out = self.net.forward_raw(x)
                    out = out.to(torch.bfloat16).contiguous()
                    terms, dout = self.env.ppo_loss(out.detach(), words[idx], act[idx], logp0[idx], a, ret[idx], w,
                                                    self.clip, self.vf_coef, self.ent_coef)
                    pi_loss, v_loss, ent = terms[0], terms[1], terms[2]
                    loss = pi_loss + self.vf_coef * v_loss - self.ent_coef * ent
                    self.opt.zero_grad(set_to_none=True)
                    out.backward(dout)
                else:
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        logits, val = self.net(x)
                    legal = legal_matrix(words[idx] & K.OBS_MASK)
                    lg = logits.float().masked_fill(~legal, float("-inf"))
                    logp_all = F.log_softmax(lg, dim=-1)
                    logp = logp_all.gather(-1, act[idx].clamp(max=53).unsqueeze(-1)).squeeze(-1)
                    wsum = w.sum().clamp(min=1)
                    ratio = (logp - logp0[idx]).exp()
                    pi_loss = -(torch.min(ratio * a, ratio.clamp(1 - self.clip, 1 + self.clip) * a) * w).sum() / wsum
                    v_loss = (((val.float() - ret[idx]) ** 2) * w).sum() / wsum
                    p = logp_all.exp()
                    ent = (-(p * torch.where(legal, logp_all, torch.zeros_like(logp_all))).sum(-1) * w).sum() / wsum
                    loss = pi_loss + self.vf_coef * v_loss - self.ent_coef * ent
                    self.opt.zero_grad(set_to_none=True)
                    loss.backward()
                stats["allreduce_bytes"] = allreduce_gradients(params)
                nn.utils.clip_grad_norm_(params, 1.0)
                self.opt.step()
                count += 1
                sums += torch.stack([loss.detach().float(), pi_loss.detach().float(), v_loss.detach().float(), ent.detach().float()])
        for k, v in zip(("loss", "pi_loss", "v_loss", "entropy"), sums.tolist()):
            stats[k] = v / max(1, count)
        return stats

    def _learn_bufs(self, M, B):
        """Buffers of the fused learner: per-sample records of a rollout of M samples, activations of a minibatch of
        at most B samples, the weight-gradient workspace."""
        lb = self._learn
        if lb is None or lb["M"] != M or lb["B"] < B:
            dev = self.device
            f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            act = lambda k: torch.zeros((B + K.LEARN_PAD, k), dtype=torch.bfloat16, device=dev)
            lb = dict(M=M, B=B, rec=f32(M, 4), stats=f32(4), scratch=f32(max((self.env.n + 255) // 256, (B + 95) // 96), 4),
                      H1=act(256), H2=act(256), dH2=act(256), dH1=act(256), dOut=act(64), terms=f32(4),
                      Xw=torch.zeros((B + K.LEARN_PAD, 4), dtype=torch.int64, device=dev),
                      running=torch.zeros(4, dtype=torch.float32, device=dev),
                      work=torch.empty(self.env.learn_workspace_bytes(), dtype=torch.uint8, device=dev))
            self._learn = lb
        return lb

    def _epoch_permutation(self, M):
        """The order in which an epoch visits the rollout's M samples: j -> (a j + b) mod M with a coprime to M and (a, b)
        drawn per epoch — a permutation whose consecutive entries lie a samples apart, so every minibatch (a run of it)
        spreads over all lock-steps and slots.  Three elementwise launches instead of torch.randperm's radix sort of M
        keys (0.18 ms of a 5 ms update at 3.1 M samples).  shuffle = "randperm" selects the latter."""
        if self.shuffle == "randperm":
            return torch.randperm(M, device=self.device, generator=self.gen)
        import math
        while True:
            a = int(torch.randint(M // 3, M, (1,), generator=self.hgen)) | 1
            if math.gcd(a, M) == 1:
                break
        b = int(torch.randint(0, M, (1,), generator=self.hgen))
        return (torch.arange(M, device=self.device, dtype=torch.int64) * a + b) % M

    def update_fused(self, buf, epochs=2, minibatches=8):
        """The update as fused launches (include/tarok_env.h tarok_learn_*): per rollout one returns kernel; per
        minibatch the forward + loss + backward chain (activations in LDS, bf16 MFMA), the three weight gradients
        as one split-K launch, ONE flat gradient all-reduce, and clip + Adam + weight-copy refresh in one launch.
        No host synchronisation until the statistics are read at the end."""
        env = self.env
        T, n = buf["act"].shape
        M = T * n
        B = -(-M // minibatches)
        lb = self._learn_bufs(M, B)
        env.learn_returns(T, buf["done"], buf["reward"], buf["words"][:T], buf["logp"], buf["val"], buf["act"], self.reward_scale,
                          lb["rec"], lb["stats"], lb["scratch"])
        words = buf["obs"].view(M, 4)
        lb["running"].zero_()
        nbytes = 0
        bias = (self._w[1], self._w[3], self._w[5])
        for _ in range(epochs):
            perm = self._epoch_permutation(M)
            for idx in perm.chunk(minibatches):
                b = idx.numel()
                env.learn_chain(b, words, idx, lb["rec"], lb["stats"], self.clip, self.vf_coef, self.ent_coef, self._wf, bias,
                                lb["Xw"], lb["H1"], lb["H2"], lb["dOut"], lb["dH2"], lb["dH1"], lb["scratch"], lb["terms"], lb["running"])
                env.learn_dw(b, lb["Xw"], lb["H1"], lb["H2"], lb["dOut"], lb["dH2"], lb["dH1"], lb["terms"], lb["work"], self.gflat)
                nbytes = allreduce_flat(self.gflat)
                env.learn_adam(self.flat, self.gflat, self.adam_m, self.adam_v, self.adam_step, self._wf, lr=self.lr,
                               max_norm=self.max_grad_norm)
        run = lb["running"].tolist()
        cnt = max(1.0, run[3])
        pi, v, ent = run[0] / cnt, run[1] / cnt, run[2] / cnt
        return dict(loss=pi + self.vf_coef * v - self.ent_coef * ent, pi_loss=pi, v_loss=v, entropy=ent, allreduce_bytes=nbytes,
                    known_frac=float(lb["stats"][2]))

    def iterate(self, T=48, epochs=2, minibatches=8):
        """One rollout + one update, timed.  Returns stats incl. env steps/s of the rollout and of the whole
        iteration (rollout + update)."""
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        buf = self.collect(T)
        torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        stats = (self.update_fused if self.fused_learner else self.update)(buf, epochs, minibatches)
        torch.cuda.synchronize(self.device)
        t2 = time.perf_counter()
        stats.update(rollout_s=t1 - t0, update_s=t2 - t1, env_steps=T * self.env.n,
                     rollout_steps_per_s=T * self.env.n / (t1 - t0), iteration_steps_per_s=T * self.env.n / (t2 - t0),
                     mean_score=float(buf["reward"].float().sum() / buf["done"].float().sum().clamp(min=1) / 4),
                     env_errors=int((buf["words"] < 0).any()))
        return stats
