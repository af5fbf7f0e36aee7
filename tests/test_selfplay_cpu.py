"""Host logic of the self-play harness (no GPU): return assignment, masked sampling, the
flattened gradient all-reduce under gloo world_size 2."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from tarok_amd import selfplay as SP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_legal_matrix_and_sampling():
    words = torch.tensor([0b1011, (1 << 53) | 1, 0, (1 << 54) - 1], dtype=torch.int64)
    legal = SP.legal_matrix(words)
    assert legal.shape == (4, 54)
    assert legal[0].nonzero().flatten().tolist() == [0, 1, 3]
    assert legal[1].nonzero().flatten().tolist() == [0, 53]
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(4, 54)
    for _ in range(50):
        a, logp = SP.sample_masked(logits, legal, g)
        assert a[2].item() == 255 and logp[2].item() == 0.0
        for r in (0, 1, 3):
            assert legal[r, a[r]].item() and logp[r].item() <= 0
    # probabilities renormalise over the legal set
    a, logp = SP.sample_masked(torch.zeros(1, 54), legal[:1], g)
    assert abs(logp.exp().item() - 1 / 3) < 1e-6


def test_assign_returns_matches_a_plain_loop():
    rnd = np.random.RandomState(0)
    T, N = 40, 7
    done = rnd.rand(T, N) < 0.15
    reward = np.where(done[..., None], rnd.randint(-90, 91, (T, N, 4)), 0).astype(np.int16)
    seat = rnd.randint(0, 4, (T, N))
    ret, known = SP.assign_returns(torch.from_numpy(done), torch.from_numpy(reward), torch.from_numpy(seat))
    for g in range(N):
        for t in range(T):
            later = [u for u in range(t, T) if done[u, g]]
            if later:
                assert known[t, g].item() and ret[t, g].item() == reward[later[0], g, seat[t, g]]
            else:
                assert not known[t, g].item()


def test_epoch_permutation_is_a_permutation_that_spreads():
    """The fused learner's epoch order j -> (a j + b) mod M (SelfPlay._epoch_permutation): a permutation of the M
    samples for the rollout sizes in use, different from epoch to epoch, and every minibatch (a run of it) reaches
    all lock-steps of the rollout."""
    import types
    for M in (48 * 65536, 48 * 4096, 24 * 4096 + 7, 1000003):
        me = types.SimpleNamespace(shuffle="affine", hgen=torch.Generator().manual_seed(1), device=torch.device("cpu"), gen=None)
        p1 = SP.SelfPlay._epoch_permutation(me, M)
        p2 = SP.SelfPlay._epoch_permutation(me, M)
        assert p1.dtype == torch.int64 and p1.numel() == M
        assert torch.equal(torch.sort(p1).values, torch.arange(M)) and torch.equal(torch.sort(p2).values, torch.arange(M))
        assert not torch.equal(p1, p2)
        if M % 48 == 0:
            n = M // 48
            for idx in p1.chunk(8):
                assert len(torch.unique(idx // n)) == 48            # every lock-step (row t = sample // n) in every minibatch


def _flat_worker(rank, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from tarok_amd import selfplay, sharding, karte
    sharding.init_process_group("gloo")
    g = torch.full((karte.MLP_PARAMS,), float(rank + 1))
    g[::7] = -2.0 * (rank + 1)
    nbytes = selfplay.allreduce_flat(g)
    q.put((rank, g.numpy(), nbytes))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks_gloo():
    """The fused learner's collective: ONE all-reduce of the flat gradient vector (TAROK_MLP_PARAMS floats), averaged."""
    from tarok_amd import karte
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    ps = [ctx.Process(target=_flat_worker, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, n0), (_, g1, n1) = got
    assert n0 == n1 == karte.MLP_PARAMS * 4
    want = np.full(karte.MLP_PARAMS, 1.5, np.float32)
    want[::7] = -3.0
    np.testing.assert_allclose(g0, want)
    np.testing.assert_allclose(g1, want)


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from tarok_amd import selfplay, sharding
    sharding.init_process_group("gloo")
    torch.manual_seed(0)
    net = selfplay.PolicyNet(32)
    x = torch.full((5, 256), float(rank + 1))
    logits, v = net(x)
    (logits.sum() + v.sum()).backward()
    local = [p.grad.clone() for p in net.parameters()]
    nbytes = selfplay.allreduce_gradients(list(net.parameters()))
    q.put((rank, [g.numpy() for g in local], [p.grad.numpy() for p in net.parameters()], nbytes))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    ps = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, a0, n0), (_, l1, a1, n1) = got
    assert n0 == n1 > 0
    for g0, g1, r0, r1 in zip(l0, l1, a0, a1):
        np.testing.assert_allclose(r0, (g0 + g1) / 2, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r0, r1)


def test_split_k_linear_gives_the_plain_gradients():
    """_LinearSplitK (weight gradient as a batch of partial products) vs nn.Linear on the same
    minibatch: outputs and all gradients agree (f32 on the CPU: to rounding of the summation order)."""
    import torch
    from tarok_amd import selfplay as SP
    torch.manual_seed(0)
    net = SP.PolicyNet(256)
    x = torch.randn(16384 * 2, 256)

    def run(split):
        net.zero_grad()
        if split:
            out = net.forward_raw(x)                          # >= 16384 rows with grad enabled: the split-K path
        else:
            h = torch.relu(net.fc1(x)); h = torch.relu(net.fc2(h)); out = net.head(h)
        (out ** 2).mean().backward()
        return [p.grad.clone() for p in net.parameters()], out.detach()

    g1, o1 = run(True)
    g0, o0 = run(False)
    assert torch.equal(o1, o0)
    for a, b in zip(g1, g0):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-7)
