"""The N>1 path on CPU: two gloo ranks each own a slice of the global game indices
(weak and strong sharding), play it — here on the CPU oracle, which stands in for the
per-rank GPU env — and sum their totals; the result must equal one process playing
all the games.  Proves offsets/RNG keying/reductions are sharding-independent."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, n_total, seed, mix, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import oracle as O
    from tarok_amd import sharding
    sharding.init_process_group("gloo")
    assert sharding.world() == (rank, rank, world_size)
    off, cnt = sharding.strong_shard(n_total, rank, world_size)
    r = O.rollout(seed, off, cnt, 0, mix, trace=False)
    totals = sharding.sum_over_ranks(list(r["scores"].astype(np.int64).sum(0)) + [r["total_steps"]])
    t = sharding.max_over_ranks([float(rank + 1)])
    sharding.barrier()
    woff, wcnt = sharding.weak_shard(100, rank)
    if rank == 0:
        q.put(([int(x) for x in totals], t[0], (woff, wcnt)))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    from oracle import oracle as O
    from oracle import tarok_spec as S
    n_total, seed = 1001, 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, seed, S.MIX_ALL, q)) for r in range(2)]
    for p in procs:
        p.start()
    totals, tmax, weak = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = O.rollout(seed, 0, n_total, 0, S.MIX_ALL, trace=False)
    assert totals[:4] == list(ref["scores"].astype(np.int64).sum(0))
    assert totals[4] == ref["total_steps"]
    assert tmax == 2.0 and weak == (0, 100)


def test_shard_arithmetic():
    from tarok_amd import sharding
    for n, w in [(10, 3), (65536, 8), (7, 8), (1001, 2)]:
        parts = [sharding.strong_shard(n, r, w) for r in range(w)]
        assert sum(c for _, c in parts) == n
        pos = 0
        for off, cnt in parts:
            assert off == pos
            pos += cnt
    assert sharding.weak_shard(65536, 3) == (3 * 65536, 65536)
