"""One process per GPU: which slice of the global game indices a rank owns, and the
(tiny) host-side reductions.  Games are independent (Tarok.py:33-35 builds N separate
Igra objects; only the score sum at Tarok.py:59-61 combines them), so the env path
needs NO collective: the deal RNG is keyed by the GLOBAL game index, each rank plays
`game_offset .. game_offset+n_local` and the totals are summed once at the end."""
import os


def world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def weak_shard(n_per_gpu, rank):
    """Weak scaling (BASELINE config 5: 8 x 65,536): every rank holds n_per_gpu games."""
    return rank * n_per_gpu, n_per_gpu


def strong_shard(n_total, rank, world_size):
    """Contiguous split of n_total games; the first n_total % world ranks get one more."""
    base, extra = divmod(n_total, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def init_process_group(backend=None):
    """nccl (= RCCL) when a GPU is visible, gloo otherwise; rendezvous on 127.0.0.1."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend, rank=world()[0], world_size=world()[2])


def _reduce(values, op):
    """All-reduce a short list of floats/ints; returns a list.  Uses a device tensor under
    nccl (RCCL) and a host tensor under gloo, so the same code runs in CPU rehearsals."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return list(values)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=op)
    return t.cpu().tolist()


def sum_over_ranks(values):
    """SUM over ranks of a short list (score totals, step counts)."""
    import torch.distributed as dist
    return _reduce(values, dist.ReduceOp.SUM)


def max_over_ranks(values):
    import torch.distributed as dist
    return _reduce(values, dist.ReduceOp.MAX)


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def gather_over_ranks(value):
    """[value of rank 0, ..., value of rank W-1] on every rank (one float per rank: per-rank timings of a bench line)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    w, r = dist.get_world_size(), dist.get_rank()
    return sum_over_ranks([float(value) if k == r else 0.0 for k in range(w)])


def group_info():
    """{"world_size", "backend"} of the process group the collectives above run on (1 / None without one)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return {"world_size": dist.get_world_size(), "backend": dist.get_backend()}
    return {"world_size": 1, "backend": None}
