"""CPU model of the device-side launch number (tarok_env.hip: launch_count / launch_counted / launch_phase).

Step launches of an env come in two grid sizes — kind 0: `groups` workgroups, kind 1: groups + ceil(groups / fan) — and
each kind counts ITS workgroups, as they start, in its own 256 sharded counters (wrapping at 64 x the shard's size);
a workgroup reads its own kind's counter (other workgroups of the running launch may or may not have added themselves
yet) and one counter of the other kind (at rest), and the launch number modulo 64 is the sum of the two quotients.
The model replays that arithmetic for random grids, random interleavings of the workgroups' reads and adds and random
sequences of launch kinds, through several wraps of the counters: every workgroup of launch L must compute L mod 64."""
import random

SHARDS, PHASES = 256, 64


def shard_size(grid, shard):
    return (grid + SHARDS - 1 - shard) // SHARDS


def other_shard(kind, block, groups):
    return (block if kind == 0 else block % groups) % SHARDS


def run(groups, fan, kinds, rnd):
    grid = {0: groups, 1: groups + (groups + fan - 1) // fan}
    counters = {0: [0] * SHARDS, 1: [0] * SHARDS}
    for L, kind in enumerate(kinds):
        g = grid[kind]
        # every workgroup reads, then adds; reads and adds of different workgroups interleave arbitrarily
        events = [(b, "read") for b in range(g)]
        rnd.shuffle(events)
        pending = []                                   # workgroups that have read and not yet added
        order = []
        for ev in events:
            order.append(ev)
            pending.append(ev[0])
            while pending and rnd.random() < 0.5:      # some earlier reader adds itself now
                order.append((pending.pop(rnd.randrange(len(pending))), "add"))
        order += [(b, "add") for b in pending]
        for b, what in order:
            s = b % SHARDS
            if what == "read":
                own = counters[kind][s]
                so = other_shard(kind, b, groups)
                assert shard_size(grid[1 - kind], so) >= 1, "the other kind's counter a workgroup reads must exist"
                oth = counters[1 - kind][so]
                phase = (own // shard_size(g, s) + oth // shard_size(grid[1 - kind], so)) % PHASES
                assert phase == L % PHASES, (groups, fan, L, kind, b, own, oth, phase)
            else:
                wrap = PHASES * shard_size(g, s) - 1    # atomicInc: old >= wrap ? 0 : old + 1
                counters[kind][s] = 0 if counters[kind][s] >= wrap else counters[kind][s] + 1


def test_every_workgroup_computes_the_launch_number_for_any_grid_and_any_mix_of_launch_kinds():
    rnd = random.Random(4)
    cases = [(256, 1), (1, 1), (2, 8), (79, 3), (255, 2), (257, 8), (1024, 4), (4096, 8)]
    cases += [(rnd.randint(1, 700), rnd.randint(1, 8)) for _ in range(12)]
    for groups, fan in cases:
        n = 150 if groups <= 1100 else 70               # more than two wraps of the 64 phases where it is cheap
        for style in range(3):
            if style == 0:
                kinds = [rnd.randint(0, 1) for _ in range(n)]
            elif style == 1:                             # long stretches of one kind, cut by single launches of the other
                kinds = [(0 if (i // 37) % 2 == 0 else 1) ^ (1 if i % 41 == 0 else 0) for i in range(n)]
            else:
                kinds = [0] * n if groups % 2 else [1] * n
            run(groups, fan, kinds, rnd)
