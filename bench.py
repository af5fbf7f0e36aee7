#!/usr/bin/env python3
"""Headline benchmark: env steps/s at 65,536 parallel 4-player games per MI355X
(BASELINE.json metric; workload = configs[2]: mixed Klop/Berac/Navadna contracts,
uniform-random policy, synthetic deals).

    python bench.py --gpus 1 --steps 100 --warmup 10          (the defaults)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One bench "step" = one pass of the hot path over the batch = ONE launch of
tarok_krog_random: `--cards-per-launch` (default 128 = thirty-two tricks) lock-steps of every
one of a rank's 65,536 games.  One lock-step = one card played in each game
(Tarok.py:48-56): legal mask of the seat to move, a uniform random legal card (the Bot
policy, Igralec.py:158-159), the card applied, trick resolution and scoring, finished
games replaced at once (auto-reset: every slot is live in every lock-step), next
observation written.  The state stays in registers for the cards of one launch and is
resident in HBM between launches; every per-card output (action, observation word, done,
scores) is written to HBM.
    value = games x cards per launch x steps x ranks / max-over-ranks time   [env steps/s]
Exactly --steps launches are timed (never fewer than one; --warmup is raised to one whole
graph, so that the capture is untimed, and to ~25 ms of launches, so that the GPU has left its
idle clocks: `warmup` reports what ran).  Weak scaling: each rank
owns its own 65,536 games (global game indices rank*65536...), no collective in the env path.

BASELINE.md's protocol: `value` is the first timed region (seed 0: the contract's number); the further
regions of --repeats rotate the seeds {0, 1, 2} (one env per seed, each warmed up and graph-captured untimed)
and are reported as a spread and per seed.

Extra objects on the JSON line:
  roofline           the dominant kernel (k_play_wide: tarok_krog_random) against HBM peak.  `achieved` =
                     HBM bytes per launch / launch duration (HIP events on the launch stream
                     over the timed region / launches).  For launches of several cards the
                     bytes are the MEASURED ones (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE of
                     this command, profiles/): such a launch keeps the state in registers and
                     does not move the 54 algorithmic B/step of SURVEY 8d, which are reported
                     beside it as `algorithmic` (that figure is not a bound: it can exceed 1).
  roofline_step_api  the reset()/step()/legal_actions() surface an external policy drives
                     (tarok_policy_random + tarok_step, one card per launch: k_policy + k_step), where
                     54 B/step IS the right accounting; `streaming` = the same path at 4 M games, where
                     the batch streams through HBM (at 65,536 games a launch is latency bound).
  config2            BASELINE configs[1]: 4,096 Tri / Dve / Ena games, the headline mode and the step API.
  issue_roofline     what really bounds k_play_wide: instruction issue (the play role's
                     instruction count per step from the committed SQ counters x the measured
                     issue cost per instruction from tools/valu_issue.hip).
  cpu_baseline       the CPU oracle (oracle/, a C port of the reference rules — test
                     infrastructure, used here only as the reported baseline) on the host
                     cores, bounded sample of the same workload.  rank 0, N=1 only.
Objects read from profiles/ carry the hash of the kernel sources they were measured on and
`stale: true` when it differs from the running library's.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 54
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s HBM3E
PROFILE_TAG = "r04"             # profiles/<tag>_* are the files read below
LAUNCH_BOUNDARY_US = 1.45       # a dependent kernel boundary on MI355X (/opt/skills/guides/MI355X_MICROARCH.md)
SEEDS = (0, 1, 2)               # SURVEY 8d / BASELINE.md: the repeats rotate these seeds
STREAM_GAMES = 1 << 22          # the step API's streaming leg
SIDE_LOCK_STEPS_CAP = 7680      # side legs: at most this many lock-steps per timed region
MIN_WARMUP_LOCK_STEPS = 49152   # the headline's untimed warm-up lasts at least this long (see main)


def graph_size(passes, limit):
    """Launches per hipGraph for a region of `passes` launches, at most `limit`: the largest divisor of
    `passes` (no eager remainder) unless that makes the graphs short (< 8 launches); then `limit` with
    the remainder launched eagerly."""
    limit = max(1, min(limit, passes))
    best = max(d for d in range(1, limit + 1) if passes % d == 0)
    return best if best >= min(8, limit) else limit


def plan_region(passes, cards, graph_lock_steps=4096, launches_per_graph=None):
    """How `passes` launches of `cards` lock-steps each are enqueued (pure; tests/test_bench_plan.py).
    launches_per_graph: use this graph size (a warm-up that must capture the timed region's graph).

    cards: 0 = tarok_policy_random + tarok_step (two kernels per lock-step), 1 = tarok_step_random,
    >= 2 = tarok_krog_random(cards).  Returns the arguments of tarok_run_random and what it will do:
    replays of one hipGraph of `per_graph` launches, then `eager` single launches.  Never zero launches."""
    passes = max(1, int(passes))
    unit = max(1, int(cards))                         # lock-steps per launch
    per_graph = graph_size(passes, max(1, int(graph_lock_steps) // unit)) if graph_lock_steps > 0 else 0
    if launches_per_graph is not None and graph_lock_steps > 0:
        per_graph = int(launches_per_graph)
        passes = max(passes, per_graph)                 # at least one whole graph
    replays = passes // per_graph if per_graph else 0
    eager = passes - replays * per_graph
    return {"cards": int(cards), "lock_steps_per_launch": unit, "launches": passes, "lock_steps": passes * unit,
            "graph_chunk": per_graph * unit, "launches_per_graph": per_graph, "graph_replays": replays,
            "eager_launches": eager, "kernels_per_launch": 2 if cards == 0 else 1}


def describe_mode(plan):
    c = plan["cards"]
    how = ("%d replay(s) of a hipGraph of %d launch(es)" % (plan["graph_replays"], plan["launches_per_graph"])
           if plan["graph_replays"] else "eager launches")
    if plan["eager_launches"]:
        how += " + %d eager launch(es)" % plan["eager_launches"]
    if c >= 2:
        from tarok_amd import karte
        return ("tarok_krog_random: %d card(s) of every game per kernel launch (4 = one trick, the reference's krog); "
                "action, observation, done, scores written to HBM for every card; %s; finished games' successors (dealt "
                "%d games ahead by the refill workgroups of the previous launch) are swapped in inside the same launch"
                % (c, how, karte.GAMES_AHEAD))
    if c == 1:
        return "tarok_step_random: one card of every game per launch, policy in-kernel; " + how
    return "tarok_policy_random + tarok_step: two launches per lock-step (what an external policy drives); " + how


def baseline_metric():
    """BASELINE.json's metric string, verbatim (it travels with the repo); the same words if the file is missing."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "env steps/sec at 65 536 parallel 4-player games; 1/2/4/8 MI355X"


def kernel_src_sha():
    """Hash of the sources libtarokenv.so is built from: what a profile was measured on."""
    from tarok_amd import _native
    h = hashlib.sha256()
    for p in _native.DEPS:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_profile(name, sha):
    """profiles/<name> -> (dict or None, provenance dict with `stale`)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, {"source": None}
    with open(path) as f:
        obj = json.load(f)
    meas = obj.get("kernel_src_sha")
    return obj, {"source": "profiles/" + name, "measured_on_kernel_src_sha": meas, "running_kernel_src_sha": sha,
                 "stale": meas != sha}


def cpu_baseline(n_games_chunk, mix, min_seconds=10.0, max_seconds=25.0):
    """Time the CPU oracle on the same synthetic workload (random-policy rollouts
    of mixed-contract games), all host cores."""
    from oracle import oracle as O
    cores = max(1, min(os.cpu_count() or 1, 64))
    try:
        cores = max(1, min(cores, len(os.sched_getaffinity(0))))
    except AttributeError:
        pass
    O.rollout(0, 0, 4096, 0, mix, threads=cores, trace=False)          # warm
    t0 = time.perf_counter()
    steps, games, ep = 0, 0, 0
    while True:
        r = O.rollout(0, 0, n_games_chunk, ep, mix, threads=cores, trace=False)
        steps += r["total_steps"]
        games += n_games_chunk
        ep += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or dt >= max_seconds:
            break
    t1 = time.perf_counter()
    r1 = O.rollout(0, 0, n_games_chunk // 8, 0, mix, threads=1, trace=False)
    dt1 = time.perf_counter() - t1
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": steps / dt, "unit": "env steps/s", "cores": cores, "cpu_model": model, "kind": "port",
            "sample": "%d random-policy games (%d steps) of the mixed-contract workload, C oracle, %d threads, %.1f s"
                      % (games, steps, cores, dt),
            "single_core_value": r1["total_steps"] / dt1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed passes of the hot path = kernel launches of --cards-per-launch lock-steps each "
                         "(default 100 x 128 = 12,800 lock-steps, ~350 games per slot)")
    ap.add_argument("--warmup", type=int, default=10, help="untimed passes (raised to one whole graph)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps passes in all, reported as a spread")
    ap.add_argument("--games", type=int, default=65536, help="games per GPU")
    ap.add_argument("--graph-chunk", type=int, default=4096, help="at most this many lock-steps per replayed hipGraph (0 = eager)")
    ap.add_argument("--cards-per-launch", type=int, default=128,
                    help="headline mode: cards of every game per launch (4 = one trick = one pass of the reference's krog; "
                         "48 = twelve tricks = the longest game; 128 = thirty-two tricks, where the time between launches "
                         "stops mattering at 65,536 games; 1 = one card per launch), at most 192")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements")
    ap.add_argument("--strict", action="store_true", help="exit 1 (after printing the line) when a side leg failed")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or not (1 <= args.cards_per_launch <= 192):
        ap.error("--steps >= 1, --warmup >= 0, 1 <= --cards-per-launch <= 192")

    import torch
    from tarok_amd import TarokVecEnv, karte as K, sharding

    import tarok_amd
    tarok_amd.build()                       # no-op when libtarokenv.so is current (hipcc, gfx950)
    rank, local_rank, world_size = sharding.world()
    if os.environ.get("TAROK_BENCH_ONE_GPU"):   # rehearsal: all ranks share GPU 0, gloo for the barrier
        local_rank = 0
    if world_size > 1:
        torch.cuda.set_device(local_rank)
        sharding.init_process_group("gloo" if os.environ.get("TAROK_BENCH_ONE_GPU") else "nccl")
    if args.gpus != world_size and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world_size), file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    n = args.games
    offset, _ = sharding.weak_shard(n, rank)
    env = TarokVecEnv(n, device=local_rank, seed=SEEDS[0], mix=K.MIX_ALL, game_offset=offset)
    sha = kernel_src_sha()
    errors = []

    def run(plan, e=None):
        # (the one-card modes leave the done row out: bit 62 of the observation word says "finished by this step")
        kw = {"done_rows": False} if plan["cards"] <= 1 else {}
        (e or env).run_random(plan["lock_steps"], cards_per_launch=plan["cards"], graph_chunk=plan["graph_chunk"], auto_reset=True, **kw)

    def timed(plan, events=None, e=None):
        """barrier + synchronize, the plan's launches, synchronize, [clock stops], barrier; MAX over ranks of the wall time.
        The clock stops when THIS rank's launches have drained, before the trailing barrier: the path has no collective,
        and a barrier of the process group (tens to hundreds of microseconds) inside a region of a millisecond would be
        read as bad scaling (round 3's two-rank rehearsal: wall 2.65 ms against 1.10 ms of HIP events).  The slowest
        rank's time is the job's: MAX over ranks.  events: (ev0, ev1, stream) recorded on the launch stream around the
        same launches."""
        sharding.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if events:
            events[0].record(events[2])
        run(plan, e)
        if events:
            events[1].record(events[2])
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        sharding.barrier()
        timed.last_rank_wall = dt
        return sharding.max_over_ranks([dt])[0]

    def leg(cards, passes, e=None):
        """A side leg: reset, warm up (graph capture untimed), one timed region.  -> (plan, seconds)"""
        chunk = min(args.graph_chunk, 1536)          # (graphs of at most 1,536 small launches)
        plan = plan_region(passes, cards, chunk)
        (e or env).reset(episode=0)
        run(plan_region(min(passes, 2 * plan["launches_per_graph"]), cards, chunk, plan["launches_per_graph"]), e)
        return plan, timed(plan, None, e)

    # ---- headline: --steps launches of tarok_krog_random(cards); per card: legal mask -> uniform random
    # legal card -> apply -> (4th card) trick winner / scoring / auto-reset swap -> next observation
    cards = args.cards_per_launch
    plan = plan_region(args.steps, cards, args.graph_chunk)
    # the untimed warm-up is never shorter than MIN_WARMUP_LOCK_STEPS lock-steps (~25 ms): a fresh process finds the
    # GPU at idle clocks, and a timed region right after a 1 ms warm-up reads 6 % low (measured: 138.7 G after 20
    # launches, 147.0 G after 400, same box, same 20 timed launches)
    wplan = plan_region(max(args.warmup, -(-MIN_WARMUP_LOCK_STEPS // max(1, cards))), cards, args.graph_chunk,
                        plan["launches_per_graph"])
    env.reset(episode=0)
    run(wplan)                                # the graph is captured here, untimed
    stream = torch.cuda.current_stream(dev)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), stream)
    ev[0].record(stream)                      # (torch creates the HIP events on their first record: not inside the timed region)
    ev[1].record(stream)
    dt = timed(plan, ev)
    ev_ms = ev[0].elapsed_time(ev[1])
    wall_per_rank = sharding.gather_over_ranks(timed.last_rank_wall)
    ev_per_rank = sharding.gather_over_ranks(ev_ms)
    pg = sharding.group_info()
    env_steps = n * plan["lock_steps"] * world_size
    value = env_steps / dt
    # 5 repeats over the seeds {0, 1, 2}, median (min-max).  `value` stays the first region (seed 0: the
    # contract's one); every further seed gets its own env, warmed up and graph-captured like the first
    seed_envs = {SEEDS[0]: env}
    spread, per_seed = [value], {str(SEEDS[0]): [value]}
    for k in range(1, max(1, args.repeats)):
        sd = SEEDS[k % len(SEEDS)]
        if sd not in seed_envs:
            e2 = TarokVecEnv(n, device=local_rank, seed=sd, mix=K.MIX_ALL, game_offset=offset)
            e2.reset(episode=0)
            run(wplan, e2)
            seed_envs[sd] = e2
        v = env_steps / timed(plan, None, seed_envs[sd])
        spread.append(v)
        per_seed.setdefault(str(sd), []).append(v)
    for sd, e2 in seed_envs.items():
        if e2 is not env:
            e2.close()
    ep, ss = env.counters()

    out = {
        "metric": baseline_metric(),
        "value": value, "unit": "env steps/s", "n_gpus": world_size, "steps": plan["launches"], "warmup": wplan["launches"],
        "steps_requested": args.steps, "warmup_requested": args.warmup,
        "ms_per_step": dt / plan["launches"] * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "step_definition": "one pass of the hot path over the batch = one kernel launch = %d lock-step(s) (cards) of each of "
                           "the %d games of a rank = %d env steps per rank" % (plan["lock_steps_per_launch"], n,
                                                                            n * plan["lock_steps_per_launch"]),
        "lock_steps_timed": plan["lock_steps"], "us_per_lock_step": dt / plan["lock_steps"] * 1e6,
        "timed_region": {"wall_ms_max_over_ranks": dt * 1e3, "hip_event_ms_rank0": ev_ms,
                         "wall_ms_per_rank": [w * 1e3 for w in wall_per_rank], "hip_event_ms_per_rank": ev_per_rank,
                         "process_group": pg, "rccl_ranks_seen": pg["world_size"] if pg["backend"] == "nccl" else 0,
                         "note": "barrier + synchronize before the region; every rank's clock stops when its own launches "
                                 "have drained (synchronize), a barrier follows OUTSIDE the clock; value uses the MAX over "
                                 "ranks of that wall time; the HIP events bracket the same launches on the launch stream"},
        "config": {"workload": "configs[2]: %d parallel envs per GPU, mixed Klop/Berac/Navadna contracts "
                               "(1/3 Klop, 1/3 Berac incl. 1/2 open, 1/3 Navadna+Solo over 7 types), uniform random policy, "
                               "auto-reset (every slot live in every step)" % n,
                   "games_per_gpu": n, "mode": describe_mode(plan), "cards_per_launch": cards, "launch_plan": plan,
                   "parallelism": "games sharded %d-way by global game index, no collective in the env path" % world_size},
        "episodes_finished_rank0": int(ep.sum()),
        "repeats": {"n": len(spread), "median": sorted(spread)[len(spread) // 2], "min": min(spread), "max": max(spread),
                    "seeds": [SEEDS[k % len(SEEDS)] for k in range(len(spread))], "per_seed": per_seed,
                    "note": "region k runs seed k mod 3 (its own env and graph); `value` is region 0"},
        "kernel_src_sha": sha,
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (k_play_wide): HIP events on the launch stream around the
        # timed region; launch duration = region time / launches (every launch gap is charged to the
        # kernel -> a lower bound on its bandwidth).
        k_us = ev_ms * 1e3 / plan["launches"]
        algo_bytes = ALGO_BYTES_PER_STEP * n * plan["lock_steps_per_launch"]
        algo = {"bytes_per_step": ALGO_BYTES_PER_STEP, "bytes_per_launch": algo_bytes,
                "achieved": algo_bytes / (k_us * 1e-6) / 1e9, "unit": "GB/s"}
        algo["frac"] = algo["achieved"] / HBM_PEAK_GBS
        # HBM-side bytes per launch: rocprofv3 PMC passes of this same command (counters cannot be read
        # from inside the process): FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_summary.py
        pmc, prov = load_profile("%s_pmc_fetch_write_%d.json" % (PROFILE_TAG, n), sha)
        traffic = pmc.get("k_play_traffic_bytes_per_launch_cards%d" % cards) if pmc else None
        roof = {"bound": "instruction issue (see issue_roofline); the HBM side is reported here: achieved / peak = hbm_frac" if cards > 1 else "hbm",
                "kernel": "k_play_wide (tarok_krog_random)" if cards > 1 else "k_step<true, .> (tarok_step_random)",
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "launch_us": k_us, "steps_per_launch": n * plan["lock_steps_per_launch"],
                "traffic": traffic, "traffic_provenance": prov, "algorithmic": algo}
        if cards > 1 and traffic:
            roof.update(accounting="measured HBM bytes (PMC FETCH_SIZE x2 + WRITE_SIZE per launch)",
                        achieved=traffic / (k_us * 1e-6) / 1e9)
            algo["note"] = ("SURVEY 8d's 54 B/step prices a kernel that streams the state once per card; a %d-card launch keeps it "
                            "in registers, so this figure is NOT a bound for it (it exceeds 1 at 16 M games)" % cards)
        else:
            roof.update(accounting="algorithmic 54 B/step (SURVEY 8d)" + ("" if cards == 1 else
                                   "; no PMC profile of this launch shape is committed, see `algorithmic.note`"),
                        achieved=algo["achieved"])
            if cards > 1:
                algo["note"] = "the state stays in registers for the cards of a launch: real traffic is lower than this figure"
        roof["frac"] = roof["hbm_frac"] = roof["achieved"] / HBM_PEAK_GBS
        roof["note"] = ("at %d games the per-GPU state (%.0f MB with the next-game lines) is cache resident and there are only "
                        "%d play waves for 1,024 SIMDs: the launch is bound by instruction issue, not by HBM (issue_roofline; "
                        "DESIGN.md has the N sweep)" % (n, n * 520 / 1e6, (n + 63) // 64))
        out["roofline"] = roof

        # ---- what actually bounds the kernel: instruction issue.  A wave that is alone on its SIMD
        # issues one instruction of ANY kind per ~4.6 cycles (tools/valu_issue.hip -> profiles/<tag>_valu_issue.json);
        # at 65,536 games every SIMD holds one play wave, and the refill waves that share the SIMDs for part of the
        # launch do not slow it (tools/first_launches.py: a launch without refill work has the same play-wave time).
        # So the one-wave ceiling is priced on the PLAY role's instructions alone: SQ counters of launches whose
        # refill workgroups have nothing to do (tools/play_only_counters.sh).  With two or more waves per SIMD the
        # SIMD's own time per VALU instruction (full rate ~2.3, half rate ~4.2 cycles) over ALL instructions of a
        # launch (tools/sq_counters.sh) is the limit.
        sq, sq_prov = load_profile("%s_sq_counters.json" % PROFILE_TAG, sha)
        po, po_prov = load_profile("%s_play_only_counters.json" % PROFILE_TAG, sha)
        vi, vi_prov = load_profile("%s_valu_issue.json" % PROFILE_TAG, sha)
        kinds = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH")
        if sq and po and vi and all(p.get("cards_per_launch") == cards and p.get("games") == n and "SQ_INSTS_VALU" in p for p in (sq, po)):
            wave_steps = n / 64.0 * cards
            valu_all = sq["SQ_INSTS_VALU"]["mean"] / wave_steps
            every_all = sum(sq[k]["mean"] for k in kinds if k in sq) / wave_steps
            every_play = sum(po[k]["mean"] for k in kinds if k in po) / wave_steps
            clock = vi.get("clock_hz", 2.4e9)
            simd_ceiling = 1024 * clock * 64.0 / (valu_all * vi["k_play_mix_cycles_per_valu"])
            lone_ceiling = 1024 * clock * 64.0 / (every_play * vi["lone_wave_cycles_per_instruction"])
            lone = (n + 63) // 64 <= 1024
            ceiling = lone_ceiling if lone else simd_ceiling
            iss = {
                "bound": "instruction issue, one wave per SIMD" if lone else "SIMD time of the VALU instructions",
                "play_role_instructions_per_step": every_play, "all_roles_instructions_per_step": every_all,
                "all_roles_valu_instructions_per_step": valu_all,
                "lone_wave_cycles_per_instruction": vi["lone_wave_cycles_per_instruction"],
                "simd_cycles_per_valu_instruction_k_play_mix": vi["k_play_mix_cycles_per_valu"], "clock_hz": clock,
                "ceiling_one_wave_per_simd": lone_ceiling, "ceiling_simd_throughput": simd_ceiling,
                "ceiling_steps_per_s_per_gpu": ceiling, "frac": value / world_size / ceiling,
                "provenance": {"counters_all_roles": sq_prov, "counters_play_role": po_prov, "issue_costs": vi_prov},
                "note": "instructions per step = SQ counters of a launch / (games / 64 x cards); ceiling = 1024 SIMDs x clock x "
                        "64 lanes / (instructions per step x cycles per instruction).  The play wave falls short of its one-wave "
                        "ceiling by the stalls the microbenchmark prices: a select through a compare costs 16.6 cycles, on a "
                        "compound condition 28.5, a taken branch ~25, a vote-and-branch ~50 (profiles/<tag>_valu_issue.json)"}
            if "SQ_WAVE_CYCLES" in po and "SQ_WAIT_ANY" in po:
                iss["play_role_wave_time_waiting_frac"] = po["SQ_WAIT_ANY"]["mean"] / po["SQ_WAVE_CYCLES"]["mean"]
            out["issue_roofline"] = iss

    if not args.no_extras:
        # ---- side measurements (not `value`); lock-steps per region = the headline's, capped
        side_steps = min(plan["lock_steps"], SIDE_LOCK_STEPS_CAP)
        # (a) the C-ABI surface an external policy drives: tarok_policy_random writes the action array,
        # tarok_step consumes it — one card per launch, state through HBM every card: 54 B/step is the
        # right accounting here
        p0, dta = leg(0, side_steps)
        s0 = n * p0["lock_steps"] * world_size
        us0 = dta / p0["lock_steps"] * 1e6
        out["api_two_kernel"] = {"value": s0 / dta, "unit": "env steps/s", "us_per_lock_step": us0, "launch_plan": p0,
                                 "note": "tarok_policy_random + tarok_step per lock-step (2 launches): what an external policy drives"}
        if rank == 0:
            ach = ALGO_BYTES_PER_STEP * n / (us0 * 1e-6) / 1e9
            out["roofline_step_api"] = {"bound": "hbm", "kernel": "k_step<false, true> (tarok_step; at this size the instantiation without refill workgroups: every step workgroup works its own lists off, bulk deals every 32nd launch) behind k_policy (tarok_policy_random)",
                                        "accounting": "algorithmic 54 B/step (SURVEY 8d) x %d games per lock-step" % n,
                                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                        "us_per_lock_step": us0, "traffic": None,
                                        # what actually bounds a lock-step at this size: two dependent launches
                                        "latency_floor": {"dependent_boundaries_per_lock_step": 2, "us_per_boundary": LAUNCH_BOUNDARY_US,
                                                          "floor_us": 2 * LAUNCH_BOUNDARY_US, "frac": 2 * LAUNCH_BOUNDARY_US / us0,
                                                          "note": "MI355X_MICROARCH.md: ~1.45 us per dependent kernel boundary; a lock-step "
                                                                  "of the external-policy path is policy launch -> step launch, each "
                                                                  "waiting for the other's output: frac = floor / measured"},
                                        "note": "wall time of the timed region / lock-steps: both launches and their gaps are "
                                                "charged to the step; at %d games (state cache resident, one wave per SIMD) a "
                                                "launch is latency bound — see profiles/ for the N sweep of this path" % n}
            # measured HBM bytes of the two kernels of a lock-step (PMC passes of the all-modes command)
            pma, pma_prov = load_profile("%s_pmc_fetch_write_all_modes_%d.json" % (PROFILE_TAG, n), sha)
            if pma and "k_step_traffic_bytes_per_launch" in pma:
                tr = pma["k_step_traffic_bytes_per_launch"] + pma.get("k_policy_traffic_bytes_per_launch", 0)
                out["roofline_step_api"].update(traffic=tr, traffic_over_algorithmic=tr / float(ALGO_BYTES_PER_STEP * n),
                                                step_kernel_traffic=pma["k_step_traffic_bytes_per_launch"],
                                                step_kernel_traffic_over_algorithmic=pma["k_step_traffic_bytes_per_launch"] / float(ALGO_BYTES_PER_STEP * n),
                                                traffic_note="traffic = k_step<false> + k_policy (the stand-in policy reads the observation word "
                                                             "and the game's RNG key: 17 B/step that are the policy's, not the env's)",
                                                traffic_provenance=pma_prov)
        # (a0) the same path where it streams: 4 M games (the state no longer fits the caches; 54 B/step against 8 TB/s
        # is a meaningful fraction here).  Its own env; skipped when the batch is not the default (N sweeps).
        if n == 65536 and world_size == 1:
            try:
                es = TarokVecEnv(STREAM_GAMES, device=local_rank, seed=SEEDS[0], mix=K.MIX_ALL)
                stream = {}
                for c_, name in ((0, "policy_plus_step"), (1, "step_random")):
                    ps_, dts = leg(c_, 384, es)
                    us = dts / ps_["lock_steps"] * 1e6
                    ach = ALGO_BYTES_PER_STEP * STREAM_GAMES / (us * 1e-6) / 1e9
                    stream[name] = {"us_per_lock_step": us, "env_steps_per_s": STREAM_GAMES / (us * 1e-6),
                                    "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
                es.close()
                stream["games"] = STREAM_GAMES
                stream["accounting"] = "algorithmic 54 B/step (SURVEY 8d) x %d games per lock-step, against 8 TB/s" % STREAM_GAMES
                led, led_prov = load_profile("%s_step_ledger.json" % PROFILE_TAG, sha)
                if led:
                    stream["traffic_bytes_per_step"] = led.get("two_kernel_bytes_per_step")
                    stream["traffic_over_algorithmic"] = led.get("two_kernel_bytes_per_step", 0) / float(ALGO_BYTES_PER_STEP)
                    stream["step_kernel_traffic_over_algorithmic"] = led.get("step_kernel_bytes_per_step", 0) / float(ALGO_BYTES_PER_STEP)
                    stream["step_random_traffic_over_algorithmic"] = (led.get("step_random_bytes_per_step") or 0) / float(ALGO_BYTES_PER_STEP)
                    if led.get("two_kernel_bytes_per_step_calibrated"):
                        # FETCH_SIZE doubled for the streamed reads only, not for the scattered record reads of the games that
                        # end (profiles/<tag>_fetch_calibration.txt); the figures above double it everywhere, as the guide says
                        stream["traffic_over_algorithmic_calibrated"] = led["two_kernel_bytes_per_step_calibrated"] / float(ALGO_BYTES_PER_STEP)
                        stream["step_random_traffic_over_algorithmic_calibrated"] = (led.get("step_random_bytes_per_step_calibrated") or 0) / float(ALGO_BYTES_PER_STEP)
                    stream["traffic_provenance"] = led_prov
                if rank == 0:
                    out["roofline_step_api"]["streaming"] = stream
            except Exception as ex:
                errors.append("step_api_streaming: " + repr(ex))
                print("bench.py: the streaming step-API leg FAILED: %r" % (ex,), file=sys.stderr)
        # (a00) BASELINE configs[1]: 4,096 Tri / Dve / Ena games — the headline mode and the step API
        try:
            n2 = 4096
            e2 = TarokVecEnv(n2, device=local_rank, seed=SEEDS[0], mix=K.MIX_NAVADNA3, game_offset=sharding.weak_shard(n2, rank)[0])
            pa, dta2 = leg(cards, max(1, side_steps // max(1, cards)), e2)
            pb, dtb2 = leg(0, side_steps, e2)
            out["config2"] = {"workload": "configs[1]: %d parallel Tri / Dve / Ena games per GPU, uniform random policy, auto-reset" % n2,
                              "headline_mode": {"value": n2 * pa["lock_steps"] * world_size / dta2, "unit": "env steps/s",
                                                "us_per_lock_step": dta2 / pa["lock_steps"] * 1e6, "cards_per_launch": cards},
                              "step_api": {"value": n2 * pb["lock_steps"] * world_size / dtb2, "unit": "env steps/s",
                                           "us_per_lock_step": dtb2 / pb["lock_steps"] * 1e6},
                              "note": "64 play waves for 1,024 SIMDs: launch-latency bound in both modes; bit-exactness of this "
                                      "configuration is pinned by the reference's digest (test_fused_rollout_digests_match_reference)"}
            e2.close()
        except Exception as ex:
            errors.append("config2: " + repr(ex))
            print("bench.py: the config-2 leg FAILED: %r" % (ex,), file=sys.stderr)
        # (a') one trick per launch (tarok_krog_random, 4 cards)
        if cards != 4:
            p4, dt4 = leg(4, max(1, side_steps // 4))
            out["one_trick_per_launch"] = {"value": n * p4["lock_steps"] * world_size / dt4, "unit": "env steps/s",
                                           "us_per_lock_step": dt4 / p4["lock_steps"] * 1e6, "launch_plan": p4,
                                           "note": "tarok_krog_random with 4 cards: 1 launch per trick (one pass of the reference's krog)"}
        # (a'') one card per launch with the policy in-kernel (tarok_step_random)
        if cards != 1:
            p1, dt1 = leg(1, side_steps)
            out["one_card_per_launch"] = {"value": n * p1["lock_steps"] * world_size / dt1, "unit": "env steps/s",
                                          "us_per_lock_step": dt1 / p1["lock_steps"] * 1e6, "launch_plan": p1,
                                          "note": "tarok_step_random: 1 launch per lock-step"}
        # (b) whole games per launch, state in registers
        sharding.barrier()
        torch.cuda.synchronize(dev)
        r = env.rollout_random(episode=0)
        torch.cuda.synchronize(dev)
        reps = 20
        t0 = time.perf_counter()
        for e in range(reps):
            r = env.rollout_random(episode=1 + e)
        torch.cuda.synchronize(dev)
        dtr = time.perf_counter() - t0
        tmax = sharding.max_over_ranks([dtr])[0]
        cnt = sharding.sum_over_ranks([float(r["nsteps"].sum().item()) * reps])[0]
        out["fused_rollout"] = {"value": cnt / tmax, "unit": "env steps/s",
                                "games_per_s": n * reps * world_size / tmax,
                                "note": "tarok_rollout_random: whole games in registers, one launch per %d games; "
                                        "no per-step HBM state, so no HBM fraction is claimed for it" % n}

        # (c) BASELINE configs 4-5: self-play with a small bf16 MLP policy (features -> MLP -> masked sample
        # -> env step in one launch), PPO-style update, gradient all-reduce over RCCL when N > 1.
        # Build-owned (the reference has no PPO): reported, never part of `value`.  A failure here is
        # printed on stderr and listed in `side_leg_errors` (and fails the run under --strict); the same
        # configuration at this size is a -m gpu test (test_selfplay_65536_envs_vs_oracle_replay).
        try:
            from tarok_amd import selfplay
            env.reset(episode=0)
            sp = selfplay.SelfPlay(env, hidden=256, seed=0)
            for _ in range(3):                                        # graph capture, buffers, first launches, clocks: untimed
                sp.iterate(T=48, epochs=1, minibatches=8)             # (an iteration is 5-6 ms: the second and third still speed up)
            its = [sp.iterate(T=48, epochs=1, minibatches=8) for _ in range(7)]
            st = sorted(its, key=lambda q: q["rollout_s"] + q["update_s"])[3]     # the median iteration of seven
            tro = sharding.max_over_ranks([st["rollout_s"], st["update_s"]])
            out["selfplay_ppo"] = {"iteration_env_steps_per_s": n * 48 * world_size / (tro[0] + tro[1]),
                                   "rollout_env_steps_per_s": n * 48 * world_size / tro[0], "rollout_us_per_lock_step": tro[0] / 48 * 1e6,
                                   "update_ms": tro[1] * 1e3, "minibatches": 8, "allreduce_bytes_per_minibatch": st["allreduce_bytes"],
                                   "policy": "MLP 256-256-256-64 (54 card logits + value), bf16 MFMA", "loss": st["loss"],
                                   "learner": "fused (tarok_learn_*)" if getattr(sp, "fused_learner", False) else "torch",
                                   "iterations_ms": [(q["rollout_s"] + q["update_s"]) * 1e3 for q in its],
                                   "note": "iteration = one rollout of 48 lock-steps + one update over its %d samples (1 epoch, 8 "
                                           "minibatches): the end-to-end figure (the median of seven iterations after three untimed ones).  Rollout: per lock-step one tarok_policy_step launch "
                                           "(features -> MLP -> masked sample -> env step), graph replayed.  Update: returns kernel; per "
                                           "minibatch forward + loss + backward chain in one MFMA kernel, the three weight gradients as "
                                           "one split-K launch, one flat gradient all-reduce, clip + Adam in one launch" % (n * 48)}
            del sp
        except Exception as ex:
            out["selfplay_ppo"] = {"error": repr(ex)}
            errors.append("selfplay_ppo: " + repr(ex))
            print("bench.py: the self-play side leg FAILED: %r" % (ex,), file=sys.stderr)

    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(1 << 20, K.MIX_ALL)
    out["side_leg_errors"] = errors

    env.close()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world_size > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if errors and args.strict:
        sys.exit(1)


if __name__ == "__main__":
    main()
