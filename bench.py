#!/usr/bin/env python3
"""Headline benchmark: env steps/s at 65,536 parallel 4-player games per MI355X
(BASELINE.json metric; workload = configs[2]: mixed Klop/Berac/Navadna contracts,
uniform-random policy, synthetic deals).

    python bench.py --gpus 1 --steps 4800 --warmup 480
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one lock-step of every game = one card played in each of the 65,536
games of a rank (Tarok.py:48-56): legal mask of the seat to move, a uniform random
legal card (the Bot policy, Igralec.py:158-159), the card applied, trick resolution
and scoring, finished games replaced at once (auto-reset: every slot is live in
every step), next observation written.  One kernel launch plays TWELVE TRICKS (48 such
steps; one trick = one pass of the reference's krog generator) with the state held in
registers in between and every per-card output (action, observation word, done,
scores) written to HBM; state is resident in HBM between launches.  One trick per
launch (--cards-per-launch 4) is reported beside it, as are  value = games x steps x ranks / max-over-ranks time.
one card per launch and the two-kernel external-policy path.  Weak scaling: each rank owns its own 65,536 games
(global game indices rank*65536...), no collective in the env path.

Extra objects on the JSON line:
  roofline      the step kernel against HBM peak: algorithmic 54 B/step
                x 65,536 games per launch / launch duration, measured with HIP events
                on the launch stream around the timed region.
  cpu_baseline  the CPU oracle (oracle/, a C port of the reference rules — test
                infrastructure, used here only as the reported baseline) on the host
                cores, bounded sample of the same workload.  rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 54
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s HBM3E


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
    ap.add_argument("--steps", type=int, default=9600, help="timed lock-steps (default ~260 games per slot)")
    ap.add_argument("--warmup", type=int, default=960, help="untimed lock-steps (default ~26 games per slot)")
    ap.add_argument("--repeats", type=int, default=5, help="extra timed regions of --steps steps, reported as a spread")
    ap.add_argument("--games", type=int, default=65536, help="games per GPU")
    ap.add_argument("--graph-chunk", type=int, default=192, help="steps per replayed hipGraph (0 = eager)")
    ap.add_argument("--prefetch-every", type=int, default=0,
                    help="extra synchronous tarok_prefetch every k steps (0: none; the step launches refill the buffers themselves)")
    ap.add_argument("--cards-per-launch", type=int, default=48,
                    help="headline mode: cards of every game per launch (4 = one trick = one pass of the reference's krog; "
                         "48 = twelve tricks; 1 = one card per launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fused-kernel side measurements")
    args = ap.parse_args()

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
    env = TarokVecEnv(n, device=local_rank, seed=0, mix=K.MIX_ALL, game_offset=offset)

    cards = max(1, args.cards_per_launch)
    pf = max(0, args.prefetch_every)
    pf = (pf + cards - 1) // cards * cards          # a prefetch period is a whole number of launches
    # a graph chunk never longer than the timed region, holding an even number of launches
    q = max(pf, 2 * cards)
    chunk = min(args.graph_chunk, (args.steps // q) * q) // q * q if args.graph_chunk > 0 else 0

    def run(steps, mode):
        """mode: 0 = policy kernel + step kernel, 1 = one card per launch, >= 2 = that many cards per launch"""
        unit = max(1, mode)
        main = steps // unit * unit
        if main:
            env.run_random(main, cards_per_launch=mode, graph_chunk=chunk, auto_reset=True, prefetch_every=pf)
        if steps - main:                                  # K not a multiple of the cards per launch: finish card by card
            env.run_random(steps - main, cards_per_launch=1, graph_chunk=0, auto_reset=True, prefetch_every=0)

    def timed(steps, mode):
        sharding.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(steps, mode)
        torch.cuda.synchronize(dev)
        sharding.barrier()
        dt = time.perf_counter() - t0
        return sharding.max_over_ranks([dt])[0]

    # ---- headline: one launch of tarok_krog_random per `cards` lock-steps: per card legal mask ->
    # uniform random legal card -> apply -> (4th card) trick winner / scoring / auto-reset swap ->
    # next observation; state read once and written once per launch, every per-card output
    # (action, observation word, done, scores) written to HBM.  Replayed as a hipGraph of
    # `graph_chunk` steps; tarok_prefetch every `prefetch_every` steps deals the finished slots'
    # next games.
    env.reset(episode=0)
    run(max(args.warmup, chunk), cards)          # at least one whole chunk: the graph is captured here, untimed
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream(dev)
    sharding.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record(stream)
    run(args.steps, cards)
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    sharding.barrier()
    dt_local = time.perf_counter() - t0
    dt = sharding.max_over_ranks([dt_local])[0]
    ev_ms = ev0.elapsed_time(ev1)
    total_steps = n * args.steps * world_size
    value = total_steps / dt
    # 5 repeats, median (min-max).  `value` stays the first region (the contract's one).
    spread = [value] + [total_steps / timed(args.steps, cards) for _ in range(max(0, args.repeats - 1))]
    ep, ss = env.counters()

    out = {
        "metric": "env steps/sec at 65,536 parallel 4-player games; 1/2/4/8 MI355X",
        "value": value, "unit": "env steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "configs[2]: %d parallel envs per GPU, mixed Klop/Berac/Navadna contracts "
                               "(1/3 Klop, 1/3 Berac incl. 1/2 open, 1/3 Navadna+Solo over 7 types), uniform random policy, "
                               "auto-reset (every slot live in every step)" % n,
                   "games_per_gpu": n,
                   "mode": "tarok_krog_random: %d card(s) of every game per kernel launch (4 = one trick, the reference's "
                           "krog); action, observation, done, scores written to HBM for every card; hipGraph of %d steps; "
                           "finished games' successors (dealt seven games ahead by the refill workgroups of the previous "
                           "launch) are swapped in inside the same launch" % (cards, chunk),
                   "cards_per_launch": cards,
                   "parallelism": "games sharded %d-way by global game index, no collective in the env path" % world_size},
        "episodes_finished_rank0": int(ep.sum()),
        "repeats": {"n": len(spread), "median": sorted(spread)[len(spread) // 2], "min": min(spread), "max": max(spread)},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (k_play<true>): HIP events on the launch stream around
        # the timed region above; launch duration = region time / launches (every launch gap is
        # charged to the kernel -> a lower bound on its bandwidth).  One launch processes n x cards
        # steps, each 54 algorithmic bytes.
        launches = args.steps // cards
        k_us = ev_ms * 1e3 / launches
        algo_bytes = ALGO_BYTES_PER_STEP * n * cards
        achieved = algo_bytes / (k_us * 1e-6) / 1e9
        # HBM-side bytes per launch come from the committed rocprofv3 PMC passes of this same
        # command (counters cannot be read from inside the process): FETCH_SIZE x2 + WRITE_SIZE
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_65536.json")
        if n == 65536 and os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get("k_play_traffic_bytes_per_launch_cards%d" % cards)
            traffic_src = "profiles/r01_pmc_fetch_write_65536.json"
        out["roofline"] = {"bound": "hbm", "kernel": "k_play<true> (%s)" % ("tarok_krog_random" if cards > 1 else "tarok_step_random"),
                           "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": algo_bytes, "launch_us": k_us, "steps_per_launch": n * cards,
                           "note": "54 B/step (SURVEY 8d) x %d games x %d cards per launch / (HIP-event time of the timed "
                                   "region / launches); at this N the per-GPU state (2 MB) is cache resident and there is one "
                                   "wave per SIMD: the launch is bound by that wave's instruction issue, see DESIGN.md for "
                                   "the N-sweep and the VALU-issue ceiling"
                                   % (n, cards)}

    if rank == 0:
        # ---- what actually bounds the kernel: the VALU instruction count per step,
        # from the committed SQ counter passes of this kernel (tools/sq_counters.sh: 4 M games, where
        # the vector ALUs are busy ~96 % of the launch).  Ceiling = SIMDs x clock / 4 cycles per
        # wave64 instruction x 64 lanes / instructions per step; reported beside the HBM roofline.
        sq = os.path.join(ROOT, "profiles", "r01_sq_counters_4194304.json")
        if os.path.exists(sq):
            with open(sq) as f:
                c = json.load(f)
            if c.get("cards_per_launch") == cards and "SQ_INSTS_VALU" in c:
                waves = c["games"] / 64.0 * c["cards_per_launch"]
                per_step = c["SQ_INSTS_VALU"]["mean"] / waves
                ceiling = 1024 * 2.3e9 / 4.0 * 64.0 / per_step
                out["issue_roofline"] = {"bound": "valu", "valu_instructions_per_step": per_step,
                                         "all_instructions_per_step": (c["SQ_INSTS_VALU"]["mean"] + c["SQ_INSTS_SALU"]["mean"] +
                                                                       c["SQ_INSTS_BRANCH"]["mean"] + c["SQ_INSTS_VMEM_WR"]["mean"] +
                                                                       c["SQ_INSTS_VMEM_RD"]["mean"]) / waves,
                                         "valu_busy_at_4M_games": c["SQ_ACTIVE_INST_VALU"]["mean"] * 4.0 / (1024 * c["GRBM_GUI_ACTIVE"]["mean"] / 8.0),
                                         "ceiling_steps_per_s_per_gpu": ceiling, "frac": value / world_size / ceiling,
                                         "source": "profiles/r01_sq_counters_4194304.json",
                                         "note": "1024 SIMDs x 2.3 GHz / 4 cycles x 64 lanes / VALU instructions per step"}

    if not args.no_extras:
        # ---- side measurements (not `value`)
        # (a) the two-kernel C-ABI path: tarok_policy_random writes the action array, tarok_step consumes it
        env.reset(episode=0)
        run(max(args.warmup, chunk), 0)
        dta = timed(args.steps, 0)
        out["api_two_kernel"] = {"value": total_steps / dta, "unit": "env steps/s", "ms_per_step": dta / args.steps * 1e3,
                                 "note": "tarok_policy_random + tarok_step per lock-step (2 launches): what an external policy drives"}
        # (a'') one trick per launch (tarok_krog_random, 4 cards)
        if cards != 4:
            env.reset(episode=0)
            run(max(args.warmup, chunk), 4)
            dt4 = timed(args.steps, 4)
            out["one_trick_per_launch"] = {"value": total_steps / dt4, "unit": "env steps/s", "ms_per_step": dt4 / args.steps * 1e3,
                                           "note": "tarok_krog_random with 4 cards: 1 launch per trick (one pass of the reference's krog)"}
        # (a') one card per launch with the policy in-kernel (tarok_step_random)
        env.reset(episode=0)
        run(max(args.warmup, chunk), 1)
        dt1 = timed(args.steps, 1)
        out["one_card_per_launch"] = {"value": total_steps / dt1, "unit": "env steps/s", "ms_per_step": dt1 / args.steps * 1e3,
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

        # (c) BASELINE configs 4-5: self-play with a small bf16 MLP policy (tarok_observe -> net ->
        # masked sample -> tarok_step), PPO-style update, gradient all-reduce over RCCL when N > 1.
        # Build-owned (the reference has no PPO): reported, never part of `value`.
        try:
            from tarok_amd import selfplay
            sp = selfplay.SelfPlay(env, hidden=256, seed=0)
            sp.iterate(T=48, epochs=1, minibatches=8)
            st = sp.iterate(T=48, epochs=1, minibatches=8)
            tro = sharding.max_over_ranks([st["rollout_s"], st["update_s"]])
            out["selfplay_ppo"] = {"rollout_env_steps_per_s": n * 48 * world_size / tro[0], "rollout_ms_per_step": tro[0] / 48 * 1e3,
                                   "update_ms": tro[1] * 1e3, "minibatches": 8, "allreduce_bytes_per_minibatch": st["allreduce_bytes"],
                                   "policy": "MLP 256-256-256-64 (54 card logits + value), bf16 MFMA", "loss": st["loss"],
                                   "note": "env steps/s including the policy: per lock-step one tarok_policy_step launch "
                                           "(features -> MLP -> masked sample -> env step), graph replayed; update = PPO-style, "
                                           "tarok_ppo_loss + torch GEMMs"}
            del sp
        except Exception as ex:                      # never let the side leg break the bench line
            out["selfplay_ppo"] = {"error": repr(ex)}

    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(1 << 20, K.MIX_ALL)

    env.close()
    if rank == 0:
        print(json.dumps(out))
    if world_size > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
