"""bench.py's step arithmetic and its JSON line, without a GPU.

Round 1's bench died on the driver's own command (`--steps 20 --warmup 5`: 20 // 48 = 0 launches,
ZeroDivisionError) — nothing on the CPU side touched that arithmetic.  Here: the pure launch plan
over every (steps, cards) pair, against a model of what tarok_run_random does with its arguments
(tarok_env.hip), and the whole of bench.main() on a fake env for the driver's exact flags."""
import json
import sys
import types

import pytest
import torch

import bench

STEPS = [1, 3, 4, 20, 47, 48, 95, 96, 200, 9600]
CARDS = [0, 1, 4, 5, 48, 64, 128]


def run_random_model(n_steps, cards, graph_chunk):
    """What tarok_run_random (tarok_env.hip) enqueues: (graph replays, eager launches) or EINVAL."""
    unit = cards if cards >= 2 else 1
    if n_steps < 0 or graph_chunk < 0 or graph_chunk > 8192 or not 0 <= cards <= 192:
        return "EINVAL"
    if n_steps % unit or graph_chunk % unit:
        return "EINVAL"
    left, replays = n_steps, 0
    if graph_chunk > 0 and left >= graph_chunk:
        replays = left // graph_chunk
        left -= replays * graph_chunk
    return replays, left // unit


@pytest.mark.parametrize("cards", CARDS)
@pytest.mark.parametrize("steps", STEPS)
def test_plan_never_times_zero_launches_and_matches_the_library(steps, cards):
    p = bench.plan_region(steps, cards, 4096)
    unit = max(1, cards)
    assert p["launches"] == steps >= 1                       # exactly the requested passes are timed
    assert p["lock_steps"] == steps * unit
    assert p["launches_per_graph"] >= 1 and p["graph_chunk"] == p["launches_per_graph"] * unit <= p["lock_steps"]
    assert p["graph_replays"] * p["launches_per_graph"] + p["eager_launches"] == p["launches"]
    assert run_random_model(p["lock_steps"], cards, p["graph_chunk"]) == (p["graph_replays"], p["eager_launches"])
    text = bench.describe_mode(p)
    if cards >= 2:
        assert "%d card(s)" % cards in text
    assert ("%d replay(s) of a hipGraph of %d launch(es)" % (p["graph_replays"], p["launches_per_graph"])) in text
    if p["eager_launches"]:
        assert "%d eager launch(es)" % p["eager_launches"] in text
    # a warm-up that captures the timed region's graph: same graph size, at least one whole graph
    for warm in (0, 1, 5, 960):
        w = bench.plan_region(warm, cards, 4096, p["launches_per_graph"])
        assert w["graph_chunk"] == p["graph_chunk"] and w["graph_replays"] >= 1 and w["launches"] >= max(warm, 1)
        assert run_random_model(w["lock_steps"], cards, w["graph_chunk"]) == (w["graph_replays"], w["eager_launches"])


def test_plan_eager_mode():
    p = bench.plan_region(20, 48, 0)
    assert p["graph_chunk"] == 0 and p["graph_replays"] == 0 and p["eager_launches"] == 20
    assert run_random_model(p["lock_steps"], 48, 0) == (0, 20)
    assert "eager launches" in bench.describe_mode(p)


class FakeEnv:
    """Stands in for TarokVecEnv: records what the bench enqueues."""
    instances = []

    def __init__(self, n, device=0, seed=0, mix=0, game_offset=0):
        self.n, self.calls, self.seed, self.mix = n, [], seed, mix
        FakeEnv.instances.append(self)

    def reset(self, episode=0):
        self.calls.append(("reset",))

    def run_random(self, n_steps, cards_per_launch=None, graph_chunk=0, auto_reset=True, done_rows=True):
        assert run_random_model(n_steps, cards_per_launch, graph_chunk) != "EINVAL"
        self.calls.append(("run", n_steps, cards_per_launch, graph_chunk))

    def counters(self):
        import numpy as np
        return np.zeros(self.n, np.int64), np.zeros((self.n, 4), np.int32)

    def rollout_random(self, episode=0):
        return {"nsteps": torch.full((self.n,), 40, dtype=torch.int16)}

    def close(self):
        pass


class FakeEvent:
    def __init__(self, enable_timing=False):
        pass

    def record(self, stream):
        pass

    def elapsed_time(self, other):
        return 0.8


@pytest.fixture
def fake_gpu(monkeypatch):
    import tarok_amd
    FakeEnv.instances.clear()
    monkeypatch.setattr(tarok_amd, "build", lambda *a, **k: None)
    monkeypatch.setattr(tarok_amd, "TarokVecEnv", FakeEnv, raising=False)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: object())
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    fake_sp = types.ModuleType("tarok_amd.selfplay")

    class SelfPlay:
        def __init__(self, *a, **k):
            raise RuntimeError("no GPU in this test")
    fake_sp.SelfPlay = SelfPlay
    monkeypatch.setitem(sys.modules, "tarok_amd.selfplay", fake_sp)
    monkeypatch.setattr(tarok_amd, "selfplay", fake_sp, raising=False)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TAROK_BENCH_ONE_GPU"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("argv", [["--gpus", "1", "--steps", "20", "--warmup", "5"],        # the driver's command
                                  ["--steps", "1", "--warmup", "0"], ["--steps", "47", "--cards-per-launch", "5"],
                                  ["--steps", "3", "--cards-per-launch", "1"], []])
def test_bench_line_on_a_fake_env(fake_gpu, monkeypatch, capsys, argv):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline"] + argv)
    bench.main()
    out = capsys.readouterr()
    line = json.loads(out.out.strip().splitlines()[-1])
    a = dict(zip(argv[::2], argv[1::2]))
    steps, cards = int(a.get("--steps", 100)), int(a.get("--cards-per-launch", 128))
    assert line["steps"] == line["steps_requested"] == steps and line["warmup"] >= line["warmup_requested"]
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["dtype"] == "u64" and line["n_gpus"] == 1
    assert line["lock_steps_timed"] == steps * cards
    cfg = line["config"]
    assert cfg["cards_per_launch"] == cards and cfg["launch_plan"]["launches"] == steps
    assert "workload" in cfg and "65536" in cfg["workload"]
    r = line["roofline"]
    assert r["launch_us"] == pytest.approx(800.0 / steps) and 0 < r["frac"] and r["algorithmic"]["bytes_per_step"] == 54
    assert r["steps_per_launch"] == 65536 * cards
    assert line["roofline_step_api"]["frac"] > 0 and line["api_two_kernel"]["value"] > 0
    # what the env was asked to do in the FIRST timed region is exactly the plan on the line:
    # reset, warm-up (untimed: holds the graph capture), then `repeats` identical regions
    env = FakeEnv.instances[0]
    p = cfg["launch_plan"]
    runs = [c for c in env.calls if c[0] == "run"]
    assert runs[0][2] == cards and runs[0][3] == p["graph_chunk"] and runs[0][1] >= max(p["graph_chunk"], bench.MIN_WARMUP_LOCK_STEPS)   # warm-up
    # five regions over the seeds 0, 1, 2, 0, 1: every seed on its own env, each warmed up untimed like the first
    region = ("run", p["lock_steps"], cards, p["graph_chunk"])
    assert env.seed == 0 and runs[1:3] == [region] * 2
    headline = [e for e in FakeEnv.instances if e.n == 65536 and e.mix == 0]
    assert [e.seed for e in headline] == [0, 1, 2]
    for e, regions in zip(headline[1:], (2, 1)):
        r2 = [c for c in e.calls if c[0] == "run"]
        assert r2[0] == runs[0] and r2[1:] == [region] * regions
    assert line["repeats"]["seeds"] == [0, 1, 2, 0, 1] and sorted(line["repeats"]["per_seed"]) == ["0", "1", "2"]
    assert "%d games ahead" % __import__("tarok_amd").karte.GAMES_AHEAD in bench.describe_mode(bench.plan_region(20, 128))
    assert line["roofline"]["hbm_frac"] == line["roofline"]["frac"] and ("issue" in line["roofline"]["bound"]) == (cards > 1)
    assert line["config2"]["headline_mode"]["value"] > 0 and line["config2"]["step_api"]["value"] > 0
    assert line["roofline_step_api"]["streaming"]["policy_plus_step"]["frac"] > 0
    # the failing self-play leg is visible: error field, side_leg_errors, stderr — and the line is still there
    assert "error" in line["selfplay_ppo"] and line["side_leg_errors"] and "FAILED" in out.err


def test_bench_strict_exits_nonzero_after_printing(fake_gpu, monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline", "--strict", "--steps", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 1
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])["side_leg_errors"]


def test_the_mode_string_names_the_library_constant():
    """bench.describe_mode quotes the dealt-ahead depth from tarok_amd.karte, which mirrors include/tarok_env.h,
    which is what the kernels are compiled with (TK_AHEAD): round 2's line said "seven" with fourteen lines."""
    import os, re
    from tarok_amd import karte
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "tarok_env.h")).read()
    src = open(os.path.join(root, "tarok_amd", "csrc", "tarok_env.hip")).read()
    assert int(re.search(r"#define TAROK_GAMES_AHEAD (\d+)", hdr).group(1)) == karte.GAMES_AHEAD
    assert re.search(r"#define TK_AHEAD TAROK_GAMES_AHEAD\b", src)
    assert int(re.search(r"#define TAROK_MLP_PARAMS (\d+)", hdr).group(1)) == karte.MLP_PARAMS
    assert "%d games ahead" % karte.GAMES_AHEAD in bench.describe_mode(bench.plan_region(20, 128))


def test_profile_provenance_is_flagged_stale(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "x.json").write_text(json.dumps({"kernel_src_sha": "abc", "v": 1}))
    obj, prov = bench.load_profile("x.json", "abc")
    assert obj["v"] == 1 and prov["stale"] is False
    obj, prov = bench.load_profile("x.json", "def")
    assert prov["stale"] is True and prov["measured_on_kernel_src_sha"] == "abc"
    assert bench.load_profile("missing.json", "abc") == (None, {"source": None})
    assert len(bench.kernel_src_sha()) == 16


def _two_rank_bench_worker(rank, world_size, port, q):
    """One gloo rank of `bench.py --gpus 2` on the fake env (spawned: the fakes are installed by hand)."""
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), TAROK_BENCH_ONE_GPU="1")
    import io, contextlib, time
    import tarok_amd
    FakeEnv.instances.clear()
    tarok_amd.build = lambda *a, **k: None
    tarok_amd.TarokVecEnv = FakeEnv
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: object()
    torch.cuda.Event = FakeEvent
    fake_sp = types.ModuleType("tarok_amd.selfplay")

    class SelfPlay:
        def __init__(self, *a, **k):
            raise RuntimeError("no GPU in this test")
    fake_sp.SelfPlay = SelfPlay
    sys.modules["tarok_amd.selfplay"] = fake_sp
    tarok_amd.selfplay = fake_sp
    # rank 1 is the slow one: its launches "take" 30 ms longer; the group's barrier costs what it costs
    real_run = FakeEnv.run_random

    def slow_run(self, *a, **k):
        real_run(self, *a, **k)
        if rank == 1:
            time.sleep(0.03)
    FakeEnv.run_random = slow_run
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--gpus", "2", "--steps", "20", "--warmup", "5"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    if rank == 0:
        q.put(buf.getvalue())


def test_two_rank_bench_line_times_each_rank_before_the_trailing_barrier():
    """`bench.py --gpus 2` as two gloo ranks on the fake env: ONE line from rank 0; value = all ranks' steps over the
    MAX over ranks of each rank's own wall time (clock stopped before the trailing barrier), per-rank wall and HIP-event
    times and the process group on the line."""
    import torch.multiprocessing as mp
    from tests.test_sharding_gloo import _free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    tr = line["timed_region"]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert tr["process_group"] == {"world_size": 2, "backend": "gloo"} and tr["rccl_ranks_seen"] == 0
    assert len(tr["wall_ms_per_rank"]) == 2 and len(tr["hip_event_ms_per_rank"]) == 2
    assert tr["wall_ms_per_rank"][1] >= 30.0 > tr["wall_ms_per_rank"][0]               # the slow rank's own time, not shared
    assert tr["wall_ms_max_over_ranks"] == pytest.approx(max(tr["wall_ms_per_rank"]))
    steps = 2 * 65536 * line["lock_steps_timed"]
    assert line["value"] == pytest.approx(steps / (tr["wall_ms_max_over_ranks"] * 1e-3))
