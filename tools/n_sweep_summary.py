#!/usr/bin/env python3
"""profiles/<tag>_n_sweep.json from the bench lines of tools/profile_round.sh's N sweep.
    python tools/n_sweep_summary.py gpurun_out/<tag> profiles/r02_n_sweep.json"""
import json, os, sys
src, out = sys.argv[1], sys.argv[2]
rows = []
sha = None
for n in (65536, 1048576, 4194304, 16777216):
    p = os.path.join(src, "bench_N%d.json" % n)
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    sha = d["kernel_src_sha"]
    algo = d["roofline"]["algorithmic"]
    rows.append({"games": n, "value_env_steps_per_s": d["value"], "us_per_lock_step": d["us_per_lock_step"],
                 "launch_us": d["roofline"]["launch_us"], "cards_per_launch": d["config"]["cards_per_launch"], "steps": d["steps"],
                 "warmup": d["warmup"], "algorithmic_54B_GBps": algo["achieved"], "algorithmic_frac": algo["frac"], "repeats": d["repeats"]})
json.dump({"source": "python bench.py --games N --steps 8 --warmup 4 --no-cpu-baseline --no-extras (tools/profile_round.sh)",
           "kernel_src_sha": sha, "rows": rows}, open(out, "w"), indent=1)
for r in rows:
    print(r["games"], "%.1f G steps/s (median %.1f)" % (r["value_env_steps_per_s"] / 1e9, r["repeats"]["median"] / 1e9), "launch %.1f us" % r["launch_us"])
