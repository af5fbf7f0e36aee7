#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into the JSON that
bench.py reads for roofline.traffic.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <cards_per_launch> <games> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  On gfx950 FETCH_SIZE counts a wide
coalesced read stream at half its bytes (MI355X_MICROARCH.md, rocprofv3/HBM section; calibrated
here on k_legal, which reads 32 B per game): it is doubled; WRITE_SIZE is exact.
The output carries the hash of the kernel sources it was measured on (bench.kernel_src_sha)."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def collect(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    name = r["Kernel_Name"].split("(")[0]
                    per[name].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    return per

def main():
    import bench
    fetch_dir, write_dir, cards, games, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fe) | set(wr)):
        if not name.startswith(("k_", "void k_")):
            continue
        e = {}
        for cname, per in (("FETCH_SIZE", fe), ("WRITE_SIZE", wr)):
            rows = per.get(name, [])
            if rows:
                # the launches of the kernel's most frequent grid: the bench also runs a 4 M-game leg of the step API
                # (64 x the bytes per launch), which must not be averaged into the 65,536-game figure
                grid = collections.Counter(g for _, g in rows).most_common(1)[0][0]
                v = [x for x, g in rows if g == grid]
                e[cname] = {"launches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v), "grid_size": grid,
                            "launches_of_other_grids": len(rows) - len(v)}
        kernels[name] = e
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                     "--steps 20 --warmup 4 --repeats 2 --no-cpu-baseline (%d cards per launch in the headline leg, %d games); KB per "
                     "launch as reported; FETCH_SIZE doubled for bytes (gfx950: a wide coalesced read stream is counted at half)" % (cards, games),
           "kernel_src_sha": bench.kernel_src_sha(),
           "kernels": kernels, "games_per_launch": games, "cards_per_launch": cards}
    def traffic(k):
        f, w = kernels[k].get("FETCH_SIZE", {}).get("mean_KB"), kernels[k].get("WRITE_SIZE", {}).get("mean_KB")
        return None if f is None or w is None else int(round((2 * f + w) * 1024))
    # the multi-card Bot-policy kernel (k_play_wide<HIST>): the headline leg dominates its launches
    play = [k for k in kernels if "k_play_wide" in k]
    if play:
        k = max(play, key=lambda q: kernels[q].get("FETCH_SIZE", {}).get("launches", 0))
        res["k_play_kernel"] = k
        res["k_play_traffic_bytes_per_launch_cards%d" % cards] = traffic(k)
        res["bytes_per_step"] = traffic(k) / (games * cards)
        res["note"] = ("the side legs (one trick / one card per launch) launch the same kernel with fewer cards: the mean over its "
                       "launches is dominated by, but not purely, the %d-card launches; the headline-only passes are in *_headline*" % cards)
    # the step API's kernel (k_step<false>: one card per launch, external action array)
    # (k_step exists in two instantiations: the bench's 65,536-game launches run the one with the bulk deals, its 4 M-game
    #  streaming leg the other — the smaller grid is the one meant here)
    step = sorted((k for k in kernels if "k_step<false" in k.replace(" ", "")),
                  key=lambda k: min(c.get("grid_size", 1 << 62) for c in kernels[k].values()))
    if step:
        res["k_step_kernel"] = step[0]
        res["k_step_traffic_bytes_per_launch"] = traffic(step[0])
        pol = sorted((k for k in kernels if k.endswith(("k_policy", "k_policy_x4"))),
                     key=lambda k: min(c.get("grid_size", 1 << 62) for c in kernels[k].values()))
        if pol:
            res["k_policy_traffic_bytes_per_launch"] = traffic(pol[0])
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))

if __name__ == "__main__":
    main()
