#!/usr/bin/env python3
"""Summarise tools/step_ledger.sh's PMC passes: bytes per env step of the step kernel and of k_policy, per variant,
split into the launches that play cards 0-2 of a trick and the ones that play the 4th card (every slot is
trick-aligned: dispatch j of the step kernel plays card j mod 4), then the ledger array by array as differences.
FETCH_SIZE is doubled (gfx950 counts a coalesced read stream at half: MI355X_MICROARCH.md, HBM section).
tools/fetch_calib (profiles/r03_fetch_calibration.txt) calibrated the counters on the step kernels' own access shapes:
streamed reads of 16 / 8 / 1 B per lane read 0.50 of their bytes (so x 2 is right for them), but the SPARSE record reads
of the lanes whose game ends (64-byte next-game lines, 32-byte Counters) read 1.00 / 1.79 of theirs — the counter already
holds the bytes fetched, doubling it counts them twice.  The "calibrated" figures take the reads that a 4th-card launch
has on top of a cards-0-2 launch (the sparse ones) once instead of twice."""
import csv, glob, os, sys, collections

def collect(d, counter):
    per = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                                       # (one pass per directory: the newest, if an older one was merged in)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    per[r["Kernel_Name"].split("(")[0]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(x)] for k, x in per.items()}

def main():
    out, n = sys.argv[1], int(sys.argv[2])
    variants = sorted({os.path.basename(p).rsplit("_", 2)[0] for p in glob.glob(os.path.join(out, "*_FETCH_SIZE")) if os.path.isdir(p)})
    table = {}
    for v in variants:
        fe, wr = collect(os.path.join(out, v + "_FETCH_SIZE"), "FETCH_SIZE"), collect(os.path.join(out, v + "_WRITE_SIZE"), "WRITE_SIZE")
        row = {}
        for name in sorted(set(fe) | set(wr)):
            is_policy = name.strip() in ("k_policy", "k_policy_x4")          # (tarok_policy_random's two forms; not k_policy_step / _mlp)
            if not ("k_step" in name or "k_play" in name or is_policy):
                continue
            short = "policy" if is_policy else "step"
            for cname, per, mul in (("R", fe, 2.0), ("W", wr, 1.0)):
                vals = per.get(name, [])[8:]                    # (skip the first two tricks after the reset)
                if not vals:
                    continue
                k0 = 8 % 4
                c012 = [x for j, x in enumerate(vals) if (j + k0) % 4 != 3]
                c3 = [x for j, x in enumerate(vals) if (j + k0) % 4 == 3]
                b = lambda xs: mul * 1024.0 * sum(xs) / len(xs) / n if xs else float("nan")
                row[short + cname] = (b(vals), b(c012), b(c3))
        table[v] = row
    def calibrated(r):
        """bytes per step with the 4th card's extra (sparse) reads counted once: (total, step kernel alone)"""
        extra = (r["stepR"][2] - r["stepR"][1]) / 2.0           # what the doubling added on a 4th-card launch
        tot = sum(r[k][0] for k in r) - extra / 4.0
        return tot, r["stepR"][0] + r["stepW"][0] - extra / 4.0
    print("step-API traffic ledger, %d games, MIX_ALL, auto-reset; bytes per env step (mean | cards 0-2 | 4th card)" % n)
    print("R = FETCH_SIZE x 2, W = WRITE_SIZE; step = the step kernel (k_step<..>, r02: k_play<false,true> / k_play_wide<true,false>)")
    for v in variants:
        r = table[v]
        cells = []
        for key in ("stepR", "stepW", "policyR", "policyW"):
            if key in r:
                cells.append("%s %6.1f | %6.1f | %6.1f" % (key, *r[key]))
        tot = sum(r[k][0] for k in r)
        cal = calibrated(r)[0] if "stepR" in r else tot
        print("%-14s %s   total %.1f B/step = %.2f x 54   calibrated %.1f = %.2f x" % (v, "   ".join(cells), tot, tot / 54.0, cal, cal / 54.0))
    if "two_base" in table and len(sys.argv) > 3:               # the JSON bench.py reads for roofline_step_api.streaming
        import json
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        r = table["two_base"]
        res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/step_ledger.py "
                         "%d two d 1 1 96 (tools/step_ledger.sh); FETCH_SIZE doubled" % n,
               "kernel_src_sha": bench.kernel_src_sha(), "games": n,
               "step_kernel_bytes_per_step": r["stepR"][0] + r["stepW"][0], "policy_kernel_bytes_per_step": r.get("policyR", (0,))[0] + r.get("policyW", (0,))[0],
               "two_kernel_bytes_per_step": sum(r[k][0] for k in r),
               "two_kernel_bytes_per_step_calibrated": calibrated(r)[0], "step_kernel_bytes_per_step_calibrated": calibrated(r)[1],
               "step_random_bytes_per_step_calibrated": calibrated(table["random_base"])[0] if "random_base" in table else None,
               "calibration": "FETCH_SIZE x 2 for the streamed reads, x 1 for the sparse record reads of ending games (profiles/r03_fetch_calibration.txt)",
               "step_random_bytes_per_step": sum(table["random_base"][k][0] for k in table.get("random_base", {})) or None,
               "by_card": {k: {"mean": r[k][0], "cards_0_2": r[k][1], "card_3": r[k][2]} for k in r}}
        with open(sys.argv[3], "w") as fh:
            json.dump(res, fh, indent=1)
    def d(a, b, key, col=0):
        try:
            return table[a][key][col] - table[b][key][col]
        except KeyError:
            return float("nan")
    print()
    print("array by array (differences between variants, bytes per env step):")
    print("  done row (1 B; two_base leaves it out, like bench.py's step-API legs): W %.2f" % d("two_done", "two_base", "stepW"))
    print("  reward rows of finished games:        W %.2f" % d("two_base", "two_noreward", "stepW"))
    if "two_spec1" in table:            # (round 3's first passes, while the speculative finish-path loads still existed)
        print("  speculative finish-path loads:        R %.2f   (SPEC on - off; 4th-card launches: %.2f)" % (d("two_spec1", "two_spec0", "stepR"), d("two_spec1", "two_spec0", "stepR", 2)))
    if "two_base" in table and "stepW" in table["two_base"]:
        r = table["two_base"]
        print("  4th card on top of cards 0-2:         R %.2f  W %.2f per 4th-card launch (seat pair 16 W; Counters, next-game line, key, list entry of the games that end)"
              % (r["stepR"][2] - r["stepR"][1], r["stepW"][2] - r["stepW"][1]))
        print("  cards 0-2 (play pair 16 R + 16 W, seat pair 16 R, card 1 R, observation 8 W = 33 R + 24 W): R %.2f  W %.2f" % (r["stepR"][1], r["stepW"][1]))
    if "two_r02" in table:
        print("  round-2 kernel (k_play<false,true>, 64 B/lane of scratch) minus this one:  R %.2f  W %.2f   (cards 0-2: R %.2f  W %.2f)"
              % (d("two_r02", "two_base", "stepR"), d("two_r02", "two_base", "stepW"), d("two_r02", "two_base", "stepR", 1), d("two_r02", "two_base", "stepW", 1)))

if __name__ == "__main__":
    main()
