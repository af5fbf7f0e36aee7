#!/usr/bin/env python3
"""Turn the raw output of tools/valu_issue (GPU box) into profiles/<tag>_valu_issue.json: the
measured issue cost of the instructions k_play is made of, and — with the static instruction mix of
k_play<true>'s trick-aligned card loop (hipcc -S, runs anywhere) — the weighted cost per VALU
instruction that bench.py's issue_roofline uses.

    python tools/valu_issue_summary.py gpurun_out/<dir>/valu_issue_raw.json profiles/r02_valu_issue.json

How the raw numbers are read (all in shader cycles per wave64 instruction):
  lone wave      per-wave stamps at one wave per SIMD: what ONE wave can issue, of any kind.
  SIMD cost      what the instruction occupies its SIMD for when >= 2 waves share it.  For ops whose
                 execution back-pressures issue (per-wave time doubles from 1 to 2 waves) this is the
                 per-wave time at 2 waves / 2; for the others (a wave still issues one per ~4.5 cycles
                 at any occupancy) the stamps only show the issue interval, and the SIMD cost comes from
                 the chip-level figure: wall time x clock x 1024 SIMDs / instructions, at 8 waves launched.
"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ISA mnemonic (suffix-stripped) -> microbenchmark row that prices it
PRICED_AS = {
    "v_and_b32": "and", "v_or_b32": "and", "v_xor_b32": "xor", "v_not_b32": "and", "v_add_u32": "add", "v_sub_u32": "add",
    "v_subrev_u32": "add", "v_add_co_u32": "add", "v_sub_co_u32": "add", "v_lshrrev_b32": "lshr", "v_lshlrev_b32": "lshr",
    "v_ashrrev_i32": "lshr", "v_mov_b32": "mov", "v_bitop3_b32": "bitop3", "v_or3_b32": "or3", "v_and_or_b32": "or3",
    "v_add3_u32": "add3", "v_lshl_or_b32": "lshl_or", "v_lshl_add_u32": "lshl_or", "v_add_lshl_u32": "lshl_or",
    "v_bcnt_u32_b32": "bcnt", "v_bfe_u32": "bfe", "v_bfe_i32": "bfe", "v_bfi_b32": "bfe", "v_alignbit_b32": "bfe",
    "v_min_u32": "min", "v_max_u32": "max", "v_min_i32": "min", "v_max_i32": "max", "v_mul_lo_u32": "mul_lo",
    "v_mul_hi_u32": "mul_hi", "v_mul_u32_u24": "mul_u24", "v_mad_u32_u24": "mad_u24", "v_cndmask_b32": "cndmask",
    "v_readlane_b32": "readlane", "v_readfirstlane_b32": "readlane", "v_lshlrev_b64": "lshl64v", "v_lshrrev_b64": "lshr64",
    "v_lshl_add_u64": "lshl_add64", "v_mov_b64": "mov64", "v_mad_u64_u32": "mad64", "v_addc_co_u32": "cndmask",
    "v_subb_co_u32": "cndmask", "v_ffbl_b32": "bcnt", "v_ffbh_u32": "bcnt", "v_lshrrev_b16": "sdwa", "v_lshlrev_b16": "sdwa",
    "v_sad_u8": "add3", "v_mbcnt_lo_u32_b32": "bcnt", "v_mbcnt_hi_u32_b32": "bcnt", "v_subbrev_co_u32": "cndmask",
}


def isa_mix():
    """Static VALU mix of the trick-aligned card loop of k_play<true> (the loop a rollout runs in)."""
    out = "/tmp/tarok_isa_mix.s"
    src = os.path.join(ROOT, "tarok_amd", "csrc", "tarok_env.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    L = open(out).read().split("\n")
    a = next(i for i, l in enumerate(L) if l.startswith("_Z11k_play_wideILb0EE"))  # k_play_wide<false> (no history): the kernel the Bot-policy launches use
    b = next(i for i in range(a, len(L)) if L[i].startswith(".Lfunc_end"))
    K = L[a:b]
    heads = [i for i, l in enumerate(K) if "Loop Header: Depth=1" in l and "Inner" not in l]

    def extent(h):
        name = re.match(r"\.(LBB\d+_\d+):", K[h]).group(1)[1:]
        return next(i for i in range(h + 1, len(K)) if K[i].startswith(".LBB") and ("Header=" + name) not in K[i] and ("Parent Loop " + name) not in K[i])
    # the play role's card loops follow the refill loop; the two trick-aligned ones are the last two, and the one
    # a rollout runs in (action and done rows given, no trick row) is the one with fewer stores
    cand = [(h, extent(h)) for h in heads[-2:]]
    h, e = min(cand, key=lambda he: sum(1 for l in K[he[0]:he[1]] if l.strip().startswith("global_store")))
    mix = collections.Counter()
    other = collections.Counter()
    for l in K[h:e]:
        if not l.startswith("\t") or l.strip().startswith((".", ";")):
            continue
        op = l.split()[0]
        if op.startswith("v_"):
            base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
            if op.endswith("_sdwa"):
                mix["sdwa"] += 1
            elif base.startswith("v_cmp"):
                mix["cmp"] += 1
            elif base in PRICED_AS:
                mix[PRICED_AS[base]] += 1
            else:
                mix["UNPRICED:" + base] += 1
        else:
            other["s_nop" if op == "s_nop" else ("salu" if op.startswith("s_") else "mem")] += 1
    return dict(mix), dict(other)


def main():
    raw_path, out_path = sys.argv[1], sys.argv[2]
    raw = json.load(open(raw_path))
    ops = raw["ops"]
    table = {}
    for name, v in ops.items():
        w = v["ind"]
        lone = w["w1"]["cyc"]
        backpressured = w["w2"]["cyc"] > 1.5 * lone          # execution, not issue, sets the pace at 2 waves
        simd = w["w2"]["cyc"] / 2.0 if backpressured else w["w8"]["chip"]
        table[name] = {"lone_wave": round(lone, 2), "simd_cost": round(simd, 2), "rate": "half" if simd > 3.2 else "full",
                       "per_wave_by_waves_per_simd": [round(w["w%d" % k]["cyc"], 2) for k in (1, 2, 4, 8)],
                       "dependent_chain_lone_wave": round(v["dep"]["w1"]["cyc"], 2),
                       "chip_level_by_waves_launched": [round(w["w%d" % k]["chip"], 2) for k in (1, 2, 4, 8)]}
    mix, other = isa_mix()
    priced = {k: n for k, n in mix.items() if not k.startswith("UNPRICED:")}
    unpriced = {k[9:]: n for k, n in mix.items() if k.startswith("UNPRICED:")}
    half_cost = table["bcnt"]["simd_cost"]
    tot = sum(priced.values()) + sum(unpriced.values())
    cyc = sum(n * table[k]["simd_cost"] for k, n in priced.items()) + sum(unpriced.values()) * half_cost
    lone = sorted(t["lone_wave"] for t in table.values())[len(table) // 2]
    sys.path.insert(0, ROOT)
    import bench                                          # (the same hash bench.py compares with: all of _native.DEPS)
    res = {
        "source": "tools/valu_issue (gfx950, %d iterations x %d instructions per wave; raw: %s) + static mix of k_play_wide's "
                  "trick-aligned card loop (hipcc -S)" % (raw["iters"], raw["instructions_per_iteration"], os.path.basename(raw_path)),
        "kernel_src_sha": bench.kernel_src_sha(),
        "clock_hz": raw["clock_mhz_mean"] * 1e6,
        "lone_wave_cycles_per_instruction": lone,
        "findings": [
            "ONE wave on a SIMD issues one instruction per ~%.1f cycles, whatever it is (VALU full or half rate, SALU, s_nop): "
            "at 65,536 games (1,024 play waves on 1,024 SIMDs) every instruction of a play wave costs that" % lone,
            "with >= 2 waves per SIMD a full-rate op (v_and/or/xor/add/sub/shift32/mov/bitop3/fma) occupies the SIMD ~%.1f cycles, "
            "a half-rate op (v_bcnt, v_bfe, v_min/max_u32, v_mul_lo/hi_u32, v_cndmask, v_cmp, v_readlane, 3-operand integer "
            "VOP3 such as v_or3/v_add3/v_lshl_or, SDWA forms, every 64-bit shift / move / add) ~%.1f" % (table["and"]["simd_cost"], half_cost),
            "a wave with 32 active lanes (low half or even lanes) issues no faster than a full one (half_waves): splitting "
            "the 65,536 games over twice as many half-filled waves buys no issue slots",
            "dependent chains of plain vector instructions cost the same as independent ones (no exposed ALU latency)",
        ] + ([
            "a per-lane select is where a lone wave loses time: through a compare (v_cmp, the two idle slots gfx950 wants before "
            "a VALU reads a mask a VALU wrote, v_cndmask) %.1f cycles, on a compound condition (v_cmp, s_and, v_cndmask) %.1f, "
            "as three plain vector instructions (difference, sign smear, v_bitop3) %.1f"
            % (3 * table["cmpsel"]["lone_wave"], 3 * table["select"]["lone_wave"], 3 * table["arithsel"]["lone_wave"]),
            "branches: s_cmp + taken branch %.1f cycles, s_cmp + branch not taken + the next instruction %.1f, and the "
            "`if (__ballot(c))` shape (v_cmp into an SGPR pair, s_cmp_lg_u64, branch taken) %.1f for its three instructions"
            % (2 * table["br_taken"]["lone_wave"], 3 * table["br_not"]["lone_wave"], 3 * table["ballot_br"]["lone_wave"]),
        ] if all(k in table for k in ("cmpsel", "select", "arithsel", "br_taken", "br_not", "ballot_br")) else []),
        "ops": table,
        "half_waves": raw["half_waves"],
        "k_play_static_mix": {"valu_by_priced_row": priced, "valu_unpriced_counted_as_half_rate": unpriced, "non_valu": other,
                              "valu_total": tot},
        "k_play_mix_cycles_per_valu": cyc / tot,
        "k_play_mix_half_rate_fraction": (sum(n for k, n in priced.items() if table[k]["rate"] == "half") + sum(unpriced.values())) / tot,
    }
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("ops", "half_waves")}, indent=1))


if __name__ == "__main__":
    main()
