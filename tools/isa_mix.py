#!/usr/bin/env python3
"""Diagnostic: per-kernel instruction mix of the gfx950 ISA (hipcc -S)."""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/tarok_isa.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", ROOT + "/include", "-S",
                       "--cuda-device-only", "-o", out, ROOT + "/tarok_amd/csrc/tarok_env.hip"], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
name, body = None, []
def report(name, body):
    ins = [l.strip().split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    tot = sum(c.values())
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    print(name[:30].ljust(32), "instr", tot, "| mul32", g(lambda k: "mul_lo" in k or "mul_hi" in k), "mad64", g(lambda k: "mad_u64" in k or "mad_i64" in k),
          "| min/max", c.get("v_min_u32", 0) + c.get("v_max_u32", 0), "| 64b shifts", g(lambda k: k.endswith("_b64") and "sh" in k),
          "| readlane", c.get("v_readlane_b32", 0), "| scratch", g(lambda k: "scratch" in k), "| global ld/st", g(lambda k: k.startswith("global_load")), g(lambda k: k.startswith("global_store")),
          "| branches", g(lambda k: k.startswith("s_cbranch")), "| bcnt", g(lambda k: "bcnt" in k))
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        name, body = m.group(1), []
    elif l.startswith(".Lfunc_end") and name:
        report(re.sub(r"^_Z\d+", "", name), body); name = None
    elif name is not None:
        body.append(l)
