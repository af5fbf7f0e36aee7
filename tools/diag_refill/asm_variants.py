#!/usr/bin/env python3
"""Round-4 diagnostic: libraries whose DEVICE code is the failing build's own assembly with one mechanical patch each
(tools/diag_refill/asm_build.sh), to tell an uninitialised-register read from a missing wait state from a code-generation
error in the refill loop of k_step.  Run HERE after gen_variants.py harness; writes tools/ab/asm_*.so.

    ctrl        the assembly unchanged (must fail like diag_harness.so)
    vzero       every VGPR but v0 zeroed at the entry of k_step<true>
    vones       ... set to all ones
    vlo / vhi   only v1..v51 / v52..v103 zeroed
    szero       every SGPR but s0..s2 zeroed at the entry
    nops        s_nop 3 behind every VALU instruction of k_step<true>
"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
WORK = "/tmp/diag_refill"
SRC = os.path.join(WORK, "diag_harness.hip")
ASM = os.path.join(WORK, "harness.s")
FUNC = "_Z6k_stepILb1EE"

def func_span(lines):
    a = next(i for i, l in enumerate(lines) if l.startswith(FUNC) and l.rstrip().split(";")[0].rstrip().endswith(":"))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return a, b

def patch(lines, kind):
    a, b = func_span(lines)
    out = list(lines)
    if kind == "ctrl":
        return out
    if kind in ("vzero", "vones", "vlo", "vhi"):
        lo, hi = {"vzero": (1, 103), "vones": (1, 103), "vlo": (1, 51), "vhi": (52, 103)}[kind]
        val = "-1" if kind == "vones" else "0"
        ins = ["\tv_mov_b32_e32 v%d, %s\n" % (r, val) for r in range(lo, hi + 1)]
        return out[:a + 1] + ins + out[a + 1:]
    if kind == "szero":
        ins = ["\ts_mov_b32 s%d, 0\n" % r for r in range(3, 92)]
        return out[:a + 1] + ins + out[a + 1:]
    if kind == "nops":
        res = out[:a + 1]
        for l in out[a + 1:b]:
            res.append(l)
            if re.match(r"\s+v_", l):
                res.append("\ts_nop 3\n")
        return res + out[b:]
    if kind == "alloc112":          # the same code with eight more VGPRs allocated: v103 is no longer the last register
        n = 0
        for i in range(len(out)):
            if out[i].strip().startswith(".amdhsa_kernel " + FUNC):
                for j in range(i, i + 60):
                    if ".amdhsa_next_free_vgpr 104" in out[j] or ".amdhsa_accum_offset 104" in out[j]:
                        out[j] = out[j].replace("104", "112"); n += 1
                break
        assert n == 2, n
        return out
    if kind == "amt_moved":         # the shift amount copied out of v103 (into the destination's own high half) first
        old = "\tv_lshlrev_b64 v[70:71], v103, 1\n"
        idx = [i for i in range(a, b) if out[i] == old]
        assert len(idx) == 1, idx
        out[idx[0]] = "\tv_mov_b32_e32 v71, v103\n\tv_lshlrev_b64 v[70:71], v71, 1\n"
        return out
    raise ValueError(kind)

def main():
    if not os.path.exists(ASM):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", ASM, SRC])
    lines = open(ASM).readlines()
    for kind in (sys.argv[1:] or ["ctrl", "vzero", "vones", "vlo", "vhi", "szero", "nops"]):
        p = os.path.join(WORK, "asm_%s.s" % kind)
        open(p, "w").writelines(patch(lines, kind))
        subprocess.check_call([os.path.join(HERE, "asm_build.sh"), SRC, p, os.path.join(ROOT, "tools", "ab", "asm_%s.so" % kind)])

if __name__ == "__main__":
    main()
