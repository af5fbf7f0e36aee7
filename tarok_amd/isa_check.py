"""Static check of the BUILT library's gfx950 code for the 64-bit-shift erratum ("the refill-role bug").

MI355X (gfx950), like gfx90a, mis-executes `v_lshlrev_b64` / `v_lshrrev_b64` / `v_ashrrev_i64` when the shift AMOUNT
(src0) is the LAST VGPR of the wave's allocation: the hardware treats the 32-bit amount as a 64-bit pair, finds the
second half outside the allocation and substitutes VGPR0 — the shift is by v0's low six bits instead (measured:
tools/shift64_probe.hip, profiles/r04_shift64_probe.txt; 1.8-6.8 % of such shifts, waves whose VGPR block does not
start at row 0).  LLVM knows the erratum for gfx90a (GCNHazardRecognizer::fixShift64HighRegBug) and does not apply
the work-around to gfx950 (ROCm 7.2.0), so any kernel whose register allocation happens to put a shift amount into
its top VGPR computes wrong results now and then — round 3's refill-role corruption.

`check(lib)` disassembles the code object inside the shared library and returns every such instruction;
`tarok_amd.build()` raises on a non-empty list, tests/test_isa_check.py runs it on the shipped library (CPU only:
llvm-objdump reads gfx950 code without a GPU)."""
import os
import re
import subprocess
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
SHIFTS = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")
GRANULE = 8        # VGPR allocation granule of gfx950 (registers per lane)


def _tool(name):
    p = os.path.join(LLVM_BIN, name)
    if not os.path.exists(p):
        raise FileNotFoundError("%s is missing (ROCm's LLVM tools are needed for the ISA check)" % p)
    return p


def extract_code_object(lib_path, out_path):
    """The gfx950 code object bundled into a HIP shared library (.hip_fatbin section)."""
    with tempfile.TemporaryDirectory() as t:
        fat = os.path.join(t, "fat.bin")
        subprocess.check_call([_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        subprocess.check_call([_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                               "--input=" + fat, "--output=" + out_path])
    return out_path


def kernel_allocations(co_path):
    """{kernel symbol: VGPRs the hardware allocates per lane}, read from the kernel DESCRIPTORS (the 64 bytes at
    <kernel>.kd: compute_pgm_rsrc1 bits 5:0 = granulated_workitem_vgpr_count, granule 8 on gfx90a and later) —
    what the dispatcher goes by, not the metadata note."""
    import struct
    data = open(co_path, "rb").read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise RuntimeError("%s is not a 64-bit ELF code object" % co_path)
    e_shoff, = struct.unpack_from("<Q", data, 0x28)
    e_shentsize, e_shnum = struct.unpack_from("<HH", data, 0x3A)
    sections = []
    for i in range(e_shnum):
        sh = struct.unpack_from("<IIQQQQIIQQ", data, e_shoff + i * e_shentsize)
        sections.append((sh[3], sh[4], sh[5]))          # address, file offset, size
    alloc = {}
    syms = subprocess.check_output([_tool("llvm-readelf"), "-s", "--wide", co_path], text=True)
    for line in syms.splitlines():
        f = line.split()
        if len(f) < 8 or not f[7].endswith(".kd"):
            continue
        addr = int(f[1], 16)
        for a, off, size in sections:
            if a and a <= addr and addr + 64 <= a + size:
                rsrc1, = struct.unpack_from("<I", data, off + (addr - a) + 48)
                alloc[f[7][:-3]] = ((rsrc1 & 63) + 1) * GRANULE
                break
    return alloc


def scan_disassembly(text, alloc):
    """Violations in `llvm-objdump -d` output: (kernel, instruction text, amount register, allocation)."""
    bad, kernel = [], None
    sym = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
    ins = re.compile(r"^\s*(%s)(?:_e64)?\s+(\S+),\s*(\S+?),\s*(\S+)" % "|".join(SHIFTS))
    for line in text.splitlines():
        m = sym.match(line)
        if m:
            kernel = m.group(1)
            continue
        m = ins.match(line)
        if not m or kernel not in alloc:
            continue
        amount = m.group(3)
        r = re.fullmatch(r"v(\d+)", amount)
        if r and int(r.group(1)) + 1 >= alloc[kernel]:
            bad.append((kernel, line.split("//")[0].strip(), int(r.group(1)), alloc[kernel]))
    return bad


def check(lib_path):
    """Every 64-bit shift of the library whose amount register is the last VGPR its kernel allocates."""
    with tempfile.TemporaryDirectory() as t:
        co = extract_code_object(lib_path, os.path.join(t, "dev.co"))
        alloc = kernel_allocations(co)
        if not alloc:
            raise RuntimeError("no kernels found in %s" % lib_path)
        text = subprocess.check_output([_tool("llvm-objdump"), "-d", co], text=True)
    return scan_disassembly(text, alloc), alloc


def describe(bad):
    return "\n".join("  %s: `%s` — v%d is the last of the %d VGPRs the kernel allocates" % b for b in bad)


if __name__ == "__main__":
    import sys
    from . import _native
    path = sys.argv[1] if len(sys.argv) > 1 else _native.LIB_PATH
    bad, alloc = check(path)
    print("%s: %d kernels, %d 64-bit shifts with the amount in the last allocated VGPR" % (path, len(alloc), len(bad)))
    if bad:
        print(describe(bad))
        sys.exit(1)
