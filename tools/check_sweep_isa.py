#!/usr/bin/env python3
"""Static check of the hand-scheduled IC(0) sweep kernels (k_sweep_skew<1>, <2>): the record loads
are inline asm that hipcc does not track, so no instruction may READ a load's destination register
between the load and the counted s_waitcnt that retires it.  The kernel's structure guarantees it as
long as the register allocator never copies an in-flight operand; this script verifies exactly
that on the generated ISA: inside the function, any v_mov / v_accvgpr copy whose SOURCE is the
destination of an asm load is reported.

usage: check_sweep_isa.py [k_pcg.s]   (default: compiles euler_amd/csrc/k_pcg.hip to a temp file)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def isa_text():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    out = os.path.join(tempfile.mkdtemp(prefix="sweep_isa_"), "k_pcg.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "euler_amd", "csrc"), "-S",
                           "--cuda-device-only", "-o", out, os.path.join(ROOT, "euler_amd", "csrc", "k_pcg.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(text, name):
    lines = text.split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[i0:i1]
    dest = set()
    in_asm = False
    for l in body:
        t = l.strip()
        if "ASMSTART" in t:
            in_asm = True
        elif "ASMEND" in t:
            in_asm = False
        elif in_asm and t.startswith(("global_load_dwordx2 ", "global_load_dword ")):   # record loads (the x4 poll loads are retired before use)
            dest |= regs(t.split()[1].rstrip(","))
    bad = []
    for l in body:
        t = l.strip()
        if t.startswith(("v_mov_b32_e32", "v_mov_b64", "v_accvgpr_write")):
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            if len(ops) >= 2 and regs(ops[1]) & dest:
                bad.append(t)
    return len(dest), bad


def main():
    text = isa_text()
    rc = 0
    for op in (1, 2):
        name = "_Z12k_sweep_skewILi%dEEv9SweepArgs" % op
        n, bad = check(text, name)
        print("%s: %d operand registers loaded by hand, %d copies of them" % (name, n, len(bad)))
        for b in bad:
            print("   COPY OF AN IN-FLIGHT OPERAND: " + b)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
