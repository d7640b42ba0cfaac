#!/usr/bin/env python3
"""Static check of the hand-scheduled IC(0) sweep kernels (k_sweep_skew<1|2, false|true>).

Their record loads are inline asm that hipcc does not track, retired by hand-counted
`s_waitcnt vmcnt(N)`.  The scheme is only correct if no instruction touches a load's destination
register between the load and the wait that retires it - which the source guarantees as long as
the register allocator never copies or reuses an in-flight operand.  This script verifies it on the
generated ISA by abstract interpretation: it walks every control-flow path of the kernel (forking at
conditional branches, memoised on the program counter and the queue of outstanding loads), models
the in-order vmcnt queue exactly (every global/buffer load, store and atomic enters it; a wait
drains it down to N) and reports any instruction that reads or overwrites a register whose load is
still in the queue.

usage: check_sweep_isa.py [k_pcg.s]   (default: compiles euler_amd/csrc/k_pcg.hip to a temp file)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.setrecursionlimit(100000)


def isa_text():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    out = os.path.join(tempfile.mkdtemp(prefix="sweep_isa_"), "k_pcg.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "euler_amd", "csrc"), "-S",
                           "--cuda-device-only", "-o", out, os.path.join(ROOT, "euler_amd", "csrc", "k_pcg.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def vregs(tok):
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)\b", tok)
    return frozenset({int(m.group(1))}) if m else frozenset()


VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store",
        "flat_atomic", "scratch_")
NO_DEST = ("global_store", "buffer_store", "flat_store", "ds_write", "s_", "v_cmp", "v_cmpx", ";", "scratch_store", "v_nop",
           "buffer_wbl2", "buffer_inv")


def parse(body):
    """-> list of (mnemonic, dest regs, source regs, text), label -> index"""
    ins, labels = [], {}
    for l in body:
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            m = re.match(r"(\.LBB\w+):", t)
            if m:
                labels[m.group(1)] = len(ins)
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ops = [o.split()[0] if o.split() else o for o in ops]          # drop modifiers such as "offset:512", "sc1"
        if mn.startswith(NO_DEST) or mn.startswith("global_load_lds") or not ops:   # (an LDS-DMA has no register destination)
            dest, srcs = frozenset(), ops
        else:
            dest, srcs = vregs(ops[0]), ops[1:]
            if mn.endswith("_dpp") or "sdwa" in mn or mn.startswith(("v_fmac", "v_mac", "v_accvgpr")):
                srcs = ops                                           # the destination is also read (old value / accumulator)
            if mn.startswith(("global_atomic", "buffer_atomic", "flat_atomic")) and " sc0" not in t:
                dest, srcs = frozenset(), ops                        # no return value
        src = frozenset().union(*[vregs(o) for o in srcs]) if srcs else frozenset()
        ins.append((mn, dest, src, t))
    return ins, labels


def check(text, name):
    lines = text.split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins, labels = parse(lines[i0 + 1:i1])
    seen, bad, loads = set(), {}, 0
    stack = [(0, ())]
    while stack:
        pc, fifo = stack.pop()
        while pc < len(ins):
            key = (pc, fifo)
            if key in seen:
                break
            seen.add(key)
            if len(seen) > 2000000:
                raise RuntimeError("state explosion")
            mn, dest, src, t = ins[pc]
            inflight = frozenset().union(*fifo) if fifo else frozenset()
            if mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    fifo = fifo[len(fifo) - n:] if n < len(fifo) else fifo
                    if n == 0:
                        fifo = ()
                pc += 1
                continue
            if (src | dest) & inflight:
                bad.setdefault(pc, t)
            if mn.startswith(VMEM):
                isload = "load" in mn or (mn.startswith(("global_atomic", "buffer_atomic", "flat_atomic")) and bool(dest))
                fifo = fifo + ((dest if isload else frozenset()),)
                loads += 1
                if len(fifo) > 64:
                    fifo = fifo[-64:]                                 # the hardware counter saturates: issue stalls until it drains
            if mn == "s_endpgm":
                break
            if mn == "s_branch":
                pc = labels[t.split()[1]]
                continue
            if mn.startswith("s_cbranch"):
                stack.append((labels[t.split()[1]], fifo))
            pc += 1
    return len(ins), len(seen), bad


def main():
    text = isa_text()
    rc = 0
    # forward / backward solve, single-GPU build and the cross-GPU (system-scope hand-off) build
    for name in ["_Z12k_sweep_skewILi%dELb%dEEv9SweepArgs" % (op, xg) for op in (1, 2) for xg in (0, 1)]:
        n, states, bad = check(text, name)
        print("%s: %d instructions, %d (pc, vmcnt queue) states explored, %d touches of an in-flight operand" % (name, n, states, len(bad)))
        for pc in sorted(bad):
            print("   IN-FLIGHT OPERAND TOUCHED at #%d: %s" % (pc, bad[pc]))
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
