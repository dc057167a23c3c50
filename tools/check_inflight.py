#!/usr/bin/env python3
"""Assembly lint for the meshlet cull kernel's hand-issued loads (no GPU needed).

The kernel issues some loads from inline assembly and never lets the compiler wait for them (k_basepass_as.hip: the
occlusion lookup of a step, consumed a step later; continuous mode's requests for the next batch).  The compiler does
not know that the destination registers are written LATER, when the load lands: if it copies, reuses or reads one of
them before a hand-counted `s_waitcnt vmcnt` has covered the load, the result is silently wrong (a dead component of a
128-bit destination is reused at once, and the load then lands on the new value).

For every such load (a `global_load_{ushort,dword,dwordx4}` with a VGPR destination inside an ;;#ASMSTART block of
meshletCullKernel) this walks the instruction stream in layout order and reports any instruction that mentions a
destination register before the first hand-written `s_waitcnt vmcnt(..)` (also inside an ASM block) that FOLLOWS THE NEXT
hand-issued LDS-DMA prefetch or, for the lookups, simply the next hand-written wait.  Layout order approximates control
flow; what it is there to catch -- copies and reuse right behind the load -- sits in the same block.

  python tools/check_inflight.py [--asm file.s]     exit status 1 when a violation is found
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "toyrenderer_amd", "csrc", "k_basepass_as.hip")
FLAGS = "-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-inline-asm --offload-arch=gfx950 --cuda-device-only -S".split()


def regs_of(tok: str):
    """VGPR numbers a token like v12 or v[4:7] names."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def all_regs(line: str):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs_of(tok)
    return out


def touched(t: str):
    """Registers an instruction reads or writes.  A packed-fp32 source pair whose op_sel / op_sel_hi select the low half
    for both results names its second register without reading it."""
    if not t.startswith("v_pk_") or not re.search(r"_f32\b", t.split()[0]):
        return all_regs(t)
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", t.split(None, 1)[1].split(" op_sel")[0].split(" neg_")[0])]
    sel = re.search(r"op_sel:\[([01,]+)\]", t)
    selhi = re.search(r"op_sel_hi:\[([01,]+)\]", t)
    sel = [int(x) for x in sel.group(1).split(",")] if sel else [0, 0, 0]
    selhi = [int(x) for x in selhi.group(1).split(",")] if selhi else [1, 1, 1]
    out = regs_of(ops[0]) if ops else set()
    for i, o in enumerate(ops[1:]):
        o = o.strip("|").lstrip("-")
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", o)
        if not m:
            out |= regs_of(o)
            continue
        lo = int(m.group(1))
        s0 = sel[i] if i < len(sel) else 0
        s1 = selhi[i] if i < len(selhi) else 1
        if s0 == 0 or s1 == 0:
            out.add(lo)
        if s0 == 1 or s1 == 1:
            out.add(lo + 1)
    return out


def kernels(text: str):
    """(name, lines) of every meshletCullKernel instantiation."""
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\S*meshletCullKernel\S*):", lines[i])
        if m:
            j = i
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            yield m.group(1), lines[i:j]
            i = j
        i += 1


def check(name: str, body):
    problems = []
    in_asm = False
    # flatten: (text, inside hand-written asm)
    ins = []
    for l in body:
        t = l.split(";")[0].strip() if ";;#" not in l else l.strip()
        if ";;#ASMSTART" in l:
            in_asm = True
            continue
        if ";;#ASMEND" in l:
            in_asm = False
            continue
        if not t or t.endswith(":") or t.startswith("."):
            continue
        ins.append((t, in_asm))
    for k, (t, hand) in enumerate(ins):
        m = re.match(r"global_load_(ushort|dword|dwordx4)\s+(v\[\d+:\d+\]|v\d+),", t)
        if not (hand and m):
            continue
        dst = regs_of(m.group(2))
        lookup = m.group(1) in ("ushort",) or (m.group(1) == "dword" and "offset" not in t)
        seen_prefetch = lookup          # the next-batch requests are covered by the first counted wait behind a ring prefetch
        for t2, hand2 in ins[k + 1:]:
            if hand2 and t2.startswith("global_load_lds"):
                seen_prefetch = True
            if hand2 and t2.startswith("s_waitcnt") and "vmcnt" in t2 and seen_prefetch:
                break
            if hand2 and re.match(r"global_load_(ushort|dword|dwordx4)\s", t2):
                continue                # the sibling loads of the same asm block (their own destinations are checked in turn)
            if t2.startswith("s_endpgm"):
                break
            hit = touched(t2) & dst
            if hit:
                problems.append(f"{name[:60]}: `{t}` -> `{t2}` touches v{sorted(hit)} while the load is in flight")
                break
    return problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("-D", action="append", default=[])
    args = ap.parse_args()
    if args.asm:
        text = open(args.asm).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-D" + d for d in args.D] + [SRC, "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
    n = 0
    bad = []
    for name, body in kernels(text):
        n += 1
        bad += check(name, body)
    print(f"{n} meshletCullKernel instantiations checked, {len(bad)} in-flight register violations")
    for b in bad:
        print("  " + b)
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
